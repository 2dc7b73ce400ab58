"""Host mirror of network/loss.py for the hot path: MultiLevelCELoss (loss.py:5-38), config 4's criterion.
forward = one fused HIP launch (softmax statistics, loss, gradient) instead of 4 CE calls + autograd."""
import torch

from . import ops


class MultiLevelCELoss(torch.nn.Module):
    def __init__(self, labelmap, level_weights=None, weight=None):
        torch.nn.Module.__init__(self)
        self.labelmap = labelmap
        self.level_weights = [1.0] * len(self.labelmap.levels) if level_weights is None else level_weights
        if weight is not None:
            raise NotImplementedError('per-class weights (loss.py:16-25) are outside the hot path (ethec_experiments passes None)')
        print('==Using the following weights config for multi level cross entropy loss: {}'.format(self.level_weights))

    def forward(self, outputs, labels, level_labels):
        """criterion(outputs [B, n_classes], labels (unused, as in the reference), level_labels [B, L]) -> mean loss."""
        return ops.MultiLevelCEFn.apply(outputs, level_labels, list(self.labelmap.levels), list(self.level_weights))
