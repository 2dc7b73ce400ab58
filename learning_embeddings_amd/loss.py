"""Host mirror of network/loss.py for the hot path: MultiLevelCELoss (loss.py:5-38), config 4's criterion.
forward = one fused HIP launch (softmax statistics, loss, gradient) instead of 4 CE calls + autograd."""
import torch

from . import ops


class MultiLevelCELoss(torch.nn.Module):
    def __init__(self, labelmap, level_weights=None, weight=None):
        torch.nn.Module.__init__(self)
        self.labelmap = labelmap
        self.level_weights = [1.0] * len(self.labelmap.levels) if level_weights is None else level_weights
        # per-class weights (loss.py:16-25): one vector over all n_classes, sliced per level by the reference; the sample's term of a
        # level is scaled by its target class's weight (CrossEntropyLoss(weight=..., reduction='none')).  Held as a buffer-like
        # attribute and moved to the logits' device on first use.
        self.weight = None if weight is None else torch.as_tensor(weight, dtype=torch.float32).reshape(-1).contiguous()
        if self.weight is not None and self.weight.numel() != sum(self.labelmap.levels):
            raise ValueError('weight must hold one entry per class (%d), got %d' % (sum(self.labelmap.levels), self.weight.numel()))
        print('==Using the following weights config for multi level cross entropy loss: {}'.format(self.level_weights))

    def forward(self, outputs, labels, level_labels):
        """criterion(outputs [B, n_classes], labels (unused, as in the reference), level_labels [B, L]) -> mean loss."""
        if self.weight is not None and self.weight.device != outputs.device:
            self.weight = self.weight.to(outputs.device)
        return ops.MultiLevelCEFn.apply(outputs, level_labels, list(self.labelmap.levels), list(self.level_weights), self.weight)
