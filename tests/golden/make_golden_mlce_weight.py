#!/usr/bin/env python3
"""Generate fixture F8b (MultiLevelCELoss with per-class weights) by IMPORTING the reference -- build container only.

    python tests/golden/make_golden_mlce_weight.py     # rewrites tests/golden/F8b_multilevel_ce_class_weights.npz

Reference entry point (file:line under /root/reference): network/loss.py:5-38 MultiLevelCELoss(labelmap, level_weights, weight) -- the
`weight is not None` branch (:16-25: one nn.CrossEntropyLoss(weight=weight[level slice], reduction='none') per level) + autograd.
Inputs are synthetic (ETHEC's level sizes 6 / 21 / 135 / 561); outputs are the reference's loss and d loss / d logits.
"""
import os, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, SynthLabelMap  # noqa: E402


def main():
    import torch
    import_reference()
    import importlib
    lossm = importlib.import_module('network.loss')
    lm = SynthLabelMap([6, 21, 135, 561])
    g = torch.Generator().manual_seed(11)
    B = 24
    logits = (torch.randn(B, lm.n_classes, generator=g) * 3).requires_grad_(True)
    lvl = torch.stack([torch.randint(0, n, (B,), generator=g) for n in lm.levels], dim=1)
    cw = torch.rand(lm.n_classes, generator=g) * 2 + 0.1
    out = {'logits': logits.detach().numpy(), 'level_labels': lvl.numpy(), 'levels': np.array(lm.levels), 'class_weights': cw.numpy(),
           'level_weights_w': np.array([1.0, 0.5, 2.0, 4.0], dtype=np.float32)}
    for tag, w in (('unw', None), ('w', [1.0, 0.5, 2.0, 4.0])):
        logits.grad = None
        crit = lossm.MultiLevelCELoss(lm, level_weights=w, weight=cw)
        ls = crit(logits, None, lvl)
        ls.backward()
        out[tag + '_loss'] = ls.detach().numpy(); out[tag + '_glogits'] = logits.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'F8b_multilevel_ce_class_weights.npz'), **out)
    print('F8b written: loss %.6f / %.6f' % (float(out['unw_loss']), float(out['w_loss'])))


if __name__ == '__main__':
    main()
