#!/usr/bin/env python3
"""One layer shape of lec_conv_f32_* in a loop, for rocprofv3 --pmc / --kernel-trace passes.
usage: prof_conv_f32.py [cin hw cout k stride pad] [--rows 512] [--iters 5] [--what fwd|dgrad|wgrad|all]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning_embeddings_amd import ops
ap = argparse.ArgumentParser(); ap.add_argument('shape', nargs='*', type=int, default=[128, 28, 128, 3, 1, 1])
ap.add_argument('--rows', type=int, default=512); ap.add_argument('--iters', type=int, default=5); ap.add_argument('--what', default='all'); ap.add_argument('--x3', action='store_true')
a = ap.parse_args()
cin, hw, cout, r, st, pad = a.shape
x = torch.randn(a.rows, cin, hw, hw, device='cuda').contiguous(memory_format=torch.channels_last)
w = (torch.randn(cout, cin, r, r, device='cuda') / (cin * r * r) ** 0.5).contiguous(memory_format=torch.channels_last)
ho = (hw + 2 * pad - r) // st + 1
dy = torch.randn(a.rows, cout, ho, ho, device='cuda').contiguous(memory_format=torch.channels_last)
dw = torch.zeros_like(w)
if a.x3:
    pf = pt = ops.conv_f32x3_split_weights(w)
for _ in range(a.iters):
    if a.x3:
        if a.what in ('fwd', 'all'): ops.conv_f32x3_fwd(x, pf, st, pad)
        if a.what in ('dgrad', 'all'): ops.conv_f32x3_dgrad(dy, pt, x.shape, st, pad)
        if a.what in ('wgrad', 'all') and hasattr(ops, 'conv_f32x3_wgrad'): ops.conv_f32x3_wgrad(dy, x, dw, st, pad)
        continue
    if a.what in ('fwd', 'all'): ops.conv_f32_fwd(x, w, st, pad)
    if a.what in ('dgrad', 'all'): ops.conv_f32_dgrad(dy, w, x.shape, st, pad)
    if a.what in ('wgrad', 'all'): ops.conv_f32_wgrad(dy, x, dw, st, pad)
torch.cuda.synchronize()
