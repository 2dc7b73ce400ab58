#!/usr/bin/env python3
"""lec_conv_f32_{fwd,dgrad,wgrad} against MIOpen's fp32 convolutions on the layer shapes of ResNet-50 at the bench batch
(512 images): microseconds per launch and TFLOP/s against the 157.3 TFLOP/s f32 matrix peak.
usage: python tools/bench_conv_f32.py [--rows 512] [--iters 5] [--json out.json]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from learning_embeddings_amd import ops

ap = argparse.ArgumentParser(); ap.add_argument('--rows', type=int, default=512); ap.add_argument('--iters', type=int, default=5)
ap.add_argument('--schedule', default='default', choices=['default', 'tile_walk', 'balanced'], help='which form of the forward / data gradient (default: the library\'s rule; tile_walk: what a two-pass step runs)')
ap.add_argument('--json', default=None); ap.add_argument('--no-lib', action='store_true'); ap.add_argument('--only', default='', help='comma-separated layer names')
a = ap.parse_args()
# (name, Cin, H, Cout, R, stride, pad): the distinct conv shapes of ResNet-50 at 224x224
SHAPES = [('stem', 4, 224, 64, 7, 2, 3),
          ('l1.c1a', 64, 56, 64, 1, 1, 0), ('l1.c2', 64, 56, 64, 3, 1, 1), ('l1.c3', 64, 56, 256, 1, 1, 0), ('l1.c1', 256, 56, 64, 1, 1, 0),
          ('l2.c1a', 256, 56, 128, 1, 1, 0), ('l2.c2s', 128, 56, 128, 3, 2, 1), ('l2.c3', 128, 28, 512, 1, 1, 0), ('l2.ds', 256, 56, 512, 1, 2, 0),
          ('l2.c1', 512, 28, 128, 1, 1, 0), ('l2.c2', 128, 28, 128, 3, 1, 1),
          ('l3.c1a', 512, 28, 256, 1, 1, 0), ('l3.c2s', 256, 28, 256, 3, 2, 1), ('l3.c3', 256, 14, 1024, 1, 1, 0), ('l3.ds', 512, 28, 1024, 1, 2, 0),
          ('l3.c1', 1024, 14, 256, 1, 1, 0), ('l3.c2', 256, 14, 256, 3, 1, 1),
          ('l4.c1a', 1024, 14, 512, 1, 1, 0), ('l4.c2s', 512, 14, 512, 3, 2, 1), ('l4.c3', 512, 7, 2048, 1, 1, 0), ('l4.ds', 1024, 14, 2048, 1, 2, 0),
          ('l4.c1', 2048, 7, 512, 1, 1, 0), ('l4.c2', 512, 7, 512, 3, 1, 1)]
dev = 'cuda'
if a.schedule != 'default':
    from learning_embeddings_amd import ops as _ops, _lib as _l
    _ops.fusion().schedule = _l.SCHEDULE_TILE_WALK if a.schedule == 'tile_walk' else _l.SCHEDULE_BALANCED


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


res = []
tot = {'own': 0.0, 'lib': 0.0}
for name, cin, hw, cout, r, st, pad in SHAPES:
    if a.only and name not in a.only.split(','):
        continue
    N = a.rows
    x = torch.randn(N, cin, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, r, r, device=dev) / (cin * r * r) ** 0.5).contiguous(memory_format=torch.channels_last)
    ho = (hw + 2 * pad - r) // st + 1
    dy = torch.randn(N, cout, ho, ho, device=dev).contiguous(memory_format=torch.channels_last)
    dw = torch.zeros_like(w)
    flops = 2.0 * N * ho * ho * cout * cin * r * r
    row = {'layer': name, 'cin': cin, 'hw': hw, 'cout': cout, 'k': r, 'stride': st, 'gflop': round(flops / 1e9, 1)}
    t = {}
    t['fwd'] = timeit(lambda: ops.conv_f32_fwd(x, w, st, pad), a.iters)
    t['fwd_stats'] = timeit(lambda: ops.conv_f32_fwd(x, w, st, pad, want_stats=True), a.iters)
    if name != 'stem':
        t['dgrad'] = timeit(lambda: ops.conv_f32_dgrad(dy, w, x.shape, st, pad), a.iters)
    t['wgrad'] = timeit(lambda: ops.conv_f32_wgrad(dy, x, dw, st, pad), a.iters)
    if not a.no_lib:
        cb = torch.ops.aten.convolution_backward
        t['lib_fwd'] = timeit(lambda: torch.ops.aten.convolution(x, w, None, [st, st], [pad, pad], [1, 1], False, [0, 0], 1), a.iters)
        if name != 'stem':
            t['lib_dgrad'] = timeit(lambda: cb(dy, x, w, None, [st, st], [pad, pad], [1, 1], False, [0, 0], 1, [True, False, False]), a.iters)
        t['lib_wgrad'] = timeit(lambda: cb(dy, x, w, None, [st, st], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False]), a.iters)
    row.update({k: round(v, 1) for k, v in t.items()})
    row['tflops_fwd'] = round(flops / t['fwd'] / 1e6, 1)
    res.append(row)
    print(json.dumps(row), flush=True)
    del x, w, dy, dw
    torch.cuda.empty_cache()
if a.json:
    json.dump(res, open(a.json, 'w'), indent=1)
