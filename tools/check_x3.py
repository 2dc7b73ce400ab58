#!/usr/bin/env python3
"""fp32-on-bf16-matrix-core convolutions (lec_conv_f32x3_*) against fp64 and against the native f32-MFMA kernels: error and time.
usage: python tools/check_x3.py [--rows 64] [--iters 3]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from learning_embeddings_amd import ops

ap = argparse.ArgumentParser(); ap.add_argument('--rows', type=int, default=64); ap.add_argument('--iters', type=int, default=3)
ap.add_argument('--time-rows', type=int, default=512); ap.add_argument('--wgrad', action='store_true')
ap.add_argument('--all', action='store_true', help='all 23 ResNet-50 shapes with their multiplicities'); ap.add_argument('--lib-time', action='store_true')
a = ap.parse_args()
SHAPES = [('stem', 4, 224, 64, 7, 2, 3), ('l1.c2', 64, 56, 64, 3, 1, 1), ('l1.c3', 64, 56, 256, 1, 1, 0), ('l1.c1', 256, 56, 64, 1, 1, 0),
          ('l2.c2s', 128, 56, 128, 3, 2, 1), ('l2.ds', 256, 56, 512, 1, 2, 0), ('l2.c2', 128, 28, 128, 3, 1, 1),
          ('l3.c3', 256, 14, 1024, 1, 1, 0), ('l3.c2', 256, 14, 256, 3, 1, 1), ('l4.c1', 2048, 7, 512, 1, 1, 0), ('l4.c2', 512, 7, 512, 3, 1, 1)]
if a.all:
    SHAPES = [('stem', 4, 224, 64, 7, 2, 3, 1),
              ('l1.c1a', 64, 56, 64, 1, 1, 0, 1), ('l1.c2', 64, 56, 64, 3, 1, 1, 3), ('l1.c3', 64, 56, 256, 1, 1, 0, 4), ('l1.c1', 256, 56, 64, 1, 1, 0, 2),
              ('l2.c1a', 256, 56, 128, 1, 1, 0, 1), ('l2.c2s', 128, 56, 128, 3, 2, 1, 1), ('l2.c3', 128, 28, 512, 1, 1, 0, 4), ('l2.ds', 256, 56, 512, 1, 2, 0, 1),
              ('l2.c1', 512, 28, 128, 1, 1, 0, 3), ('l2.c2', 128, 28, 128, 3, 1, 1, 3),
              ('l3.c1a', 512, 28, 256, 1, 1, 0, 1), ('l3.c2s', 256, 28, 256, 3, 2, 1, 1), ('l3.c3', 256, 14, 1024, 1, 1, 0, 6), ('l3.ds', 512, 28, 1024, 1, 2, 0, 1),
              ('l3.c1', 1024, 14, 256, 1, 1, 0, 5), ('l3.c2', 256, 14, 256, 3, 1, 1, 5),
              ('l4.c1a', 1024, 14, 512, 1, 1, 0, 1), ('l4.c2s', 512, 14, 512, 3, 2, 1, 1), ('l4.c3', 512, 7, 2048, 1, 1, 0, 3), ('l4.ds', 1024, 14, 2048, 1, 2, 0, 1),
              ('l4.c1', 2048, 7, 512, 1, 1, 0, 2), ('l4.c2', 512, 7, 512, 3, 1, 1, 2)]
else:
    SHAPES = [s_ + (1,) for s_ in SHAPES]
dev = 'cuda'
cl = lambda t: t.contiguous(memory_format=torch.channels_last)


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def relerr(a_, ref):
    return float((a_.double() - ref).abs().max() / ref.abs().max())


for name, cin, hw, cout, r, st, pad, count in SHAPES:
    torch.manual_seed(0)
    N = a.rows
    x = cl(torch.randn(N, cin, hw, hw, device=dev)); w = cl(torch.randn(cout, cin, r, r, device=dev) / (cin * r * r) ** 0.5)
    ho = (hw + 2 * pad - r) // st + 1
    dy = cl(torch.randn(N, cout, ho, ho, device=dev))
    pf = pt = ops.conv_f32x3_split_weights(w)
    y64 = F.conv2d(x.double(), w.double(), None, st, pad)
    row = {'layer': name, 'count': count}
    y3 = ops.conv_f32x3_fwd(x, pf, st, pad); y1 = ops.conv_f32_fwd(x, w, st, pad)
    row['fwd_err_x3'] = '%.2e' % relerr(y3, y64); row['fwd_err_f32'] = '%.2e' % relerr(y1, y64)
    row['fwd_err_lib'] = '%.2e' % relerr(F.conv2d(x, w, None, st, pad), y64)
    if name != 'stem':
        dx64 = torch.ops.aten.convolution_backward(dy.double(), x.double(), w.double(), None, [st, st], [pad, pad], [1, 1], False, [0, 0], 1, [True, False, False])[0]
        dx3 = ops.conv_f32x3_dgrad(dy, pt, x.shape, st, pad); dx1 = ops.conv_f32_dgrad(dy, w, x.shape, st, pad)
        row['dgrad_err_x3'] = '%.2e' % relerr(dx3, dx64); row['dgrad_err_f32'] = '%.2e' % relerr(dx1, dx64)
        row['dgrad_err_lib'] = '%.2e' % relerr(torch.ops.aten.convolution_backward(dy, x, w, None, [st, st], [pad, pad], [1, 1], False, [0, 0], 1, [True, False, False])[0], dx64)
    wg_ok = a.wgrad and ops.conv_f32x3_wgrad_preferred(cin, cout, r, r)
    if a.wgrad and not wg_ok:
        dw64_ = torch.ops.aten.convolution_backward(dy.double(), x.double(), w.double(), None, [st, st], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        dw1_ = torch.zeros_like(w); ops.conv_f32_wgrad(dy, x, dw1_, st, pad); row['wgrad_err_f32'] = '%.2e' % relerr(dw1_, dw64_)
    if wg_ok:
        dw64 = torch.ops.aten.convolution_backward(dy.double(), x.double(), w.double(), None, [st, st], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        dw3 = torch.zeros_like(w); ops.conv_f32x3_wgrad(dy, x, dw3, st, pad)
        dw1 = torch.zeros_like(w); ops.conv_f32_wgrad(dy, x, dw1, st, pad)
        row['wgrad_err_x3'] = '%.2e' % relerr(dw3, dw64); row['wgrad_err_f32'] = '%.2e' % relerr(dw1, dw64)
        dwl = torch.ops.aten.convolution_backward(dy, x, w, None, [st, st], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        row['wgrad_err_lib'] = '%.2e' % relerr(dwl, dw64)
    del x, dy, y64
    if a.time_rows:
        N = a.time_rows
        x = cl(torch.randn(N, cin, hw, hw, device=dev)); dy = cl(torch.randn(N, cout, ho, ho, device=dev))
        flops = 2.0 * N * ho * ho * cout * cin * r * r
        t3 = timeit(lambda: ops.conv_f32x3_fwd(x, pf, st, pad), a.iters); t1 = timeit(lambda: ops.conv_f32_fwd(x, w, st, pad), a.iters)
        row['fwd_us_x3'] = round(t3, 1); row['fwd_us_f32'] = round(t1, 1); row['fwd_tflops_x3'] = round(flops / t3 / 1e6, 1)
        if name != 'stem':
            t3 = timeit(lambda: ops.conv_f32x3_dgrad(dy, pt, x.shape, st, pad), a.iters); t1 = timeit(lambda: ops.conv_f32_dgrad(dy, w, x.shape, st, pad), a.iters)
            row['dgrad_us_x3'] = round(t3, 1); row['dgrad_us_f32'] = round(t1, 1)
        cb = torch.ops.aten.convolution_backward
        if a.lib_time:
            row['fwd_us_lib'] = round(timeit(lambda: F.conv2d(x, w, None, st, pad), a.iters), 1)
            if name != 'stem':
                row['dgrad_us_lib'] = round(timeit(lambda: cb(dy, x, w, None, [st, st], [pad, pad], [1, 1], False, [0, 0], 1, [True, False, False]), a.iters), 1)
            row['wgrad_us_lib'] = round(timeit(lambda: cb(dy, x, w, None, [st, st], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False]), a.iters), 1)
        if a.wgrad and not wg_ok:
            dw = torch.zeros_like(w); row['wgrad_us_f32'] = round(timeit(lambda: ops.conv_f32_wgrad(dy, x, dw, st, pad), a.iters), 1)
        if wg_ok:
            dw = torch.zeros_like(w)
            t3 = timeit(lambda: ops.conv_f32x3_wgrad(dy, x, dw, st, pad), a.iters); t1 = timeit(lambda: ops.conv_f32_wgrad(dy, x, dw, st, pad), a.iters)
            row['wgrad_us_x3'] = round(t3, 1); row['wgrad_us_f32'] = round(t1, 1)
        del x, dy
    print(json.dumps(row), flush=True)
    torch.cuda.empty_cache()
