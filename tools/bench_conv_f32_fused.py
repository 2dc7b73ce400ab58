#!/usr/bin/env python3
"""The fused forms of the fp32 data / weight gradients (lec_conv_f32_dgrad_fused, lec_conv_f32_wgrad_fused) against the kernels and
BatchNorm passes they replace, per stride-1 layer shape of ResNet-50 at the bench batch: microseconds per launch.
usage: python tools/bench_conv_f32_fused.py [--rows 512] [--iters 5]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning_embeddings_amd import ops
from learning_embeddings_amd._lib import lib, check, dptr, stream_ptr

ap = argparse.ArgumentParser(); ap.add_argument('--rows', type=int, default=512); ap.add_argument('--iters', type=int, default=5)
ap.add_argument('--only', default='')
a = ap.parse_args()
SHAPES = [('l1.c1', 256, 56, 64, 1, 0), ('l1.c2', 64, 56, 64, 3, 1), ('l1.c3', 64, 56, 256, 1, 0),
          ('l2.c1', 512, 28, 128, 1, 0), ('l2.c2', 128, 28, 128, 3, 1), ('l2.c3', 128, 28, 512, 1, 0),
          ('l3.c1', 1024, 14, 256, 1, 0), ('l3.c2', 256, 14, 256, 3, 1), ('l3.c3', 256, 14, 1024, 1, 0),
          ('l4.c1', 2048, 7, 512, 1, 0), ('l4.c2', 512, 7, 512, 3, 1), ('l4.c3', 512, 7, 2048, 1, 0)]
dev = 'cuda'


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / iters * 1e3, 1)


cl = lambda t: t.contiguous(memory_format=torch.channels_last)
for name, cin, hw, cout, r, pad in SHAPES:
    if a.only and a.only not in name:
        continue
    N = a.rows
    M = N * hw * hw
    xin = cl(torch.randn(N, cin, hw, hw, device=dev))                   # the layer's input z (a BatchNorm output) and that BatchNorm's input
    xbn_in = cl(torch.randn(N, cin, hw, hw, device=dev)); dres = cl(torch.randn(N, cin, hw, hw, device=dev))
    mask = torch.randint(0, 256, (M * cin // 8,), dtype=torch.uint8, device=dev)
    mean_i = torch.randn(cin, device=dev); inv_i = torch.rand(cin, device=dev) + 0.5
    w = cl(torch.randn(cout, cin, r, r, device=dev) / (cin * r * r) ** 0.5)
    g = cl(torch.randn(N, cout, hw, hw, device=dev)); xbn_out = cl(torch.randn(N, cout, hw, hw, device=dev))
    coef = torch.randn(3 * cout, device=dev); gamma_o = torch.rand(cout, device=dev) + 0.5; mean_o = torch.randn(cout, device=dev); inv_o = torch.rand(cout, device=dev) + 0.5
    dw = torch.zeros_like(w)
    rec = {'x': xbn_in, 'mask': mask, 'mean': mean_i, 'invstd': inv_i, 'dres': dres}
    row = {'layer': name, 'gflop': round(2.0 * M * cout * cin * r * r / 1e9, 1)}
    row['dgrad'] = timeit(lambda: ops.conv_f32_dgrad(g, w, xin.shape, 1, pad), a.iters)
    row['dgrad_fold'] = timeit(lambda: ops.conv_f32_dgrad_fused(g, w, xin.shape, 1, pad, fold=rec), a.iters)
    row['wgrad'] = timeit(lambda: ops.conv_f32_wgrad(g, xin, dw, 1, pad), a.iters)
    if r == 1:
        row['dgrad_xf'] = timeit(lambda: ops.conv_f32_dgrad_fused(g, w, xin.shape, 1, pad, xf=(xbn_out, coef)), a.iters)
        row['dgrad_xf_fold'] = timeit(lambda: ops.conv_f32_dgrad_fused(g, w, xin.shape, 1, pad, xf=(xbn_out, coef), fold=rec), a.iters)
        row['wgrad_xf'] = timeit(lambda: ops.conv_f32_wgrad(g, xin, dw, 1, pad, xf=(xbn_out, coef)), a.iters)
    # the BatchNorm passes they replace: pass 1 of the input-side BatchNorm (reads dz, dres, x, mask; writes g), pass 2 of the output-side one
    ws = ops._bn_workspace(xin.device)
    dg = torch.empty(cin, device=dev); db = torch.empty(cin, device=dev); gout = torch.empty_like(xin)
    row['bn_pass1_in'] = timeit(lambda: check(lib.lec_bn_bwd_pass1_f32(dptr(xin), dptr(dres), dptr(mask), dptr(xbn_in), M, cin, dptr(mean_i), dptr(inv_i), dptr(gout),
                                                                       dptr(dg), dptr(db), dptr(ws), ws.numel(), 0, stream_ptr())), a.iters)
    dxo = torch.empty_like(g)
    row['bn_pass2_out'] = timeit(lambda: check(lib.lec_bn_bwd_apply_f32(dptr(g), dptr(xbn_out), M, cout, dptr(gamma_o), dptr(mean_o), dptr(inv_o), dptr(dxo),
                                                                        dptr(ws), ws.numel(), stream_ptr())), a.iters)
    print(json.dumps(row), flush=True)
    del xin, xbn_in, dres, mask, w, g, xbn_out, dw, gout, dxo
    torch.cuda.empty_cache()
