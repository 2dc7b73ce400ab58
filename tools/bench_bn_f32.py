#!/usr/bin/env python3
"""Per-kernel bandwidth of the fused BatchNorm family on fp32 / bf16 activations, through the C ABI (HIP events per call).
usage: bench_bn_f32.py [--dtype fp32|bf16]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning_embeddings_amd import ops
from learning_embeddings_amd._lib import lib, check, dptr, stream_ptr
ap = argparse.ArgumentParser(); ap.add_argument('--dtype', default='fp32'); a = ap.parse_args()
dt = torch.float32 if a.dtype == 'fp32' else torch.bfloat16
es = 4 if a.dtype == 'fp32' else 2
sfx = '_f32' if a.dtype == 'fp32' else ''
dev = 'cuda'


def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for (N, Cc, H) in [(512, 256, 56), (512, 64, 56), (512, 512, 28), (512, 1024, 14), (512, 2048, 7)]:
    M = N * H * H
    x = torch.randn(N, Cc, H, H, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    r = torch.randn_like(x); dy = torch.randn_like(x); y = torch.empty_like(x); dx = torch.empty_like(x); g = torch.empty_like(x)
    w = torch.ones(Cc, device=dev); b = torch.zeros(Cc, device=dev); rm = torch.zeros(Cc, device=dev); rv = torch.ones(Cc, device=dev)
    sm = torch.empty(Cc, device=dev); si = torch.empty(Cc, device=dev); dg = torch.empty(Cc, device=dev); db = torch.empty(Cc, device=dev)
    mask = torch.empty(M * (Cc // 8), dtype=torch.uint8, device=dev)
    ws = ops._bn_workspace(x.device)
    f = lambda name: getattr(lib, name + sfx)
    fwd = lambda: check(f('lec_bn_fwd')(dptr(x), dptr(r), M, Cc, dptr(w), dptr(b), 1e-5, 0.1, dptr(rm), dptr(rv), 1, dptr(sm), dptr(si), dptr(y), 1, dptr(mask), dptr(ws), ws.numel(), stream_ptr()))
    fwd_pre = lambda: check(f('lec_bn_fwd_prestat')(dptr(x), dptr(r), M, Cc, dptr(w), dptr(b), 1e-5, 0.1, dptr(rm), dptr(rv), 64, dptr(sm), dptr(si), dptr(y), 1, dptr(mask), dptr(ws), ws.numel(), stream_ptr()))
    p1 = lambda: check(f('lec_bn_bwd_pass1')(dptr(dy), dptr(r), dptr(mask), dptr(x), M, Cc, dptr(sm), dptr(si), dptr(g), dptr(dg), dptr(db), dptr(ws), ws.numel(), 0, stream_ptr()))
    p2 = lambda: check(f('lec_bn_bwd_apply')(dptr(g), dptr(x), M, Cc, dptr(w), dptr(sm), dptr(si), dptr(dx), dptr(ws), ws.numel(), stream_ptr()))
    el = M * Cc
    t_f, t_fp, t_1, t_2 = t(fwd), t(fwd_pre), t(p1), t(p2)
    t_stats = t_f - t_fp
    print('[%4d,%4d,%2d,%2d] %s  stats %6.1f us %5.2f TB/s | apply+res %6.1f us %5.2f TB/s | bwd pass1 (dy,dy2,x,mask->g) %6.1f us %5.2f TB/s | bwd pass2 %6.1f us %5.2f TB/s'
          % (N, Cc, H, H, a.dtype, t_stats, el * es / t_stats / 1e6, t_fp, (el * es * 3 + el / 8) / t_fp / 1e6, t_1, (el * es * 4 + el / 8) / t_1 / 1e6, t_2, el * es * 3 / t_2 / 1e6))
