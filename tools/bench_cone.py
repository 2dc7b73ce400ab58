#!/usr/bin/env python3
"""Micro-benchmark of the fused joint-loss kernel and the table step (HIP events on the launch stream).
Algorithmic bytes per positive (fwd+bwd): (2+2K)(2*D*4 + 4*D + 4) + (1+2K)*8."""
import argparse, json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from learning_embeddings_amd import ops, _lib


def alg_bytes(B, K, D):
    return B * ((2 + 2 * K) * (2 * D * 4 + 4 * D + 4) + (1 + 2 * K) * 8)


def time_joint(B, K, D, N, M, iters=50, grad=True):
    dev = 'cuda'
    g = torch.Generator(device='cpu').manual_seed(0)
    W = torch.randn(N, D, generator=g); W = (W / W.norm(dim=1, keepdim=True) * (0.1 + 0.05 * torch.rand(N, 1, generator=g))).to(dev)
    R = (torch.randn(M, D, generator=g) * 0.3).to(dev)
    frm = torch.randint(0, N, (B,), generator=g, dtype=torch.int32).to(dev)
    to = (-1 - torch.randint(0, M, (B,), generator=g, dtype=torch.int32)).to(dev)
    neg = torch.randint(0, N, (B, 2 * K), generator=g, dtype=torch.int32).to(dev)
    gt = torch.zeros_like(W) if grad else None; gf = torch.zeros_like(R) if grad else None
    for _ in range(5):
        ops.joint_loss_raw(W, R, frm, to, neg, None, 0.1, 0.01, 0, 1, 1, gt, gf)
    torch.cuda.synchronize()
    # Launches are captured into a HIP graph and replayed, so that the per-launch time is the kernel's (plus the ~1.5 us
    # dependent-launch boundary), not the Python/ctypes launch path's: at the small sizes the host is slower than the GPU.
    t = None
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            g_ = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_, stream=side):
                for _ in range(iters):
                    ops.joint_loss_raw(W, R, frm, to, neg, None, 0.1, 0.01, 0, 1, 1, gt, gf)
            g_.replay(); side.synchronize()
            ts = []
            for _ in range(5):
                a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
                a.record(side); g_.replay(); b.record(side); side.synchronize()
                ts.append(a.elapsed_time(b) * 1e-3 / iters)
            t = sorted(ts)[len(ts) // 2]
        torch.cuda.current_stream().wait_stream(side)
    except Exception as e:                                      # capture unsupported: eager timing
        print('graph capture failed (%s); eager timing' % e, file=sys.stderr)
    if t is None:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for a, b in evs:
            a.record(); ops.joint_loss_raw(W, R, frm, to, neg, None, 0.1, 0.01, 0, 1, 1, gt, gf); b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) * 1e-3 for a, b in evs)
        t = ts[len(ts) // 2]
    return {'B': B, 'K': K, 'D': D, 'N': N, 'pairs': B * (1 + 2 * K), 'us': t * 1e6, 'alg_MB': alg_bytes(B, K, D) / 1e6,
            'GBps': alg_bytes(B, K, D) / t / 1e9, 'Mpairs_s': B * (1 + 2 * K) / t / 1e6}


def time_table(N, D, iters=50):
    dev = 'cuda'
    W = torch.randn(N, D, device=dev) * 0.05; g = torch.randn(N, D, device=dev); m = torch.zeros_like(W); v = torch.zeros_like(W)
    for _ in range(5):
        ops.table_step_adam(W, g, m, v, 1, 1e-4, 0.1)
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        ops.table_step_adam(W, g, m, v, i + 1, 1e-4, 0.1)
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) * 1e-3 / iters
    return {'N': N, 'D': D, 'us': t * 1e6, 'GBps': 7 * N * D * 4 / t / 1e9}


def time_bn(N=512, C=256, H=56, W=56, residual=True, iters=20):
    """Fused BatchNorm(+residual)+ReLU forward and backward on one ResNet-50 layer shape (NHWC bf16): achieved HBM GB/s
    against the algorithmic bytes  fwd: x (stats) + x + y [+ residual] + mask ; bwd: 2 x (dy + x + mask) + dx [+ d residual]."""
    dev = 'cuda'
    x = torch.randn(N, C, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    r = torch.randn(N, C, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) if residual else None
    dy = torch.randn(N, C, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    rm = torch.zeros(C, device=dev); rv = torch.ones(C, device=dev)
    el = N * C * H * W
    fwd_b = el * (2 + 2 + 2 + (2 if residual else 0) + 0.125); bwd_b = el * (2 * (2 + 2 + 0.125) + 2 + (2 if residual else 0))
    from learning_embeddings_amd._lib import lib, check, dptr, stream_ptr
    y = torch.empty_like(x); dx = torch.empty_like(x); dr = torch.empty_like(x) if residual else None
    mask = torch.empty(el // 8, dtype=torch.uint8, device=dev)
    sm = torch.empty(C, device=dev); si = torch.empty(C, device=dev); dg = torch.empty(C, device=dev); db = torch.empty(C, device=dev)
    ws = ops._bn_workspace(x.device)
    M = N * H * W
    def run():                                                  # straight through the C ABI: no autograd bookkeeping in the timing
        check(lib.lec_bn_fwd(dptr(x), dptr(r), M, C, dptr(w), dptr(b), 1e-5, 0.1, dptr(rm), dptr(rv), 1, dptr(sm), dptr(si), dptr(y), 1,
                             dptr(mask), dptr(ws), ws.numel(), stream_ptr()))
        check(lib.lec_bn_bwd(dptr(dy), None, None, dptr(mask), dptr(x), M, C, dptr(w), dptr(sm), dptr(si), dptr(dx), dptr(dr), dptr(dg), dptr(db), 1,
                             dptr(ws), ws.numel(), 0, stream_ptr()))
    for _ in range(3): run()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); c = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): run()
    c.record(); torch.cuda.synchronize()
    t = a.elapsed_time(c) * 1e-3 / iters
    return {'shape': [N, C, H, W], 'residual': residual, 'us_fwd_bwd': t * 1e6, 'alg_MB': (fwd_b + bwd_b) / 1e6, 'GBps': (fwd_b + bwd_b) / t / 1e9}


if __name__ == '__main__':
    out = []
    for B, K, D, N, M in ((128, 5, 10, 723, 128), (256, 5, 10, 2000, 256), (256, 256, 10, 50000, 256), (256, 256, 128, 50000, 256),
                          (4096, 256, 10, 50000, 4096), (4096, 64, 128, 50000, 4096), (16384, 5, 10, 2000, 16384)):
        r = time_joint(B, K, D, N, M); out.append(r); print(json.dumps(r))
    for shp in ((512, 256, 56, 56), (512, 64, 112, 112), (512, 1024, 14, 14)):
        print(json.dumps(time_bn(*shp)))
    for N, D in ((723, 10), (50000, 10), (50000, 128), (1000000, 128)):
        r = time_table(N, D); print(json.dumps(r))
