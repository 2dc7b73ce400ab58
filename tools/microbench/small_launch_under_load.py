#!/usr/bin/env python3
"""How long a tiny dependent launch takes on stream B while stream A keeps the chip busy (round 4: the BatchNorm finalize launches of the two-pass step take 7 us alone and ~49 us
next to the other pass's kernels).  Stream A: back-to-back fp32 convolutions (or BatchNorm apply passes, or nothing); stream B: 200 back-to-back launches of (a) a one-element fill,
(b) lec_bn_fwd_finalize on 512 partial rows x 256 channels; HIP events around the 200."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from learning_embeddings_amd import ops, _lib
dev = 'cuda'
# which form of the forward kernel stream A runs: the tile walk (what a two-pass step uses) unless CONV_SCHEDULE=auto (then the balanced / stream-K kernel takes this shape)
if os.environ.get('CONV_SCHEDULE', 'tile_walk') == 'tile_walk':
    ops.fusion().schedule = _lib.SCHEDULE_TILE_WALK
COUT = int(os.environ.get('CONV_COUT', '256'))                # 64: the narrow tile (128 x 64, 55 KB of LDS per workgroup instead of 74)
x = torch.randn(256, 256, 28, 28, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(COUT, 256, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
big = torch.randn(256, 256, 56, 56, device=dev).contiguous(memory_format=torch.channels_last)
C = int(os.environ.get('FIN_C', '256'))                       # channels of the finalize launch: C / 8 workgroups
ROWS = int(os.environ.get('FIN_ROWS', '512'))
ws = torch.zeros(_lib.lib.lec_bn_workspace_bytes(C), dtype=torch.uint8, device=dev)
gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev); rm = torch.zeros(C, device=dev); rv = torch.ones(C, device=dev)
sm = torch.empty(C, device=dev); si = torch.empty(C, device=dev)
one = torch.zeros(1, device=dev)
A, B = torch.cuda.Stream(), torch.cuda.Stream(priority=-1 if os.environ.get("B_HIGH") else 0)


def load_conv(n):
    for _ in range(n):
        ops.conv_f32_fwd(x, w, 1, 1, want_stats=False)


def load_bn(n):
    for _ in range(n):
        torch.add(big, 1.0, out=big)


ma = torch.randn(8192, 8192, device=dev); mb = torch.randn(8192, 8192, device=dev); mc = torch.empty(8192, 8192, device=dev)
xl = x.clone(); wl = w.clone()


def load_gemm(n):                                               # the library's fp32 GEMM (hipBLASLt / rocBLAS): someone else's MFMA kernel
    for _ in range(n):
        torch.matmul(ma, mb, out=mc)


def load_libconv(n):                                            # MIOpen's fp32 convolution on the same shape
    for _ in range(n):
        torch.nn.functional.conv2d(xl, wl, padding=1)


def small_fill():
    one.fill_(1.0)


def small_finalize():
    _lib.check(_lib.lib.lec_bn_fwd_finalize(256 * 28 * 28, C, _lib.dptr(gamma), _lib.dptr(beta), 1e-5, 0.1, _lib.dptr(rm), _lib.dptr(rv), ROWS, _lib.dptr(sm), _lib.dptr(si),
                                            _lib.dptr(ws), ws.numel(), _lib.stream_ptr()))


sc_ = torch.empty(C, device=dev); sh_ = torch.empty(C, device=dev)
small_vec = torch.zeros(8192, device=dev)


def small_coeffs():                                             # one workgroup of 256 threads, 13 registers, no LDS (bn_eval_coeff_kernel)
    _lib.check(_lib.lib.lec_bn_eval_coeffs_f32(C, _lib.dptr(gamma), _lib.dptr(beta), 1e-5, _lib.dptr(rm), _lib.dptr(rv), _lib.dptr(sc_), _lib.dptr(sh_), _lib.stream_ptr()))


def small_add():                                                # a framework elementwise kernel over 32 KB
    small_vec.add_(1.0)


for lname, load in (('nothing', None), ('fp32 3x3 convolutions', load_conv), ('library fp32 GEMM 8192^3', load_gemm), ('library fp32 3x3 convolution', load_libconv), ('streaming elementwise passes', load_bn)):
    for sname, small in (('one-element fill', small_fill), ('32 KB framework add', small_add), ('bn_eval_coeffs (1 workgroup, no LDS)', small_coeffs), ('bn_fwd_finalize %d x %d' % (ROWS, C), small_finalize)):
        torch.cuda.synchronize()
        a0 = torch.cuda.Event(enable_timing=True); a1 = torch.cuda.Event(enable_timing=True)
        nA = 200 if load is load_bn else 60
        if load is not None:
            with torch.cuda.stream(A):
                a0.record(); load(nA); a1.record()
        time.sleep(0.002)
        with torch.cuda.stream(B):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            small(); e0.record()
            for _ in range(200):
                small()
            e1.record()
        e1.synchronize()
        busy = not A.query()
        torch.cuda.synchronize()
        print('stream A: %-30s stream B: %-40s %.1f us per launch%s%s' % (lname, sname, e0.elapsed_time(e1) * 1e3 / 200, '' if (load is None or busy) else '  (stream A ran dry before B finished)',
              '' if load is None else ';  stream A: %.0f us per kernel' % (a0.elapsed_time(a1) * 1e3 / nA)))
if True:                                                        # stream A alone, for the per-kernel baseline
    for lname, load, nA in (('fp32 3x3 convolutions', load_conv, 60), ('streaming elementwise passes', load_bn, 200)):
        torch.cuda.synchronize()
        a0 = torch.cuda.Event(enable_timing=True); a1 = torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(A):
            a0.record(); load(nA); a1.record()
        torch.cuda.synchronize()
        print('stream A alone: %-30s %.0f us per kernel' % (lname, a0.elapsed_time(a1) * 1e3 / nA))
