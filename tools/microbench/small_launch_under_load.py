#!/usr/bin/env python3
"""How long a tiny dependent launch takes on stream B while stream A keeps the chip busy (round 4: the BatchNorm finalize launches of the two-pass step take 7 us alone and ~49 us
next to the other pass's kernels).  Stream A: back-to-back fp32 convolutions (or BatchNorm apply passes, or nothing); stream B: 200 back-to-back launches of (a) a one-element fill,
(b) lec_bn_fwd_finalize on 512 partial rows x 256 channels; HIP events around the 200."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from learning_embeddings_amd import ops, _lib
dev = 'cuda'
x = torch.randn(256, 256, 28, 28, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(256, 256, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
big = torch.randn(256, 256, 56, 56, device=dev).contiguous(memory_format=torch.channels_last)
C = int(os.environ.get('FIN_C', '256'))                       # channels of the finalize launch: C / 8 workgroups
ROWS = int(os.environ.get('FIN_ROWS', '512'))
ws = torch.zeros(_lib.lib.lec_bn_workspace_bytes(C), dtype=torch.uint8, device=dev)
gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev); rm = torch.zeros(C, device=dev); rv = torch.ones(C, device=dev)
sm = torch.empty(C, device=dev); si = torch.empty(C, device=dev)
one = torch.zeros(1, device=dev)
A, B = torch.cuda.Stream(), torch.cuda.Stream()


def load_conv(n):
    for _ in range(n):
        ops.conv_f32_fwd(x, w, 1, 1, want_stats=False)


def load_bn(n):
    for _ in range(n):
        torch.add(big, 1.0, out=big)


def small_fill():
    one.fill_(1.0)


def small_finalize():
    _lib.check(_lib.lib.lec_bn_fwd_finalize(256 * 28 * 28, C, _lib.dptr(gamma), _lib.dptr(beta), 1e-5, 0.1, _lib.dptr(rm), _lib.dptr(rv), ROWS, _lib.dptr(sm), _lib.dptr(si),
                                            _lib.dptr(ws), ws.numel(), _lib.stream_ptr()))


for lname, load in (('nothing', None), ('fp32 3x3 convolutions', load_conv), ('streaming elementwise passes', load_bn)):
    for sname, small in (('one-element fill', small_fill), ('bn_fwd_finalize %d x %d' % (ROWS, C), small_finalize)):
        torch.cuda.synchronize()
        if load is not None:
            with torch.cuda.stream(A):
                load(60 if load is load_conv else 200)
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
        print('stream A: %-30s stream B: %-28s %.1f us per launch%s' % (lname, sname, e0.elapsed_time(e1) * 1e3 / 200, '' if (load is None or busy) else '  (stream A ran dry before B finished)'))
