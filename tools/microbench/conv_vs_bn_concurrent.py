#!/usr/bin/env python3
"""What do a matrix-bound fp32 convolution and an HBM-bound BatchNorm apply pass cost each other when they run concurrently on two streams (the situation of the
two-pass fp32 step)?  Stream A: back-to-back convolutions (tile walk, 256 rows); stream B: back-to-back BatchNorm apply passes.  Reported: each stream's time alone,
both streams' completion times when started together, and the 'serial fraction': (T_both - max(T_a, T_b)) / min(T_a, T_b) -- 0 = perfect overlap, 1 = no overlap at all.
Variants of the BatchNorm stream's grid (LEC_BN_BLOCKS is process-wide, so the cap is emulated by the tensor's row count) are not needed: the library default is used."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from learning_embeddings_amd import ops, _lib
dev = 'cuda'
ops.fusion().schedule = _lib.SCHEDULE_TILE_WALK
A, B = torch.cuda.Stream(), torch.cuda.Stream()


def conv_case(name):
    if name == '3x3 256->256 @14 (l3.c2)':
        x = torch.randn(256, 256, 14, 14, device=dev).contiguous(memory_format=torch.channels_last); w = torch.randn(256, 256, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
        return lambda: ops.conv_f32_fwd(x, w, 1, 1, want_stats=False)
    if name == '3x3 128->128 @28 (l2.c2)':
        x = torch.randn(256, 128, 28, 28, device=dev).contiguous(memory_format=torch.channels_last); w = torch.randn(128, 128, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
        return lambda: ops.conv_f32_fwd(x, w, 1, 1, want_stats=False)
    if name == '1x1 64->256 @56 (l1.c3)':
        x = torch.randn(256, 64, 56, 56, device=dev).contiguous(memory_format=torch.channels_last); w = torch.randn(256, 64, 1, 1, device=dev).contiguous(memory_format=torch.channels_last)
        return lambda: ops.conv_f32_fwd(x, w, 1, 0, want_stats=False)
    if name == '1x1 1024->256 @14 (l3.c1)':
        x = torch.randn(256, 1024, 14, 14, device=dev).contiguous(memory_format=torch.channels_last); w = torch.randn(256, 1024, 1, 1, device=dev).contiguous(memory_format=torch.channels_last)
        return lambda: ops.conv_f32_fwd(x, w, 1, 0, want_stats=False)


big = torch.randn(256, 256, 56, 56, device=dev).contiguous(memory_format=torch.channels_last)
out = torch.empty_like(big)


def bn_pass():
    torch.add(big, 1.0, out=out)                      # a streaming pass of the same byte count as a BatchNorm apply (8 B per element), library kernel


def timed(fa, na, fb, nb):
    torch.cuda.synchronize()
    ea0, ea1, eb0, eb1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
    if fa:
        with torch.cuda.stream(A):
            ea0.record()
            for _ in range(na): fa()
            ea1.record()
    if fb:
        with torch.cuda.stream(B):
            eb0.record()
            for _ in range(nb): fb()
            eb1.record()
    torch.cuda.synchronize()
    ta = ea0.elapsed_time(ea1) if fa else 0.0; tb = eb0.elapsed_time(eb1) if fb else 0.0
    both = max(ea0.elapsed_time(eb1), ea0.elapsed_time(ea1)) if (fa and fb) else max(ta, tb)
    return ta, tb, both


print('| convolution (256 rows, tile walk) | conv alone ms | streaming alone ms | conv when both ms | streaming when both ms | both done ms | serial fraction |')
print('|---|---|---|---|---|---|---|')
for name in ('3x3 256->256 @14 (l3.c2)', '3x3 128->128 @28 (l2.c2)', '1x1 64->256 @56 (l1.c3)', '1x1 1024->256 @14 (l3.c1)'):
    f = conv_case(name)
    for _ in range(3): f(); bn_pass()
    ta, _, _ = timed(f, 40, None, 0)
    per_bn = timed(None, 0, bn_pass, 20)[1] / 20
    nb = max(1, int(round(ta / per_bn)))               # equal amounts of time on both streams
    _, tb, _ = timed(None, 0, bn_pass, nb)
    ca, cb, both = timed(f, 40, bn_pass, nb)
    print('| %s | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f |' % (name, ta, tb, ca, cb, both, (both - max(ta, tb)) / min(ta, tb)))
