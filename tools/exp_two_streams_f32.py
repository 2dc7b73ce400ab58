#!/usr/bin/env python3
"""Experiment (fp32, the headline precision): ResNet-50 fwd+bwd of 512 rows as ONE pass against TWO independent 256-row passes on two
HIP streams -- the matrix-bound convolutions of one pass beside the HBM-bound BatchNorm passes of the other -- with and without the
weight gradients on a third stream.  Eager launches (the host enqueues a step in 10-20 ms against > 100 ms of GPU time).
usage: python tools/exp_two_streams_f32.py      (LEC_FOLD_BN_BWD_F32 / LEC_LAZY_BN_PASS2_F32 = 0 for the unfused BatchNorm backward)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learning_embeddings_amd.resnet import resnet50, WgradOverlap  # noqa: E402


def build():
    torch.manual_seed(0)
    m = resnet50(num_classes=10).cuda().to(memory_format=torch.channels_last).train()
    for p in m.parameters():
        p.grad = torch.zeros_like(p)
    return m


def run(model, x, g):
    model(x).backward(g)


def timed(fn, reps=6):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    R = int(os.environ.get('ROWS', 512))
    m1, m2 = build(), build()
    x = torch.rand(R, 3, 224, 224, device='cuda').contiguous(memory_format=torch.channels_last)
    g = torch.randn(R, 10, device='cuda')
    h = R // 2
    xa, xb = x[:h].contiguous(memory_format=torch.channels_last), x[h:].contiguous(memory_format=torch.channels_last)
    ga, gb = g[:h].contiguous(), g[h:].contiguous()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def one():
        run(m1, x, g)
        if WgradOverlap.instance.side is not None:
            WgradOverlap.instance.join()

    def seq_halves():
        run(m1, xa, ga); run(m2, xb, gb)
        if WgradOverlap.instance.side is not None:
            WgradOverlap.instance.join()

    def two_streams():
        cur = torch.cuda.current_stream()
        sa.wait_stream(cur); sb.wait_stream(cur)
        with torch.cuda.stream(sa):
            run(m1, xa, ga)
        with torch.cuda.stream(sb):
            run(m2, xb, gb)
        cur.wait_stream(sa); cur.wait_stream(sb)
        if WgradOverlap.instance.side is not None:
            WgradOverlap.instance.join()

    def two_streams_staggered():
        """forward of half b starts when half a's forward is done: b's forward runs beside a's backward"""
        cur = torch.cuda.current_stream()
        sa.wait_stream(cur); sb.wait_stream(cur)
        with torch.cuda.stream(sa):
            ya = m1(xa)
        with torch.cuda.stream(sb):
            yb = m2(xb)
        with torch.cuda.stream(sa):
            ya.backward(ga)
        with torch.cuda.stream(sb):
            yb.backward(gb)
        cur.wait_stream(sa); cur.wait_stream(sb)
        if WgradOverlap.instance.side is not None:
            WgradOverlap.instance.join()

    for side in (True, False):
        WgradOverlap.instance = WgradOverlap(side_stream=side)
        tag = 'wgrad on a side stream' if side else 'wgrad in line'
        print('[%s] one pass of %d rows          : %.2f ms' % (tag, R, timed(one)), flush=True)
        print('[%s] two passes of %d, in turn    : %.2f ms' % (tag, h, timed(seq_halves)), flush=True)
        print('[%s] two passes of %d, 2 streams  : %.2f ms' % (tag, h, timed(two_streams)), flush=True)
        print('[%s] two passes, fwd a | fwd b | bwd a | bwd b enqueue order: %.2f ms' % (tag, timed(two_streams_staggered)), flush=True)
    WgradOverlap.instance = None


if __name__ == '__main__':
    main()
