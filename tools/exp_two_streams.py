#!/usr/bin/env python3
"""Experiment: ResNet-50 fwd+bwd (fused BN kernels, bf16 NHWC) as ONE 512-row pass on one stream against TWO independent
256-row passes on two streams (HBM-bound BatchNorm of one pass beside MFMA-bound convs of the other).  Both variants
are captured in a hipGraph so that host launch time does not matter."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learning_embeddings_amd import miopen_tuning
miopen_tuning.setup()
from learning_embeddings_amd.resnet import resnet50, WgradOverlap  # noqa: E402


def build():
    m = resnet50(); m.fc = torch.nn.Linear(2048, 10)
    return m.cuda().to(memory_format=torch.channels_last).train()


def run(model, x, g):
    with torch.autocast('cuda', dtype=torch.bfloat16):
        y = model(x)
    y.float().backward(g)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    WgradOverlap.instance = None
    R = int(os.environ.get('ROWS', 512))
    m1, m2 = build(), build()
    x = torch.rand(R, 3, 224, 224, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    g = torch.randn(R, 10, device='cuda')
    h = R // 2
    xa, xb = x[:h].contiguous(memory_format=torch.channels_last), x[h:].contiguous(memory_format=torch.channels_last)
    ga, gb = g[:h].contiguous(), g[h:].contiguous()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def one():
        run(m1, x, g)

    def seq_halves():
        run(m1, xa, ga); run(m2, xb, gb)

    def two_streams():
        cur = torch.cuda.current_stream()
        sa.wait_stream(cur); sb.wait_stream(cur)
        with torch.cuda.stream(sa):
            run(m1, xa, ga)
        with torch.cuda.stream(sb):
            run(m2, xb, gb)
        cur.wait_stream(sa); cur.wait_stream(sb)

    print('one pass of %d rows        : %.2f ms' % (R, timed(one)), flush=True)
    print('two passes of %d, in turn  : %.2f ms' % (h, timed(seq_halves)), flush=True)
    print('two passes of %d, 2 streams: %.2f ms' % (h, timed(two_streams)), flush=True)


if __name__ == '__main__':
    main()
