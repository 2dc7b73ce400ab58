#!/usr/bin/env python3
"""Per-shape timing of ResNet-50's 1x1 stride-1 convolutions (batch 512, bf16 NHWC): MIOpen (the solver the shipped
find-db selects) against the same contraction issued as a plain GEMM (hipBLASLt through torch.mm), for forward, data
gradient and weight gradient.  Used to decide, per shape and direction, which library call `resnet.Conv2d` makes.

    python tools/bench_conv1x1.py > gpurun_out/conv1x1.json
"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learning_embeddings_amd import miopen_tuning  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3          # us


def main():
    miopen_tuning.setup()
    B = int(os.environ.get('LEC_B', 512))
    dev = 'cuda'
    shapes = [(64, 64, 56), (64, 256, 56), (256, 64, 56), (256, 128, 56), (128, 512, 28), (512, 128, 28), (512, 256, 28),
              (256, 1024, 14), (1024, 256, 14), (1024, 512, 14), (512, 2048, 7), (2048, 512, 7)]
    for ci, co, hw in shapes:
        x = torch.randn(B, ci, hw, hw, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = torch.randn(co, ci, 1, 1, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(B, co, hw, hw, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        M = B * hw * hw
        x2 = x.permute(0, 2, 3, 1).reshape(M, ci); dy2 = dy.permute(0, 2, 3, 1).reshape(M, co); w2 = w.view(co, ci)
        assert x2.is_contiguous() and dy2.is_contiguous()
        cb = torch.ops.aten.convolution_backward
        args = ([0], [1, 1], [0, 0], [1, 1], False, [0, 0], 1)
        r = {'cin': ci, 'cout': co, 'hw': hw, 'M': M, 'gflop': 2.0 * M * ci * co / 1e9}
        r['fwd_miopen_us'] = timed(lambda: torch.nn.functional.conv2d(x, w))
        r['fwd_gemm_us'] = timed(lambda: x2 @ w2.t())
        r['dgrad_miopen_us'] = timed(lambda: cb(dy, x, w, *args, [True, False, False]))
        r['dgrad_gemm_us'] = timed(lambda: dy2 @ w2)
        r['wgrad_miopen_us'] = timed(lambda: cb(dy, x, w, *args, [False, True, False]))
        r['wgrad_gemm_us'] = timed(lambda: dy2.t() @ x2)
        r['ideal_us'] = max(r['gflop'] * 1e9 / 2.5e15, (M * (ci + co) * 2) / 6.3e12) * 1e6
        print(json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items()}), flush=True)


if __name__ == '__main__':
    main()
