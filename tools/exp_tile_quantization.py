#!/usr/bin/env python3
"""How much of a convolution kernel's time is tile quantization: the same 3x3 256 -> 256 layer (and a 1x1 1024 -> 256) at pixel counts that give
1 568 tiles (ResNet-50 layer3 at 512 rows: 3.06 rounds of 512 workgroup slots), 784 (1.53), 392 (0.77) and at counts that fill whole rounds
(1 024, 512, 1 536 tiles).  TFLOP/s per case; the ratio is the prize of a balanced (stream-K) schedule, whose numbers stand beside it (schedule 0 / 1).  usage: python tools/exp_tile_quantization.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning_embeddings_amd import ops


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


print('| layer | input | tiles | rounds of 512 | forward, tile walk: us / TFLOP/s | forward, balanced | data gradient, tile walk | data gradient, balanced |')
print('|---|---|---|---|---|---|---|---|')
for cin, cout, r, pad in ((256, 256, 3, 1), (1024, 256, 1, 0), (512, 512, 3, 1)):
    w = (torch.randn(cout, cin, r, r, device='cuda') / (cin * r * r) ** 0.5).contiguous(memory_format=torch.channels_last)
    for n, h, wd in ((512, 14, 14), (256, 14, 14), (128, 14, 14), (512, 8, 16), (256, 8, 16), (512, 12, 16), (512, 7, 7), (256, 7, 7), (320, 7, 7), (640, 7, 7)):
        x = torch.randn(n, cin, h, wd, device='cuda').contiguous(memory_format=torch.channels_last)
        dy = torch.randn(n, cout, h, wd, device='cuda').contiguous(memory_format=torch.channels_last)
        M = n * h * wd
        tiles = -(-M // 128) * -(-cout // 128)
        fl = 2.0 * M * cout * cin * r * r
        row = []
        for mode in (0, 1):                                      # tile walk | the launcher's choice (balanced where the last round is under 92 % full)
            prev = ops.fusion().schedule; ops.fusion().schedule = mode
            tf = timeit(lambda: ops.conv_f32_fwd(x, w, 1, pad, want_stats=True)); td = timeit(lambda: ops.conv_f32_dgrad(dy, w, x.shape, 1, pad))
            ops.fusion().schedule = prev
            row += [tf, fl / tf / 1e6, td, fl / td / 1e6]
        print('| %d -> %d %dx%d | %d x %dx%d | %d | %.2f | %.0f / %.1f | %.0f / %.1f | %.0f / %.1f | %.0f / %.1f |'
              % (cin, cout, r, r, n, h, wd, tiles, tiles / 512.0, row[0], row[1], row[4], row[5], row[2], row[3], row[6], row[7]), flush=True)
