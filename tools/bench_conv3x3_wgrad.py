#!/usr/bin/env python3
"""lec_conv3x3_c64_wgrad (MFMA weight gradient of layer1's 3x3 convolution, float atomics into the gradient buffer) against the
library's weight-gradient convolution (+ the copy of its result into the fp32 gradient slot), at the bench batch."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learning_embeddings_amd import miopen_tuning; miopen_tuning.setup()
from learning_embeddings_amd import ops


def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    B = int(os.environ.get('LEC_B', 512)); hw = 56
    g = torch.Generator(device='cpu').manual_seed(1)
    x = (torch.randn(B // 8, 64, hw, hw, generator=g) * 0.7).to('cuda').to(torch.bfloat16).repeat(8, 1, 1, 1).contiguous(memory_format=torch.channels_last)
    dy = (torch.randn(B // 8, 64, hw, hw, generator=g) * 0.1).to('cuda').to(torch.bfloat16).repeat(8, 1, 1, 1).contiguous(memory_format=torch.channels_last)
    dw = torch.zeros(64, 64, 3, 3, device='cuda').contiguous(memory_format=torch.channels_last)
    ops.conv3x3_c64_wgrad(dy, x, dw)
    w4 = torch.zeros(64, 64, 3, 3, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ref = torch.ops.aten.convolution_backward(dy.float(), x.float(), w4.float(), None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (False, True, False))[1]
    err = ((dw - ref).abs().max() / ref.abs().max()).item()
    slot = torch.zeros_like(dw)

    def lib():
        gw = torch.ops.aten.convolution_backward(dy, x, w4, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (False, True, False))[1]
        slot.copy_(gw)
    t_own = timed(lambda: ops.conv3x3_c64_wgrad(dy, x, dw)); t_lib = timed(lib)
    M = B * hw * hw
    print(json.dumps({'shape': '64->64 3x3 @%dx%d' % (hw, hw), 'images': B, 'max_rel_err_vs_fp32': err, 'own_us': round(t_own, 1), 'library_us': round(t_lib, 1),
                      'own_GBps': round(M * 128 * 2 / t_own / 1e3, 1), 'own_TFLOPs': round(2 * M * 64 * 64 * 9 / t_own / 1e6, 1)}), flush=True)


if __name__ == '__main__':
    main()
