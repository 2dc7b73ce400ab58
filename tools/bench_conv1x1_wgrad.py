#!/usr/bin/env python3
"""lec_conv1x1_wgrad (MFMA weight gradient, float atomics into the gradient buffer) against the library's weight-gradient
convolution (+ its fp32 cast and copy), correctness and time, at ResNet-50's wide 1x1 shapes and the bench batch."""
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
    B = int(os.environ.get('LEC_B', 512))
    for ci, co, hw in ((64, 256, 56), (64, 64, 56), (128, 512, 28), (256, 64, 56), (256, 128, 56), (512, 128, 28), (256, 1024, 14), (1024, 256, 14)):
        M = B * hw * hw
        g = torch.Generator(device='cpu').manual_seed(ci + co)
        x = (torch.randn(M // 8, ci, generator=g) * 0.7).to('cuda').to(torch.bfloat16).repeat(8, 1)
        dy = (torch.randn(M // 8, co, generator=g) * 0.1).to('cuda').to(torch.bfloat16).repeat(8, 1)
        dw = torch.zeros(co, ci, device='cuda')
        ops.conv1x1_wgrad_rows(dy, x, dw)
        ref = 8 * (dy[:M // 8].float().double().t() @ x[:M // 8].float().double())
        err = ((dw.double() - ref).abs().max() / ref.abs().max()).item()
        x4 = x.view(B, hw, hw, ci).permute(0, 3, 1, 2); dy4 = dy.view(B, hw, hw, co).permute(0, 3, 1, 2)
        w4 = torch.zeros(co, ci, 1, 1, device='cuda', dtype=torch.bfloat16)
        slot = torch.zeros(co, ci, 1, 1, device='cuda')

        def lib():
            gw = torch.ops.aten.convolution_backward(dy4, x4, w4, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1, (False, True, False))[1]
            slot.copy_(gw)
        t_own = timed(lambda: ops.conv1x1_wgrad_rows(dy, x, dw)); t_lib = timed(lib)
        print(json.dumps({'cin': ci, 'cout': co, 'M': M, 'max_rel_err': err, 'own_us': round(t_own, 1), 'library_us': round(t_lib, 1),
                          'own_GBps': round(M * (ci + co) * 2 / t_own / 1e3, 1)}), flush=True)


if __name__ == '__main__':
    main()
