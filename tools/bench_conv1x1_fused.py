#!/usr/bin/env python3
"""lec_conv1x1_fwd_stats (MFMA 1x1 conv forward + BatchNorm statistics epilogue) against the pair it replaces
(MIOpen convolution + bn_stats pass), correctness and time, at ResNet-50's layer1 shapes and the bench batch."""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learning_embeddings_amd import miopen_tuning; miopen_tuning.setup()
from learning_embeddings_amd import ops, _lib
from learning_embeddings_amd._lib import lib, check, dptr, stream_ptr


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
    for ci, co, hw in ((64, 256, 56), (64, 64, 56), (128, 512, 28), (256, 64, 56), (256, 128, 56), (512, 128, 28)):
        M = B * hw * hw
        g = torch.Generator(device='cpu').manual_seed(ci + co)
        x = (torch.randn(M, ci, generator=g) * 0.7).to('cuda').to(torch.bfloat16)
        w = (torch.randn(co, ci, generator=g) * 0.2).to('cuda').to(torch.bfloat16)
        y = torch.empty(M, co, device='cuda', dtype=torch.bfloat16)
        ws = ops._bn_workspace(x.device)
        npart = C.c_int(0)

        def fused():
            check(lib.lec_conv1x1_fwd(dptr(x), dptr(w), 0, M, ci, co, dptr(y), dptr(ws), ws.numel(), C.byref(npart), stream_ptr()))
        fused(); torch.cuda.synchronize()
        part = ws[:npart.value * 2 * co * 4].view(torch.float32).view(npart.value, 2, co).double().sum(0).cpu()
        ref = (x.float() @ w.float().t())
        yb = ref.to(torch.bfloat16)
        err = (y.float() - ref).abs().max().item() / ref.abs().max().item()
        mism = (y != yb).float().mean().item()
        s_ref = yb.float().double().sum(0).cpu(); q_ref = (yb.float().double() ** 2).sum(0).cpu()
        s_err = ((part[0] - s_ref).abs().max() / (s_ref.abs().max() + 1e-9)).item(); q_err = ((part[1] - q_ref).abs().max() / q_ref.abs().max()).item()
        x4 = x.view(B, hw, hw, ci).permute(0, 3, 1, 2); w4 = w.view(co, ci, 1, 1)
        t_f = timed(fused)
        t_conv = timed(lambda: torch.nn.functional.conv2d(x4, w4))
        y4 = torch.nn.functional.conv2d(x4, w4)
        rm = torch.zeros(co, device='cuda'); rv = torch.ones(co, device='cuda'); gam = torch.ones(co, device='cuda'); bet = torch.zeros(co, device='cuda')
        sm = torch.empty(co, device='cuda'); si = torch.empty(co, device='cuda'); yo = torch.empty_like(y4)

        def bn_full():
            check(lib.lec_bn_fwd(dptr(y4), None, M, co, dptr(gam), dptr(bet), 1e-5, 0.1, dptr(rm), dptr(rv), 1, dptr(sm), dptr(si), dptr(yo), 1, None, dptr(ws), ws.numel(), stream_ptr()))

        def bn_pre():
            check(lib.lec_bn_fwd_prestat(dptr(y4), None, M, co, dptr(gam), dptr(bet), 1e-5, 0.1, dptr(rm), dptr(rv), npart.value, dptr(sm), dptr(si), dptr(yo), 1, None, dptr(ws), ws.numel(), stream_ptr()))
        t_bn = timed(bn_full)
        fused(); t_bnp = timed(bn_pre)
        print(json.dumps({'cin': ci, 'cout': co, 'M': M, 'max_rel_err': err, 'frac_not_bit_equal_to_rounded_fp32_matmul': mism,
                          'sum_rel_err': s_err, 'sumsq_rel_err': q_err, 'fused_us': round(t_f, 1), 'miopen_conv_us': round(t_conv, 1),
                          'bn_fwd_full_us': round(t_bn, 1), 'bn_fwd_prestat_us': round(t_bnp, 1),
                          'fused_GBps': round(M * (ci + co) * 2 / t_f / 1e3, 1), 'n_partials': npart.value}), flush=True)


if __name__ == '__main__':
    main()
