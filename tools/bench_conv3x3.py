#!/usr/bin/env python3
"""lec_conv3x3_c64_fwd against MIOpen (forward and data gradient of ResNet-50 layer1's conv2 at the bench batch)."""
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
    for Cc, H in ((64, 56), (128, 28)):
        run(Cc, H)


def run(Cc, H):
    B = int(os.environ.get('LEC_B', 512))
    g = torch.Generator(device='cpu').manual_seed(0)
    x = (torch.randn(B, Cc, H, H, generator=g) * 0.7).to('cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cc, Cc, 3, 3, generator=g) * 0.05).to('cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y = ops.conv3x3_c64(x, w, want_stats=True)
    n = ops._BN_WS_OWNER[1]; ops._BN_WS_OWNER[0] = 0
    part = ops._bn_workspace(x.device)[:n * 2 * Cc * 4].view(torch.float32).view(n, 2, Cc).double().sum(0)
    ref = torch.nn.functional.conv2d(x[:8].float(), w.float(), padding=1)
    err = (y[:8].float() - ref).abs().max().item() / ref.abs().max().item()
    yd = y.float().double()
    s_err = ((part[0] - yd.sum((0, 2, 3))).abs().max() / yd.sum((0, 2, 3)).abs().max()).item()
    t_own = timed(lambda: ops.conv3x3_c64(x, w, want_stats=True)); ops._BN_WS_OWNER[0] = 0
    t_mi = timed(lambda: torch.nn.functional.conv2d(x, w, padding=1))
    cb = torch.ops.aten.convolution_backward
    dy = torch.randn_like(x)
    t_mi_d = timed(lambda: cb(dy, x, w, [0], [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False]))
    gx = ops.conv3x3_c64(dy, w, w_transposed=True)
    gref = cb(dy[:4].float(), x[:4].float(), w.float(), [0], [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]
    derr = (gx[:4].float() - gref).abs().max().item() / gref.abs().max().item()
    t_own_d = timed(lambda: ops.conv3x3_c64(dy, w, w_transposed=True))
    print(json.dumps({'C': Cc, 'H': H, 'fwd_rel_err': err, 'stats_rel_err': s_err, 'dgrad_rel_err': derr, 'own_fwd_us': round(t_own, 1), 'miopen_fwd_us': round(t_mi, 1),
                      'own_dgrad_us': round(t_own_d, 1), 'miopen_dgrad_us': round(t_mi_d, 1),
                      'own_fwd_TFLOPs': round(2 * B * H * H * Cc * Cc * 9 / t_own / 1e6, 1), 'own_fwd_GBps': round(B * H * H * Cc * 2 * 2 / t_own / 1e3, 1)}))


if __name__ == '__main__':
    main()
