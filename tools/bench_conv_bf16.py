"""Per-layer timing of the bf16 convolution family (csrc/conv_bf16.hip) on the ResNet-50 layer shapes at `--rows` images, next to the library
kernels the 16-bit step used until round 5 (aten.convolution / convolution_backward = MIOpen / CK / hipBLASLt) and, where one exists, the round-1
special-case kernel (conv1x1 / conv3x3_c64).  Prints microseconds per launch and the algorithmic HBM rate (2 B x (input + output elements)).
    python tools/bench_conv_bf16.py [--rows 512] [--iters 20] [--no-lib] [--special] [--json out.json]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from learning_embeddings_amd import ops

# (name, Cin, H, Cout, R, stride, pad): every distinct convolution of ResNet-50 at 224 x 224
LAYERS = [
    ('stem 7x7/2', 8, 224, 64, 7, 2, 3),
    ('l1 1x1 64->64', 64, 56, 64, 1, 1, 0), ('l1 3x3 64', 64, 56, 64, 3, 1, 1), ('l1 1x1 64->256', 64, 56, 256, 1, 1, 0), ('l1 1x1 256->64', 256, 56, 64, 1, 1, 0),
    ('l2 1x1 256->128', 256, 56, 128, 1, 1, 0), ('l2 3x3/2 128', 128, 56, 128, 3, 2, 1), ('l2 1x1 128->512', 128, 28, 512, 1, 1, 0),
    ('l2 ds 1x1/2 256->512', 256, 56, 512, 1, 2, 0), ('l2 1x1 512->128', 512, 28, 128, 1, 1, 0), ('l2 3x3 128', 128, 28, 128, 3, 1, 1),
    ('l3 1x1 512->256', 512, 28, 256, 1, 1, 0), ('l3 3x3/2 256', 256, 28, 256, 3, 2, 1), ('l3 1x1 256->1024', 256, 14, 1024, 1, 1, 0),
    ('l3 ds 1x1/2 512->1024', 512, 28, 1024, 1, 2, 0), ('l3 1x1 1024->256', 1024, 14, 256, 1, 1, 0), ('l3 3x3 256', 256, 14, 256, 3, 1, 1),
    ('l4 1x1 1024->512', 1024, 14, 512, 1, 1, 0), ('l4 3x3/2 512', 512, 14, 512, 3, 2, 1), ('l4 1x1 512->2048', 512, 7, 2048, 1, 1, 0),
    ('l4 ds 1x1/2 1024->2048', 1024, 14, 2048, 1, 2, 0), ('l4 1x1 2048->512', 2048, 7, 512, 1, 1, 0), ('l4 3x3 512', 512, 7, 512, 3, 1, 1),
]
COUNT = {'stem 7x7/2': 1, 'l1 1x1 64->64': 1, 'l1 3x3 64': 3, 'l1 1x1 64->256': 4, 'l1 1x1 256->64': 2, 'l2 1x1 256->128': 1, 'l2 3x3/2 128': 1,
         'l2 1x1 128->512': 4, 'l2 ds 1x1/2 256->512': 1, 'l2 1x1 512->128': 3, 'l2 3x3 128': 3, 'l3 1x1 512->256': 1, 'l3 3x3/2 256': 1,
         'l3 1x1 256->1024': 6, 'l3 ds 1x1/2 512->1024': 1, 'l3 1x1 1024->256': 5, 'l3 3x3 256': 5, 'l4 1x1 1024->512': 1, 'l4 3x3/2 512': 1,
         'l4 1x1 512->2048': 3, 'l4 ds 1x1/2 1024->2048': 1, 'l4 1x1 2048->512': 2, 'l4 3x3 512': 2}


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000.0 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=512); ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--no-lib', action='store_true'); ap.add_argument('--special', action='store_true'); ap.add_argument('--json', default=None); ap.add_argument('--only', default=None)
    a = ap.parse_args()
    dev = 'cuda'
    rows = []
    tot = {'own_fwd': 0.0, 'own_dgrad': 0.0, 'own_wgrad': 0.0, 'lib_fwd': 0.0, 'lib_dgrad': 0.0, 'lib_wgrad': 0.0}
    print('%-24s %9s %9s %9s | %9s %9s %9s | %6s %6s %6s' % ('layer (x count)', 'fwd us', 'dgrad us', 'wgrad us', 'lib fwd', 'lib dgrad', 'lib wgrad', 'TB/s f', 'TB/s d', 'TB/s w'))
    for name, cin, hw, cout, r, st, pd in LAYERS:
        if a.only and a.only not in name:
            continue
        N = a.rows
        x = torch.randn(N, cin, hw, hw, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, r, r, device=dev) / (cin * r * r) ** 0.5).bfloat16().contiguous(memory_format=torch.channels_last)
        ho = (hw + 2 * pd - r) // st + 1
        dy = torch.randn(N, cout, ho, ho, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
        dw = torch.zeros(cout, cin, r, r, device=dev).contiguous(memory_format=torch.channels_last)
        if cin == 8:                                            # the stem as the step runs it: 3 real channels, its own forward kernel, a 3-channel gradient slot
            x[:, 3:] = 0; w[:, 3:] = 0
            dw = torch.zeros(cout, 3, r, r, device=dev).contiguous(memory_format=torch.channels_last)
        if cin == 8 and ops.conv_bf16_stem_supported(x):
            t_f = timeit(lambda: ops.conv_bf16_stem_fwd(x, w, want_stats=True), a.iters)
        else:
            t_f = timeit(lambda: ops.conv_bf16_fwd(x, w, st, pd, want_stats=True), a.iters)
        ops._BN_WS_OWNER[0] = 0
        t_d = float('nan')
        if cin != 8:
            wt = ops.conv_bf16_wt(w)
            t_d = timeit(lambda: ops.conv_bf16_dgrad(dy, wt, x.shape, st, pd), a.iters)
        t_w = timeit(lambda: ops.conv_bf16_wgrad(dy, x, dw, st, pd), a.iters)
        # the round-1 special-case kernels where they serve the shape (1x1 stride 1: conv1x1_*; 3x3 stride 1 at 64 / 128 channels: conv3x3_c64)
        s_f = s_d = s_w = float('nan')
        M = N * ho * ho
        if a.special and r == 1 and st == 1:
            xr = x.permute(0, 2, 3, 1).reshape(M, cin); dyr = dy.permute(0, 2, 3, 1).reshape(M, cout); w2 = w.view(cout, cin)
            if ops.conv1x1_supported(cin, cout, M):
                s_f = timeit(lambda: ops.conv1x1_rows(xr, w2, want_stats=True), a.iters); ops._BN_WS_OWNER[0] = 0
            if ops.conv1x1_supported(cout, cin, M):
                s_d = timeit(lambda: ops.conv1x1_rows(dyr, w2, w_transposed=True), a.iters)
            if ops.conv1x1_wgrad_supported(cin, cout, M):
                dw2 = torch.zeros(cout, cin, device=dev)
                s_w = timeit(lambda: ops.conv1x1_wgrad_rows(dyr, xr, dw2), a.iters)
        if a.special and r == 3 and st == 1 and cin == cout and cin in (64, 128):
            s_f = timeit(lambda: ops.conv3x3_c64(x, w, want_stats=True), a.iters); ops._BN_WS_OWNER[0] = 0
            s_d = timeit(lambda: ops.conv3x3_c64(dy, w, w_transposed=True), a.iters)
            if cin == 64:
                s_w = timeit(lambda: ops.conv3x3_c64_wgrad(dy, x, dw), a.iters)
        if a.special:
            print('%-24s %9.1f %9.1f %9.1f   (special-case kernels)' % ('', s_f, s_d, s_w))
        l_f = l_d = l_w = float('nan')
        if not a.no_lib:
            l_f = timeit(lambda: torch.ops.aten.convolution(x, w, None, [st, st], [pd, pd], [1, 1], False, [0, 0], 1), a.iters)
            if cin != 8:
                l_d = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [st, st], [pd, pd], [1, 1], False, [0, 0], 1, [True, False, False]), a.iters)
            l_w = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [st, st], [pd, pd], [1, 1], False, [0, 0], 1, [False, True, False]), a.iters)
        bytes_io = 2.0 * (x.numel() + dy.numel())
        c = COUNT[name]
        print('%-24s %9.1f %9.1f %9.1f | %9.1f %9.1f %9.1f | %6.2f %6.2f %6.2f' % ('%s x%d' % (name, c), t_f, t_d, t_w, l_f, l_d, l_w,
                                                                                bytes_io / t_f / 1e6, bytes_io / t_d / 1e6, bytes_io / t_w / 1e6))
        for k, v in (('own_fwd', t_f), ('own_dgrad', t_d), ('own_wgrad', t_w), ('lib_fwd', l_f), ('lib_dgrad', l_d), ('lib_wgrad', l_w)):
            if v == v:
                tot[k] += c * v
        rows.append({'layer': name, 'count': c, 'own_us': [t_f, t_d, t_w], 'special_us': [s_f, s_d, s_w], 'lib_us': [l_f, l_d, l_w], 'io_bytes': bytes_io})
        del x, w, dy, dw
    print('network totals (ms, layer counts applied): ' + ', '.join('%s %.2f' % (k, v / 1000.0) for k, v in tot.items()))
    if a.json:
        json.dump({'rows': a.rows, 'layers': rows, 'totals_ms': {k: v / 1000.0 for k, v in tot.items()}}, open(a.json, 'w'), indent=1)


if __name__ == '__main__':
    main()
