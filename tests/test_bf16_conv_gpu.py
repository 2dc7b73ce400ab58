"""The bf16 convolution family (csrc/conv_bf16.hip: config 5's 16-bit conv stack, oe_h.py:331-351 -> torchvision resnet50) against plain fp32
PyTorch on bf16-rounded operands: every ResNet-50 layer shape, stride and direction.  Small-integer operands make every product and partial sum
exactly representable (bf16 holds integers up to 256, fp32 accumulation up to 2^24), so forward and gradients must EQUAL the reference whatever the
summation order (outputs: that exact sum rounded once to bf16) -- a wrong operand layout, tap, parity class or transposed read is a mismatch, not noise."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from learning_embeddings_amd import ops  # noqa: E402

DEV = 'cuda'


def _cl(t):
    return t.to(DEV).contiguous(memory_format=torch.channels_last)


# (N, Cin, H, W, Cout, R, stride, pad): the layer shapes of ResNet-50 / -18 at small pixel counts + ragged sizes (row tails, channel tails of a tile)
BF16_CASES = [
    (2, 64, 8, 8, 64, 1, 1, 0), (3, 64, 7, 5, 256, 1, 1, 0), (2, 256, 6, 6, 64, 1, 1, 0), (2, 256, 8, 8, 128, 1, 1, 0), (2, 512, 4, 4, 128, 1, 1, 0),
    (2, 1024, 5, 5, 256, 1, 1, 0), (2, 256, 7, 7, 1024, 1, 1, 0), (2, 2048, 3, 3, 512, 1, 1, 0), (3, 512, 4, 4, 2048, 1, 1, 0), (2, 1024, 4, 4, 512, 1, 1, 0),
    (2, 64, 8, 8, 64, 3, 1, 1), (3, 128, 9, 7, 128, 3, 1, 1), (2, 256, 6, 6, 256, 3, 1, 1), (2, 512, 7, 7, 512, 3, 1, 1), (33, 128, 12, 12, 128, 3, 1, 1),
    (3, 128, 12, 8, 128, 3, 2, 1), (2, 256, 10, 10, 256, 3, 2, 1), (2, 512, 14, 14, 512, 3, 2, 1), (2, 64, 9, 9, 128, 3, 2, 1),
    (2, 256, 8, 8, 512, 1, 2, 0), (2, 512, 6, 6, 1024, 1, 2, 0), (2, 1024, 4, 4, 2048, 1, 2, 0), (2, 64, 8, 8, 128, 1, 2, 0),
    (2, 8, 20, 20, 64, 7, 2, 3), (3, 8, 17, 13, 64, 7, 2, 3), (2, 8, 12, 12, 64, 3, 1, 1),
]


@pytest.mark.parametrize('N,Cin,H,W,Cout,R,stride,pad', BF16_CASES)
def test_conv_bf16_fwd_dgrad_wgrad_exact_on_integers_and_vs_fp32(N, Cin, H, W, Cout, R, stride, pad):
    g = torch.Generator(device='cpu').manual_seed(Cin * 31 + Cout + R)
    for kind in ('int', 'rand'):
        if kind == 'int':
            x = torch.randint(-3, 4, (N, Cin, H, W), generator=g).float(); w = torch.randint(-2, 3, (Cout, Cin, R, R), generator=g).float()
        else:
            x = torch.randn(N, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, R, R, generator=g) / (Cin * R * R) ** 0.5
        x = _cl(x.bfloat16()); w = _cl(w.bfloat16())
        xr = x.double().requires_grad_(True); wr = w.double().requires_grad_(True)
        yr = F.conv2d(xr, wr, None, stride, pad)
        dy = (torch.randint(-2, 3, yr.shape, generator=g).float() if kind == 'int' else torch.randn(yr.shape, generator=g))
        dy = _cl(dy.bfloat16())
        yr.backward(dy.double())
        y = ops.conv_bf16_fwd(x, w, stride, pad, want_stats=True)
        ws = ops._bn_workspace(x.device).view(torch.float32)
        k = ops._BN_WS_OWNER[1]; ops._BN_WS_OWNER[0] = 0
        part = ws[:k * 2 * Cout].view(k, 2, Cout).double().sum(0)
        y2 = ops.conv_bf16_fwd(x, w, stride, pad)
        assert torch.equal(y, y2), 'the statistics epilogue must not change the output'
        assert y.shape == yr.shape and y.is_contiguous(memory_format=torch.channels_last) and y.dtype == torch.bfloat16
        dw = torch.zeros_like(w, dtype=torch.float32)
        ops.conv_bf16_wgrad(dy, x, dw, stride, pad)
        dx = None
        if Cout % 64 == 0:
            wt = ops.conv_bf16_wt(w)
            assert torch.equal(wt.permute(1, 0, 2, 3), w), 'transposed weights'
            dx = ops.conv_bf16_dgrad(dy, wt, x.shape, stride, pad)
        if kind == 'int':
            # exact integer sums in fp32, ONE rounding to bf16 on the way out (round to nearest even, as torch's cast)
            assert torch.equal(y, yr.detach().float().bfloat16()), 'forward'
            if dx is not None:
                assert torch.equal(dx, xr.grad.float().bfloat16()), 'data gradient'
            assert torch.equal(dw.double(), wr.grad), 'weight gradient'
            yb = y.double()
            assert torch.allclose(part[0], yb.sum(dim=(0, 2, 3)), rtol=1e-5, atol=1e-3) and torch.allclose(part[1], (yb ** 2).sum(dim=(0, 2, 3)), rtol=1e-5, atol=1e-3)
        else:
            # outputs are rounded to bf16 (8 significant bits): half an ulp of the largest value + fp32 accumulation noise
            tolb = lambda ref: 2.0 ** -8 * ref.abs().max().item()
            assert (y.double() - yr.detach()).abs().max().item() <= tolb(yr.detach())
            if dx is not None:
                assert (dx.double() - xr.grad).abs().max().item() <= tolb(xr.grad)
            assert (dw.double() - wr.grad).abs().max().item() <= 2e-5 * wr.grad.abs().max().item() + 1e-4     # fp32 output
            yb = y.double()
            assert torch.allclose(part[0], yb.sum(dim=(0, 2, 3)), rtol=1e-4, atol=1e-2), 'statistics are those of the ROUNDED output'
            assert torch.allclose(part[1], (yb ** 2).sum(dim=(0, 2, 3)), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize('N,H,W', [(3, 64, 64), (2, 128, 128), (2, 224, 224), (5, 62, 64)])
def test_conv_bf16_stem_forward_kernel_matches_the_three_channel_convolution(N, H, W):
    """lec_conv_bf16_stem_fwd (the stem's own kernel: image widths 64 / 128 / 224, even heights) against F.conv2d on the 3 real channels: exact on small integers
    (one rounding to bf16 on the way out), the statistics partials those of the rounded output, and identical to the generic kernel on the zero-padded operands.
    Channels 4..7 are never read: garbage there must not matter."""
    g = torch.Generator(device='cpu').manual_seed(H + W)
    x3 = torch.randint(-3, 4, (N, 3, H, W), generator=g).float(); w3 = torch.randint(-2, 3, (64, 3, 7, 7), generator=g).float()
    x8 = torch.zeros(N, 8, H, W); x8[:, :3] = x3; w8 = torch.zeros(64, 8, 7, 7); w8[:, :3] = w3
    xg = x8.clone(); xg[:, 4:] = torch.randn(N, 4, H, W, generator=g) * 100.0                  # channels 4..7: never read
    x8 = _cl(x8.bfloat16()); w8 = _cl(w8.bfloat16()); xg = _cl(xg.bfloat16())
    assert ops.conv_bf16_stem_supported(x8)
    yr = F.conv2d(x3.double(), w3.double(), None, 2, 3)
    y = ops.conv_bf16_stem_fwd(xg, w8, want_stats=True)
    k = ops._BN_WS_OWNER[1]; ops._BN_WS_OWNER[0] = 0
    part = ops._bn_workspace(x8.device).view(torch.float32)[:k * 2 * 64].view(k, 2, 64).double().sum(0)
    assert y.shape == yr.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(y.cpu(), yr.float().bfloat16())
    yb = y.double()
    assert torch.allclose(part[0], yb.sum(dim=(0, 2, 3)), rtol=1e-5, atol=1e-2) and torch.allclose(part[1], (yb ** 2).sum(dim=(0, 2, 3)), rtol=1e-5, atol=1e-1)
    assert torch.equal(y, ops.conv_bf16_fwd(x8, w8, 2, 3)) and torch.equal(y, ops.conv_bf16_stem_fwd(xg, w8))
    # random data: the same fp32 sums as the generic kernel up to summation order
    xr = torch.zeros(N, 8, H, W); xr[:, :3] = torch.randn(N, 3, H, W, generator=g); wr = torch.zeros(64, 8, 7, 7); wr[:, :3] = torch.randn(64, 3, 7, 7, generator=g) / 12.0
    xr = _cl(xr.bfloat16()); wr = _cl(wr.bfloat16())
    ya = ops.conv_bf16_stem_fwd(xr, wr).float(); yb2 = ops.conv_bf16_fwd(xr, wr, 2, 3).float()
    assert (ya - yb2).abs().max().item() <= 2.0 ** -7 * yb2.abs().max().item()
    # the weight gradient of the same layer (lec_conv_bf16_wgrad hands these sizes to the stem's own kernel): exact integer sums in fp32, added to what is there
    dyi = torch.randint(-2, 3, (N, 64, H // 2, W // 2), generator=g).float()
    x3r = x3.double().requires_grad_(False); w3r = w3.double().requires_grad_(True)
    F.conv2d(x3r, w3r, None, 2, 3).backward(dyi.double())
    dw = torch.ones(64, 3, 7, 7, device=DEV).contiguous(memory_format=torch.channels_last)
    ops.conv_bf16_wgrad(_cl(dyi.bfloat16()), xg, dw, 2, 3)
    assert torch.equal(dw.double().cpu(), w3r.grad + 1.0)


def test_conv_bf16_stem_three_channel_weight_gradient_slot():
    """The stem: x carries zero channels 3..7 and the gradient slot has 3 channels."""
    g = torch.Generator(device='cpu').manual_seed(5)
    x3 = torch.randint(-3, 4, (2, 3, 18, 18), generator=g).float()
    x8 = torch.zeros(2, 8, 18, 18); x8[:, :3] = x3
    dy = torch.randint(-2, 3, (2, 64, 9, 9), generator=g).float()
    xr = x3.double(); wr = torch.zeros(64, 3, 7, 7, dtype=torch.double, requires_grad=True)
    F.conv2d(xr, wr, None, 2, 3).backward(dy.double())
    dw = torch.zeros(64, 3, 7, 7, device=DEV).contiguous(memory_format=torch.channels_last)
    ops.conv_bf16_wgrad(_cl(dy.bfloat16()), _cl(x8.bfloat16()), dw, 2, 3)
    assert torch.equal(dw.double().cpu(), wr.grad)


@pytest.mark.parametrize('N,C,H,W,Cout,R,relu,res', [(2, 128, 8, 8, 128, 3, True, True), (3, 256, 5, 7, 64, 1, True, False), (2, 64, 8, 8, 64, 3, True, True),
                                                    (2, 512, 4, 4, 128, 1, False, True), (2, 64, 6, 6, 256, 1, True, True), (33, 128, 12, 12, 128, 3, True, False),
                                                    (5, 256, 9, 9, 1024, 1, True, True)])      # 256 destination channels, 1x1: the 128 x 256 eight-wave tile
def test_conv_bf16_dgrad_fold_is_pass1_of_the_batchnorm_backward(N, C, H, W, Cout, R, relu, res):
    """lec_conv_bf16_dgrad with the fold: g = mask * (dx + dres) rounded to bf16 and the partial sums (sum g, sum g xhat) -- what lec_bn_bwd_pass1 computes
    from the unfused data gradient."""
    g = torch.Generator(device='cpu').manual_seed(C + Cout + R)
    pad = R // 2
    xbn = _cl((torch.randn(N, C, H, W, generator=g)).bfloat16())                        # the BatchNorm's input
    mean = xbn.float().mean(dim=(0, 2, 3)); invstd = 1.0 / (xbn.float().var(dim=(0, 2, 3), unbiased=False) + 1e-5).sqrt()
    bits = (torch.rand(N, C, H, W, generator=g) > 0.4).to(DEV)
    mask = None
    if relu:
        b = bits.permute(0, 2, 3, 1).reshape(-1, C // 8, 8).to(torch.uint8)
        mask = (b * (2 ** torch.arange(8, device=DEV, dtype=torch.uint8))).sum(-1).to(torch.uint8).reshape(-1).contiguous()
    dres = _cl(torch.randn(N, C, H, W, generator=g).bfloat16()) if res else None
    dy = _cl(torch.randn(N, Cout, H, W, generator=g).bfloat16())
    w = _cl((torch.randn(Cout, C, R, R, generator=g) / (C * R * R) ** 0.5).bfloat16())
    wt = ops.conv_bf16_wt(w)
    dx = ops.conv_bf16_dgrad(dy, wt, xbn.shape, 1, pad)
    rec = {'x': xbn, 'mask': mask, 'mean': mean, 'invstd': invstd, 'dres': dres}
    gk = ops.conv_bf16_dgrad(dy, wt, xbn.shape, 1, pad, fold=rec)
    k = ops._BN_WS_OWNER[1]; ops._BN_WS_OWNER[0] = 0; ops.fusion().folded.clear()
    part = ops._bn_workspace(xbn.device).view(torch.float32)[:k * 2 * C].view(k, 2, C).double().sum(0)
    gr = dx.float() + (dres.float() if res else 0.0)
    if relu:
        gr = torch.where(bits, gr, torch.zeros_like(gr))
    gr = gr.bfloat16()
    assert torch.equal(gk, gr), 'g'
    xhat = (xbn.float() - mean.view(1, C, 1, 1)) * invstd.view(1, C, 1, 1)
    s1 = gr.double().sum(dim=(0, 2, 3)); s2 = (gr.float() * xhat).double().sum(dim=(0, 2, 3))
    assert torch.allclose(part[0], s1, rtol=1e-4, atol=1e-2) and torch.allclose(part[1], s2, rtol=1e-4, atol=2e-2)


def test_conv_bf16_full_size_layers_match_fp32_reference_on_rounded_operands():
    """Config 5's chunk size: 3x3 128 -> 128 @28 and the stride-2 256 -> 512 downsample at 64 rows, against fp32 F.conv2d on the GPU."""
    g = torch.Generator(device='cpu').manual_seed(11)
    for (N, Cin, H, W, Cout, R, stride, pad) in [(64, 128, 28, 28, 128, 3, 1, 1), (64, 256, 56, 56, 512, 1, 2, 0), (32, 128, 56, 56, 128, 3, 2, 1)]:
        x = _cl(torch.randn(N, Cin, H, W, generator=g).bfloat16()); w = _cl((torch.randn(Cout, Cin, R, R, generator=g) / (Cin * R * R) ** 0.5).bfloat16())
        xr = x.float().requires_grad_(True); wr = w.float().requires_grad_(True)
        yr = F.conv2d(xr, wr, None, stride, pad)
        dy = _cl(torch.randn(yr.shape, generator=g).bfloat16())
        yr.backward(dy.float())
        y = ops.conv_bf16_fwd(x, w, stride, pad)
        dx = ops.conv_bf16_dgrad(dy, ops.conv_bf16_wt(w), x.shape, stride, pad)
        dw = torch.zeros_like(w, dtype=torch.float32); ops.conv_bf16_wgrad(dy, x, dw, stride, pad)
        assert (y.float() - yr.detach()).abs().max().item() <= 2.0 ** -8 * yr.abs().max().item()
        assert (dx.float() - xr.grad).abs().max().item() <= 2.0 ** -8 * xr.grad.abs().max().item()
        assert (dw - wr.grad).abs().max().item() <= 1e-3 * wr.grad.abs().max().item()


def test_conv_bf16_sixteen_wave_tiles_statistics_and_fold_at_the_size_the_default_rule_picks_them():
    """Layer3's 3x3 (256 -> 256 @14 x 14) at 212 rows = 41 552 pixels = 163 m-tiles of 256: K = 2 304 >= 1 024 and >= 160 tiles, so forward and data gradient run the 256 x 256
    sixteen-wave kernel (one partial row per m-TILE, fold operands fetched after the K loop).  Exact on small integers; the statistics partials, the fold's g and its sums
    against the unfused forms."""
    N, C, H = 212, 256, 14
    g = torch.Generator(device='cpu').manual_seed(5)
    x = _cl(torch.randint(-2, 3, (N, C, H, H), generator=g).float().bfloat16()); w = _cl(torch.randint(-1, 2, (C, C, 3, 3), generator=g).float().bfloat16())
    yr = F.conv2d(x.float(), w.float(), None, 1, 1)
    y = ops.conv_bf16_fwd(x, w, 1, 1, want_stats=True)
    k = ops._BN_WS_OWNER[1]; ops._BN_WS_OWNER[0] = 0
    assert k == (N * H * H + 255) // 256, 'one partial row per 256-pixel tile: the sixteen-wave kernel ran'
    part = ops._bn_workspace(x.device).view(torch.float32)[:k * 2 * C].view(k, 2, C).double().sum(0)
    assert torch.equal(y, yr.bfloat16())
    yb = y.double()
    assert torch.allclose(part[0], yb.sum(dim=(0, 2, 3)), rtol=1e-5, atol=1e-2) and torch.allclose(part[1], (yb ** 2).sum(dim=(0, 2, 3)), rtol=1e-5, atol=1e-2)
    # data gradient, plain and folded (mask + residual)
    dy = _cl(torch.randint(-2, 3, (N, C, H, H), generator=g).float().bfloat16())
    wt = ops.conv_bf16_wt(w)
    dxr = torch.nn.grad.conv2d_input(x.shape, w.float(), dy.float(), 1, 1)
    dx = ops.conv_bf16_dgrad(dy, wt, x.shape, 1, 1)
    assert torch.equal(dx, dxr.bfloat16())
    xbn = _cl(torch.randn(N, C, H, H, generator=g).bfloat16())
    mean = xbn.float().mean(dim=(0, 2, 3)); invstd = 1.0 / (xbn.float().var(dim=(0, 2, 3), unbiased=False) + 1e-5).sqrt()
    bits = (torch.rand(N, C, H, H, generator=g) > 0.4).to(DEV)
    b = bits.permute(0, 2, 3, 1).reshape(-1, C // 8, 8).to(torch.uint8)
    mask = (b * (2 ** torch.arange(8, device=DEV, dtype=torch.uint8))).sum(-1).to(torch.uint8).reshape(-1).contiguous()
    dres = _cl(torch.randint(-2, 3, (N, C, H, H), generator=g).float().bfloat16())
    gk = ops.conv_bf16_dgrad(dy, wt, x.shape, 1, 1, fold={'x': xbn, 'mask': mask, 'mean': mean, 'invstd': invstd, 'dres': dres})
    k = ops._BN_WS_OWNER[1]; ops._BN_WS_OWNER[0] = 0; ops.fusion().folded.clear()
    assert k == (N * H * H + 255) // 256
    part = ops._bn_workspace(x.device).view(torch.float32)[:k * 2 * C].view(k, 2, C).double().sum(0)
    gr = torch.where(bits, dx.float() + dres.float(), torch.zeros_like(dxr)).bfloat16()
    assert torch.equal(gk, gr)
    xhat = (xbn.float() - mean.view(1, C, 1, 1)) * invstd.view(1, C, 1, 1)
    assert torch.allclose(part[0], gr.double().sum(dim=(0, 2, 3)), rtol=1e-4, atol=1e-2)
    assert torch.allclose(part[1], (gr.float() * xhat).double().sum(dim=(0, 2, 3)), rtol=1e-4, atol=5e-2)


def test_resnet50_bf16_family_everywhere_vs_special_cases_vs_library():
    """A bottleneck ResNet with every layer kind of ResNet-50 (stem, four stages with their strided and downsample layers) at bf16 (autocast), forward + backward with a flat arena, three ways on the same weights and batch: the default routing (round-1 special cases
    where they exist, the family elsewhere), the family for EVERY layer (LEC_CONV_BF16=2: stem, strided layers, parity classes, fold epilogues, transposed weights -- all in
    one network) and the library for everything the special cases do not serve (LEC_CONV_BF16=0: MIOpen / hipBLASLt).  Outputs and every parameter gradient agree to bf16
    noise; a wrong layout anywhere shows up as a cosine near zero in every layer upstream of it.  The own routings launch no library convolution."""
    from learning_embeddings_amd import resnet as R, parallel
    torch.manual_seed(0)
    x0 = torch.rand(16, 3, 128, 128).to(DEV).contiguous(memory_format=torch.channels_last)
    gsave = None; res = {}
    state = None
    for mode in ('1', '2', '0'):
        torch.manual_seed(1)
        net = R.ResNet(R.Bottleneck, [1, 1, 1, 1], 10).to(DEV).to(memory_format=torch.channels_last).train()   # every layer KIND of ResNet-50 once (a 16-block random-init net in
        if state is None:                                                                                   # bf16 is chaotic: any two correct implementations decorrelate)
            state = {k: v.clone() for k, v in net.state_dict().items()}
        net.load_state_dict(state)
        arena = parallel.FlatArena(net.parameters(), DEV); arena.enable_lowp_transposed()
        net.wgrad_overlap = R.WgradOverlap(None, arena, side_stream=False)
        old = R.CONV_BF16; R.CONV_BF16 = mode
        try:
            R.library_launches(reset=True)
            arena.zero_grad()
            with torch.autocast('cuda', dtype=torch.bfloat16):
                y = net(x0)
            if gsave is None:
                gsave = torch.randn_like(y.float())
            y.float().backward(gsave)
            torch.cuda.synchronize()
            libs = R.library_launches()
        finally:
            R.CONV_BF16 = old
        if mode != '0':
            assert sum(libs.values()) == 0, (mode, libs)
        else:
            assert sum(libs.values()) > 0
        res[mode] = (y.detach().float().clone(), [p.grad.float().clone() for p in net.parameters()])
    cos = lambda a, b: F.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0).item()
    stats = {}
    for a_, b_ in (('1', '2'), ('1', '0'), ('2', '0')):
        cs = [cos(a, b) for a, b in zip(res[a_][1], res[b_][1]) if a.numel() > 64]
        stats[(a_, b_)] = (cos(res[a_][0], res[b_][0]), min(cs), sum(cs) / len(cs))
    print('output / min / mean parameter-gradient cosines:', stats)
    # measured: the two own routings 0.99999 / 0.992 / 0.995 (output / min / mean); either against the library 0.9998 / 0.885 / 0.933 (another accumulation order in every layer)
    co, mn, me = stats[('1', '2')]
    assert co > 0.999 and mn > 0.97 and me > 0.985, stats
    for k in (('1', '0'), ('2', '0')):
        co, mn, me = stats[k]
        assert co > 0.99 and mn > 0.7 and me > 0.88, stats


def test_transposed_bf16_weight_arena_follows_the_adam_step():
    """FlatArena.enable_lowp_transposed: the data gradients' operand ([Cin][RS][Cout] per layer, one launch behind the Adam kernel) equals the bf16 shadow transposed, at
    construction and after optimizer steps, for every convolution weight of ResNet-18 (1x1 downsample, 3x3, strided; the 3-channel stem has no data gradient and no twin)."""
    from learning_embeddings_amd import resnet as R, parallel
    torch.manual_seed(0)
    net = R.resnet18(10).to(DEV).to(memory_format=torch.channels_last).train()
    arena = parallel.FlatArena(net.parameters(), DEV); arena.enable_lowp_transposed()
    convs = [m for m in net.modules() if isinstance(m, torch.nn.Conv2d)]
    def check():
        n = 0
        for c in convs:
            wt = arena.lowp_t_view(c.weight); w = arena.lowp_view(c.weight)
            if c.in_channels % 8 or c.out_channels % 64:
                assert wt is None
                continue
            assert wt is not None and tuple(wt.shape) == (c.in_channels, c.out_channels) + tuple(c.kernel_size)
            assert torch.equal(wt.permute(1, 0, 2, 3), w), c
            assert torch.equal(w.float(), c.weight.data.bfloat16().float())
            n += 1
        assert n >= 19
    check()
    for step in range(2):
        arena.grad.normal_()
        arena.adam_step(1e-2)
        check()
