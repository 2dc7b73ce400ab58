"""The backbone's kernels at the REFERENCE's precision (fp32 activations: oe_h.py:281-328 runs torchvision's ResNet in fp32,
no AMP): fused BatchNorm(+add)(+ReLU), max pooling and the whole ResNet through them, against plain PyTorch fp32 ops."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from learning_embeddings_amd import ops, _lib  # noqa: E402
from learning_embeddings_amd.resnet import resnet18, resnet50, BatchNormAct2d, WgradOverlap  # noqa: E402

DEV = 'cuda'


def _cl(t):
    return t.to(DEV).contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize('N,C,H,W', [(4, 64, 14, 14), (2, 2048, 7, 7), (8, 256, 9, 5), (3, 24, 5, 5), (16, 64, 56, 56), (5, 1024, 3, 3)])
@pytest.mark.parametrize('res,relu', [(False, True), (True, True), (False, False), (True, False)])
def test_bn_f32_fwd_bwd_vs_torch(N, C, H, W, res, relu):
    g = torch.Generator(device='cpu').manual_seed(N * 1000 + C)
    x = _cl(torch.randn(N, C, H, W, generator=g) * 1.5 + 0.3)
    r = _cl(torch.randn(N, C, H, W, generator=g)) if res else None
    w = (torch.rand(C, generator=g) + 0.5).to(DEV); b = (torch.randn(C, generator=g) * 0.2).to(DEV)
    dy = _cl(torch.randn(N, C, H, W, generator=g))
    rm = torch.zeros(C, device=DEV); rv = torch.ones(C, device=DEV); rm2 = rm.clone(); rv2 = rv.clone()
    xa = x.clone().requires_grad_(True); ra = r.clone().requires_grad_(True) if res else None
    wa = w.clone().requires_grad_(True); ba = b.clone().requires_grad_(True)
    y = ops.BNActFn.apply(xa, ra, wa, ba, rm, rv, True, 0.1, 1e-5, relu)
    y.backward(dy)
    # float64 reference of the same op
    xb = x.double().requires_grad_(True); rb = r.double().requires_grad_(True) if res else None
    wb = w.double().requires_grad_(True); bb = b.double().requires_grad_(True)
    yr = F.batch_norm(xb, rm2.double(), rv2.double(), wb, bb, True, 0.1, 1e-5)
    if res:
        yr = yr + rb
    if relu:
        yr = F.relu(yr)
    yr.backward(dy.double())
    assert y.dtype == torch.float32 and y.is_contiguous(memory_format=torch.channels_last)
    # tolerance: fp32 rounding of a normalised value (a few ulp at |y| ~ 10)
    assert (y.double() - yr).abs().max().item() <= 2e-5 * (1 + yr.abs().max().item())
    m64 = x.double().mean(dim=(0, 2, 3)); v64 = x.double().var(dim=(0, 2, 3), unbiased=True)
    assert torch.allclose(rm.double(), 0.1 * m64, atol=1e-6, rtol=1e-5) and torch.allclose(rv.double(), 0.9 + 0.1 * v64, atol=1e-6, rtol=1e-5)
    # elements whose |y| sits within fp32 rounding of the ReLU kink may flip mask: compare robustly
    scale = xb.grad.abs().max().item() + 1e-12
    bad = ((xa.grad.double() - xb.grad).abs() > 2e-4 * scale).float().mean().item()
    assert bad < 1e-4, bad
    # d gamma / d beta: a mask flip at the kink moves one channel's sum by one element's worth; allow a couple of such channels
    for a_, b_ in ((wa.grad.double(), wb.grad), (ba.grad.double(), bb.grad)):
        off = (a_ - b_).abs() > 1e-4 * b_.abs().max().item() + 1e-4 * b_.abs()
        assert int(off.sum()) <= 2 and (a_ - b_).abs().max().item() < 30.0, (int(off.sum()), (a_ - b_).abs().max().item())
    if res:
        assert ((ra.grad.double() - rb.grad).abs() > 1e-5 * (rb.grad.abs().max().item() + 1e-12)).float().mean().item() < 1e-4


def test_bn_f32_forked_output_two_gradient_streams():
    g = torch.Generator(device='cpu').manual_seed(5)
    N, C, H, W = 6, 128, 7, 9
    x = _cl(torch.randn(N, C, H, W, generator=g)); r = _cl(torch.randn(N, C, H, W, generator=g))
    w = (torch.rand(C, generator=g) + 0.5).to(DEV); b = torch.zeros(C, device=DEV)
    d1 = _cl(torch.randn(N, C, H, W, generator=g)); d2 = _cl(torch.randn(N, C, H, W, generator=g))
    rm = torch.zeros(C, device=DEV); rv = torch.ones(C, device=DEV)
    xa = x.clone().requires_grad_(True); ra = r.clone().requires_grad_(True)
    ya, yb = ops.BNActFn.apply(xa, ra, w, b, rm, rv, True, 0.1, 1e-5, True, True)
    torch.autograd.backward([ya, yb], [d1, d2])
    xb = x.clone().requires_grad_(True); rb = r.clone().requires_grad_(True)
    yr = F.relu(F.batch_norm(xb, None, None, w, b, True, 0.1, 1e-5) + rb)
    yr.backward(d1 + d2)
    assert torch.allclose(xa.grad, xb.grad, rtol=1e-4, atol=1e-5 * xb.grad.abs().max().item())
    assert torch.allclose(ra.grad, rb.grad, rtol=1e-5, atol=1e-6)


def test_bn_f32_eval_mode_uses_running_stats():
    C = 64
    x = _cl(torch.randn(4, C, 8, 8))
    m = BatchNormAct2d(C, relu=True).to(DEV)
    with torch.no_grad():
        m.running_mean.normal_(); m.running_var.uniform_(0.5, 2); m.weight.uniform_(0.5, 1.5); m.bias.normal_()
    m.eval()
    y = m(x)
    yr = F.relu(F.batch_norm(x, m.running_mean, m.running_var, m.weight, m.bias, False, 0.1, m.eps))
    assert torch.allclose(y, yr, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('N,C,H,W', [(4, 64, 16, 16), (2, 8, 6, 10), (8, 64, 112, 112)])
def test_maxpool3x3s2_f32_vs_torch(N, C, H, W):
    x = _cl(torch.randn(N, C, H, W)).requires_grad_(True)
    y = ops.MaxPool3x3s2Fn.apply(x)
    dy = _cl(torch.randn_like(y))
    y.backward(dy)
    xr = x.detach().clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 3, 2, 1)
    yr.backward(dy)
    assert y.dtype == torch.float32 and torch.equal(y, yr)                 # the max is one of the inputs: exact
    assert torch.allclose(x.grad, xr.grad, rtol=1e-6, atol=1e-6)           # <= 4 window gradients summed per position


def test_resnet_f32_fused_path_matches_stock_torch():
    """Whole ResNet-18 in fp32: liblecone's BatchNorm / pooling kernels (and whichever convolution path is active) against
    stock torch ops on the same weights and batch.  Same precision on both sides, so the match is tight -- this is the
    precision the reference trains in."""
    torch.manual_seed(0)
    net = resnet18(num_classes=10).to(DEV).to(memory_format=torch.channels_last)
    x = _cl(torch.rand(16, 3, 64, 64))
    net.train()
    res = {}
    g = None
    for tag in ('fused', 'stock'):
        net.zero_grad()
        BatchNormAct2d.fused_enabled = tag == 'fused'
        try:
            y = net(x)
        finally:
            BatchNormAct2d.fused_enabled = True
        if g is None:
            g = torch.randn_like(y)
        y.backward(g)
        res[tag] = (y.detach().clone(), {n: p.grad.clone() for n, p in net.named_parameters()})
    ya, yb = res['fused'][0], res['stock'][0]
    assert (ya - yb).abs().max().item() <= 1e-4 * (1 + yb.abs().max().item())
    for n in res['stock'][1]:
        a, b = res['fused'][1][n].double().flatten(), res['stock'][1][n].double().flatten()
        cos = float(a @ b / (a.norm() * b.norm() + 1e-300))
        assert cos > 0.9999, (n, cos)


def test_bf16_step_drift_from_the_reference_precision_is_bounded():
    """The bf16 conv stack (MI355X-native storage, licensed by BASELINE.json's config 5 only) against the fp32 one (the
    reference's precision) on the same ResNet-50, same weights, same batch: drift of the raw CNN outputs and of the cone
    energies computed from them.  The bound is what a random-init network shows (the measured values are printed)."""
    from learning_embeddings_amd.oe_h import FeatCNN
    torch.manual_seed(0)
    D = 10
    f32 = FeatCNN(image_dir='', output_dim=D, K=0.1, compute_dtype=torch.float32).to(DEV)
    b16 = FeatCNN(image_dir='', output_dim=D, K=0.1, compute_dtype=torch.bfloat16).to(DEV)
    b16.load_state_dict(f32.state_dict())
    f32.train(); b16.train()
    x = _cl(torch.rand(32, 3, 224, 224))
    with torch.no_grad():
        ra = f32.forward_raw(x); rb = b16.forward_raw(x)
    rel = ((ra - rb).norm() / ra.norm()).item()
    cos = F.cosine_similarity(ra.flatten().double(), rb.flatten().double(), dim=0).item()
    # cone energies of (label, image) pairs from both sets of image points
    lab = torch.randn(32, D, device=DEV); lab = lab / lab.norm(dim=1, keepdim=True) * 0.3
    ea = ops.pair_energy(lab, f32.soft_clip(ra), 0.1); eb = ops.pair_energy(lab, b16.soft_clip(rb), 0.1)
    de = (ea - eb).abs().max().item()
    print('bf16 vs fp32 ResNet-50 (random init, batch 32): raw-output relative L2 drift %.4f, cosine %.5f, max |dE| %.4f' % (rel, cos, de))
    # measured on the MI355X (random-init ResNet-50, batch 32): relative drift 0.094, cosine 0.9956, max |dE| 0.063
    assert np.isfinite(rel) and rel < 0.2 and cos > 0.98
    assert de < 0.2


def test_split_mode_cone_energies_within_the_north_star_tolerance(monkeypatch):
    """north_star: cone energies within 1e-4 of the reference's fp32 path.  ResNet-50 (random init, batch 32, 224 x 224, training-mode
    BatchNorm) -> raw features -> cone energies, three times on the same weights and batch: liblecone's f32-MFMA convolutions (exact
    fp32), the split mode (fp32 products as six bf16 products), stock torch / MIOpen fp32.  The split mode must sit as close to the
    exact kernels as the two exact implementations sit to each other, and within 1e-4 on the energies."""
    from learning_embeddings_amd.oe_h import FeatCNN
    from learning_embeddings_amd import resnet as R
    torch.manual_seed(0)
    D = 10
    net = FeatCNN(image_dir='', output_dim=D, K=0.1, compute_dtype=torch.float32).to(DEV)
    net.train()
    x = _cl(torch.rand(32, 3, 224, 224))
    lab = torch.randn(32, D, device=DEV); lab = lab / lab.norm(dim=1, keepdim=True) * 0.3
    out = {}
    for tag in ('native', 'x3', 'stock'):
        monkeypatch.setattr(R, 'F32_MODE', 'x3' if tag == 'x3' else 'native')
        monkeypatch.setattr(R, 'MFMA_F32', tag != 'stock')          # (off: also the no-grad inference branch of Conv2d.forward takes the library)
        WgradOverlap.instance = WgradOverlap() if tag != 'stock' else None
        try:
            with torch.no_grad():
                raw = net.forward_raw(x)
            torch.cuda.synchronize()
        finally:
            WgradOverlap.instance = None
        out[tag] = (raw.clone(), ops.pair_energy(lab, net.soft_clip(raw), 0.1).clone())
    rel = lambda a, b: ((a - b).norm() / a.norm()).item()
    d_x3 = (out['native'][1] - out['x3'][1]).abs().max().item(); d_lib = (out['native'][1] - out['stock'][1]).abs().max().item()
    r_x3 = rel(out['native'][0], out['x3'][0]); r_lib = rel(out['native'][0], out['stock'][0])
    print('ResNet-50 raw outputs, relative L2 to the f32-MFMA path: split %.2e, MIOpen fp32 %.2e; max |dE|: split %.2e, MIOpen %.2e' % (r_x3, r_lib, d_x3, d_lib))
    assert d_x3 <= 1e-4
    assert r_x3 <= 3.0 * r_lib + 1e-6


# ---------------------------------------------------------------------------------------------------------------- fp32 convolutions
CONV_CASES = [  # N, Cin, H, W, Cout, R, stride, pad
    (2, 64, 12, 12, 64, 1, 1, 0), (2, 64, 12, 12, 256, 1, 1, 0), (3, 256, 9, 7, 64, 1, 1, 0), (2, 64, 10, 14, 64, 3, 1, 1),
    (2, 128, 12, 12, 128, 3, 2, 1), (2, 128, 11, 9, 128, 3, 2, 1), (2, 256, 8, 8, 512, 1, 2, 0), (1, 512, 7, 7, 512, 3, 1, 1),
    (2, 4, 32, 32, 64, 7, 2, 3), (1, 1024, 5, 5, 2048, 1, 2, 0), (5, 64, 3, 3, 128, 3, 1, 1), (2, 8, 6, 6, 16, 3, 1, 1),
    # the shifted-dense weight gradient (round 4: stride-1 'same' layers with Cin >= 128): ragged, non-square, 128 x 128 and 64 x 256 tiles, a 3 x 3 image (falls back)
    (3, 128, 9, 7, 128, 3, 1, 1), (2, 256, 10, 6, 64, 3, 1, 1), (2, 128, 5, 5, 256, 3, 1, 1), (7, 128, 3, 3, 128, 3, 1, 1), (33, 128, 28, 28, 128, 3, 1, 1),
    # ... its stride-2 form (input grid = 2 x output grid): 3x3 / pad 1 non-square, a downsample 1x1, 64 output channels (64 x 256 tile), one that falls back (Cin = 128, Cout = 64)
    (3, 256, 12, 8, 128, 3, 2, 1), (2, 512, 14, 14, 1024, 1, 2, 0), (2, 256, 12, 12, 64, 3, 2, 1), (2, 128, 16, 16, 64, 3, 2, 1), (9, 128, 28, 28, 128, 3, 2, 1),
]


@pytest.mark.parametrize('N,Cin,H,W,Cout,R,stride,pad', CONV_CASES)
def test_conv_f32_fwd_dgrad_wgrad_exact_on_integers_and_vs_float64(N, Cin, H, W, Cout, R, stride, pad):
    """lec_conv_f32_{fwd,dgrad,wgrad}: (a) small-integer operands -- every product and partial sum is exactly representable, so the
    result must EQUAL the float64 convolution, whatever the summation order (a wrong operand layout, tap or parity class shows up
    as a mismatch, not as noise); (b) random operands against float64 to fp32 accumulation noise."""
    g = torch.Generator(device='cpu').manual_seed(Cin * 31 + Cout + R)
    for kind in ('int', 'rand'):
        if kind == 'int':
            x = torch.randint(-3, 4, (N, Cin, H, W), generator=g).float(); w = torch.randint(-2, 3, (Cout, Cin, R, R), generator=g).float()
        else:
            x = torch.randn(N, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, R, R, generator=g) / (Cin * R * R) ** 0.5
        x = _cl(x); w = _cl(w)
        xr = x.double().requires_grad_(True); wr = w.double().requires_grad_(True)
        yr = F.conv2d(xr, wr, None, stride, pad)
        dy = (torch.randint(-2, 3, yr.shape, generator=g).float() if kind == 'int' else torch.randn(yr.shape, generator=g))
        dy = _cl(dy)
        yr.backward(dy.double())
        y = ops.conv_f32_fwd(x, w, stride, pad, want_stats=True)
        ws = ops._bn_workspace(x.device).view(torch.float32)
        k = ops._BN_WS_OWNER[1]; ops._BN_WS_OWNER[0] = 0
        part = ws[:k * 2 * Cout].view(k, 2, Cout).double().sum(0)
        dx = ops.conv_f32_dgrad(dy, w, x.shape, stride, pad)
        dw = torch.zeros_like(w)
        ops.conv_f32_wgrad(dy, x, dw, stride, pad)
        assert y.shape == yr.shape and y.is_contiguous(memory_format=torch.channels_last)
        if kind == 'int':
            assert torch.equal(y.double(), yr.detach()), 'forward'
            assert torch.equal(dx.double(), xr.grad), 'data gradient'
            assert torch.equal(dw.double(), wr.grad), 'weight gradient'
            assert torch.equal(part[0], yr.detach().sum(dim=(0, 2, 3))) and torch.equal(part[1], (yr.detach() ** 2).sum(dim=(0, 2, 3)))
        else:
            tol = lambda ref: 2e-5 * ref.abs().max().item()
            assert (y.double() - yr.detach()).abs().max().item() <= tol(yr.detach())
            assert (dx.double() - xr.grad).abs().max().item() <= tol(xr.grad)
            assert (dw.double() - wr.grad).abs().max().item() <= tol(wr.grad)
            assert torch.allclose(part[0], yr.detach().sum(dim=(0, 2, 3)), rtol=1e-4, atol=1e-3)


X3_CASES = CONV_CASES[:11] + [(2, 32, 6, 6, 32, 3, 1, 1), (3, 256, 10, 10, 128, 3, 1, 1), (2, 128, 16, 16, 256, 1, 1, 0)]


@pytest.mark.parametrize('N,H,W', [(3, 64, 64), (2, 128, 128), (2, 224, 224), (5, 62, 64)])
def test_conv_f32_stem_forward_kernel_matches_the_three_channel_convolution(N, H, W):
    """lec_conv_f32_stem_fwd (the fp32 stem's own kernel: image widths 64 / 128 / 224, even heights) against F.conv2d on the 3 real channels in float64: exact on
    small integers, fp32-grade on random data, the statistics partials (sum y, sum y^2) those of its output, and equal to the generic kernel up to summation order.
    Channel 3 is never multiplied: garbage there must not matter."""
    g = torch.Generator(device='cpu').manual_seed(H + W)
    x3 = torch.randint(-3, 4, (N, 3, H, W), generator=g).float(); w3 = torch.randint(-2, 3, (64, 3, 7, 7), generator=g).float()
    x4 = torch.zeros(N, 4, H, W); x4[:, :3] = x3; w4 = torch.zeros(64, 4, 7, 7); w4[:, :3] = w3
    xg = x4.clone(); xg[:, 3] = torch.randn(N, H, W, generator=g) * 100.0
    x4 = _cl(x4); w4 = _cl(w4); xg = _cl(xg)
    assert ops.conv_f32_stem_supported(x4)
    yr = F.conv2d(x3.double(), w3.double(), None, 2, 3)
    y = ops.conv_f32_stem_fwd(xg, w4, want_stats=True)
    k = ops._BN_WS_OWNER[1]; ops._BN_WS_OWNER[0] = 0
    part = ops._bn_workspace(x4.device).view(torch.float32)[:k * 2 * 64].view(k, 2, 64).double().sum(0)
    assert y.shape == yr.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(y.double().cpu(), yr)
    yb = y.double()
    assert torch.allclose(part[0], yb.sum(dim=(0, 2, 3)), rtol=1e-5, atol=1e-2) and torch.allclose(part[1], (yb ** 2).sum(dim=(0, 2, 3)), rtol=1e-5, atol=1e-1)
    assert torch.equal(y, ops.conv_f32_fwd(x4, w4, 2, 3)) and torch.equal(y, ops.conv_f32_stem_fwd(xg, w4))
    xr = torch.zeros(N, 4, H, W); xr[:, :3] = torch.rand(N, 3, H, W, generator=g); wr = torch.zeros(64, 4, 7, 7); wr[:, :3] = torch.randn(64, 3, 7, 7, generator=g) / 12.0
    yr2 = F.conv2d(xr[:, :3].double(), wr[:, :3].double(), None, 2, 3)
    xr = _cl(xr); wr = _cl(wr)
    ya = ops.conv_f32_stem_fwd(xr, wr).double().cpu()
    assert (ya - yr2).abs().max().item() <= 2e-6 * yr2.abs().max().item() + 1e-6
    assert torch.equal(ops.conv_f32_stem_fwd(xr, wr), ops.conv_f32_stem_fwd(xr, wr)), 'deterministic'
    # the weight gradient of the same layer (lec_conv_f32_wgrad_c3 hands these sizes to the stem's own kernel): exact integer sums, added to what is there
    dyi = torch.randint(-2, 3, (N, 64, H // 2, W // 2), generator=g).float()
    w3r = w3.double().requires_grad_(True)
    F.conv2d(x3.double(), w3r, None, 2, 3).backward(dyi.double())
    dw = torch.ones(64, 3, 7, 7, device=DEV).contiguous(memory_format=torch.channels_last)
    ops.conv_f32_wgrad_c3(_cl(dyi), xg, dw, 2, 3)
    assert torch.equal(dw.double().cpu(), w3r.grad + 1.0)
    dyr = torch.randn(N, 64, H // 2, W // 2, generator=g); wrr = wr.detach().cpu()[:, :3].double().requires_grad_(True)
    F.conv2d(xr.cpu()[:, :3].double(), wrr, None, 2, 3).backward(dyr.double())
    dw2 = torch.zeros(64, 3, 7, 7, device=DEV).contiguous(memory_format=torch.channels_last)
    ops.conv_f32_wgrad_c3(_cl(dyr), xr, dw2, 2, 3)
    assert (dw2.double().cpu() - wrr.grad).abs().max().item() <= 2e-5 * wrr.grad.abs().max().item() + 1e-5


@pytest.mark.parametrize('N,Cin,H,W,Cout,R,stride,pad', X3_CASES)
def test_conv_f32x3_exact_on_integers_and_fp32_grade_vs_float64(N, Cin, H, W, Cout, R, stride, pad):
    """lec_conv_f32x3_{fwd,dgrad,wgrad} (fp32 products as six bf16 products on the matrix cores).
    (a) 'int': operands that fit the leading bf16 piece -- the result must EQUAL the float64 convolution;
    (b) 'wide': 10-bit integers (two pieces per operand: the h*m, m*h and m*m products carry the answer) sized so that every partial
        sum stays below 2^24 -- again EQUAL, so a lost or misplaced piece is a mismatch, not noise;
    (c) 'rand': against float64, next to the exact-fp32 kernels of this library (f32-input MFMA) and of MIOpen on the same
        operands: the split kernels' error must lie within twice the larger of the two (measured: at or below the f32-MFMA
        kernel's for the forward and the data gradient, within 2x of it for the weight gradient), and under 5e-6 of the
        tensor's maximum in any case."""
    g = torch.Generator(device='cpu').manual_seed(Cin * 37 + Cout + R)
    stem = Cin < 32
    K = Cin * R * R
    for kind in ('int', 'wide', 'rand'):
        if kind == 'int':
            x = torch.randint(-3, 4, (N, Cin, H, W), generator=g).float(); w = torch.randint(-2, 3, (Cout, Cin, R, R), generator=g).float()
        elif kind == 'wide':
            bx = 511; bw = max(2, min(300, (1 << 24) // (bx * K) - 1))          # |sum| <= K * bx * bw < 2^24
            x = torch.randint(-bx, bx + 1, (N, Cin, H, W), generator=g).float(); w = torch.randint(-bw, bw + 1, (Cout, Cin, R, R), generator=g).float()
        else:
            x = torch.randn(N, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, R, R, generator=g) / K ** 0.5
        x = _cl(x); w = _cl(w)
        xr = x.double().requires_grad_(True); wr = w.double().requires_grad_(True)
        yr = F.conv2d(xr, wr, None, stride, pad)
        npix = yr.shape[0] * yr.shape[2] * yr.shape[3]
        if kind == 'int':
            dy = torch.randint(-2, 3, yr.shape, generator=g).float()
        elif kind == 'wide':
            bd = max(1, min(300, (1 << 24) // (511 * max(npix, Cout * R * R)) - 1))     # weight gradient: sums over pixels; data gradient: over Cout * taps
            dy = torch.randint(-bd, bd + 1, yr.shape, generator=g).float()
            if bd * 300 * Cout * R * R >= (1 << 24):                            # (the data gradient multiplies dy by w)
                dy = dy.clamp(-((1 << 24) // (300 * Cout * R * R)), (1 << 24) // (300 * Cout * R * R))
        else:
            dy = torch.randn(yr.shape, generator=g)
        dy = _cl(dy)
        yr.backward(dy.double())
        planes = ops.conv_f32x3_split_weights(w)
        y = ops.conv_f32x3_fwd(x, planes, stride, pad, want_stats=True)
        ws = ops._bn_workspace(x.device).view(torch.float32)
        k = ops._BN_WS_OWNER[1]; ops._BN_WS_OWNER[0] = 0
        part = ws[:k * 2 * Cout].view(k, 2, Cout).double().sum(0)
        dx = None if stem else ops.conv_f32x3_dgrad(dy, planes, x.shape, stride, pad)
        dw = None
        if ops.conv_f32x3_wgrad_supported(Cin, Cout, R, R):
            dw = torch.zeros_like(w); ops.conv_f32x3_wgrad(dy, x, dw, stride, pad)
        assert y.shape == yr.shape and y.is_contiguous(memory_format=torch.channels_last)
        if kind != 'rand':
            assert torch.equal(y.double(), yr.detach()), 'forward (%s)' % kind
            assert dx is None or torch.equal(dx.double(), xr.grad), 'data gradient (%s)' % kind
            assert dw is None or torch.equal(dw.double(), wr.grad), 'weight gradient (%s)' % kind
            if kind == 'int':
                assert torch.equal(part[0], yr.detach().sum(dim=(0, 2, 3))) and torch.equal(part[1], (yr.detach() ** 2).sum(dim=(0, 2, 3)))
        else:
            err = lambda t, ref: (t.double() - ref).abs().max().item() / ref.abs().max().item()
            cb = torch.ops.aten.convolution_backward
            # (yardstick: the exact kernels' error with the SAME summation structure -- whole-K chains, i.e. the tile walk; cut along K the exact
            # kernels sum shorter chains and err 2 - 3 x less on these small launches)
            with _Balanced(0):
                y1 = ops.conv_f32_fwd(x, w, stride, pad)
            yl = F.conv2d(x, w, None, stride, pad)
            e3, e1, el = err(y, yr.detach()), err(y1, yr.detach()), err(yl, yr.detach())
            assert e3 <= 2.0 * max(e1, el) + 1e-7 and e3 < 5e-6, ('forward', e3, e1, el)
            if dx is not None:
                with _Balanced(0):
                    d1 = ops.conv_f32_dgrad(dy, w, x.shape, stride, pad)
                dl = cb(dy, x, w, None, [stride] * 2, [pad] * 2, [1, 1], False, [0, 0], 1, [True, False, False])[0]
                e3, e1, el = err(dx, xr.grad), err(d1, xr.grad), err(dl, xr.grad)
                assert e3 <= 2.0 * max(e1, el) + 1e-7 and e3 < 5e-6, ('data gradient', e3, e1, el)
            if dw is not None:
                w1 = torch.zeros_like(w); ops.conv_f32_wgrad(dy, x, w1, stride, pad)
                wl = cb(dy, x, w, None, [stride] * 2, [pad] * 2, [1, 1], False, [0, 0], 1, [False, True, False])[1]
                e3, e1, el = err(dw, wr.grad), err(w1, wr.grad), err(wl, wr.grad)
                assert e3 <= 2.5 * max(e1, el) + 1e-7 and e3 < 5e-6, ('weight gradient', e3, e1, el)
            assert torch.allclose(part[0], yr.detach().sum(dim=(0, 2, 3)), rtol=1e-4, atol=1e-3)


def test_conv_f32x3_weight_pieces_add_up_to_the_weight_exactly():
    """lec_conv_f32x3_split_weights: in the tile-major planes, h + m + l of every element IS the fp32 weight (bit for bit), every
    piece a bf16, padding rows / k are zeros -- for both layouts."""
    g = torch.Generator(device='cpu').manual_seed(5)
    for (cout, cin, r) in [(64, 4, 7), (192 // 3 * 2, 32, 3), (256, 128, 1), (64, 64, 3)]:
        w = _cl(torch.randn(cout, cin, r, r, generator=g) * torch.logspace(-6, 3, cout).view(-1, 1, 1, 1))
        pl = ops.conv_f32x3_split_weights(w)
        wk = w.permute(0, 2, 3, 1).reshape(cout, r * r * cin)                                   # [co][tap * Cin + ci]
        wt = w.permute(1, 2, 3, 0).reshape(cin, r * r * cout)                                   # [ci][tap * Cout + co]
        for planes, mat in ((pl.fwd, wk), (pl.t, wt)):
            ncol, kdim = mat.shape
            bn = 64 if ncol <= 64 else 128
            nt, kc = (ncol + bn - 1) // bn, (kdim + 15) // 16
            p = (planes.view(nt, kc, 3, bn, 16).int() & 0xffff) << 16                           # bf16 bit patterns -> fp32 bit patterns
            f = p.view(torch.float32) if p.dtype == torch.float32 else p.to(torch.int32).view(torch.float32)
            tot = (f[:, :, 0] + f[:, :, 1]) + f[:, :, 2]                                         # [nt, kc, bn, 16]
            full = tot.permute(0, 2, 1, 3).reshape(nt * bn, kc * 16)
            assert torch.equal(full[:ncol, :kdim], mat), (cout, cin, r)
            assert not full[ncol:].any() and not full[:, kdim:].any()


def test_resnet_f32_split_mode_matches_exact_mode(monkeypatch):
    """A ResNet-50 forward + backward with LEC_CONV_F32_MODE=x3 (forward, data and weight gradients on the bf16 matrix cores) against
    the same pass on the f32-input MFMA: outputs, input gradient and every parameter gradient agree as closely as the f32-MFMA path agrees
    with stock torch."""
    from learning_embeddings_amd import resnet as R
    torch.manual_seed(0)
    net = resnet50(num_classes=10).to(DEV).to(memory_format=torch.channels_last)
    net.train()
    x0 = _cl(torch.rand(6, 3, 96, 96))
    res = {}
    g = None
    for mode in ('native', 'x3'):
        monkeypatch.setattr(R, 'F32_MODE', mode)
        for p_ in net.parameters():
            p_.grad = torch.zeros_like(p_)
        x = x0.clone().requires_grad_(True)
        WgradOverlap.instance = WgradOverlap()
        try:
            y = net(x)
            if g is None:
                g = torch.randn_like(y)
            y.backward(g)
            WgradOverlap.instance.join()
            torch.cuda.synchronize()
        finally:
            WgradOverlap.instance = None
        res[mode] = (y.detach().clone(), x.grad.clone(), [p_.grad.clone() for p_ in net.parameters()])
    (y0, dx0, g0), (y1, dx1, g1) = res['native'], res['x3']
    # the same bounds as the f32-MFMA path against stock torch below: through 50 layers of batch statistics over a handful of pixels
    # and ReLU kinks, two fp32-grade convolution families differ by a few per cent in the deepest gradients
    assert (y0 - y1).abs().max().item() <= 2e-4 * (1 + y0.abs().max().item())
    cos = lambda a, b: float(a.double().flatten() @ b.double().flatten() / (a.double().norm() * b.double().norm() + 1e-300))
    assert cos(dx0, dx1) > 0.9995
    for a, b in zip(g0, g1):
        assert cos(a, b) > 0.999, (tuple(a.shape), cos(a, b))


@pytest.mark.parametrize('arch,hw,n', [('resnet18', 64, 8), ('resnet50', 96, 6)])
def test_resnet_f32_own_convolutions_match_stock_torch(arch, hw, n):
    """The whole backbone at the reference's precision through liblecone only -- f32-MFMA convolutions (statistics epilogue, parity-class
    data gradients of the strided layers, the 3-channel stem through a zero 4th channel, weight gradients on the side stream),
    BatchNorm(+add)(+ReLU) and max pooling -- against stock torch fp32 ops on the same weights and batch: outputs, input gradient,
    every parameter gradient."""
    from learning_embeddings_amd import resnet as R
    torch.manual_seed(0)
    net = (resnet18 if arch == 'resnet18' else resnet50)(num_classes=10).to(DEV).to(memory_format=torch.channels_last)
    net.train()
    x0 = _cl(torch.rand(n, 3, hw, hw))
    res = {}
    g = None
    for tag in ('own', 'stock'):
        for p_ in net.parameters():
            p_.grad = torch.zeros_like(p_)                      # the weight-gradient kernels accumulate into existing slots
        x = x0.clone().requires_grad_(True)
        WgradOverlap.instance = WgradOverlap() if tag == 'own' else None
        BatchNormAct2d.fused_enabled = tag == 'own'
        try:
            y = net(x)
            if g is None:
                g = torch.randn_like(y)
            y.backward(g)
            if WgradOverlap.instance is not None:
                WgradOverlap.instance.join()
            torch.cuda.synchronize()
        finally:
            WgradOverlap.instance = None
            BatchNormAct2d.fused_enabled = True
        res[tag] = (y.detach().clone(), x.grad.clone(), {k: p_.grad.clone() for k, p_ in net.named_parameters()})
    ya, yb = res['own'][0], res['stock'][0]
    assert (ya - yb).abs().max().item() <= 2e-4 * (1 + yb.abs().max().item())
    cos = lambda a, b: float(a.double().flatten() @ b.double().flatten() / (a.double().norm() * b.double().norm() + 1e-300))
    assert cos(res['own'][1], res['stock'][1]) > 0.9995
    for k in res['stock'][2]:
        c = cos(res['own'][2][k], res['stock'][2][k])
        assert c > 0.999, (k, c)


# ------------------------------------------------------------------------------------------------ round 3: BatchNorm pieces inside the fp32 convolutions
def _bn_record(N, C, H, W, seed, res=True, relu=True):
    """A real fp32 BatchNorm(+residual)(+ReLU) forward through BNActFn: its input, output, ReLU bitmask (lec_bn_fwd_f32's layout) and
    saved statistics, as the FusionContext record a consumer convolution's data gradient folds."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    xbn = _cl(torch.randn(N, C, H, W, generator=g) * 1.3 + 0.2)
    r = _cl(torch.randn(N, C, H, W, generator=g)) if res else None
    w = (torch.rand(C, generator=g) + 0.5).to(DEV); b = (torch.randn(C, generator=g) * 0.2).to(DEV)
    rm = torch.zeros(C, device=DEV); rv = torch.ones(C, device=DEV)
    xa = xbn.clone().requires_grad_(True)
    ops.fusion().reset()
    out = ops.BNActFn.apply(xa, r, w, b, rm, rv, True, 0.1, 1e-5, relu, res)      # fork <=> residual here
    z = out[0] if res else out
    rec = dict(ops.fusion().forks[z.data_ptr()])
    ops.fusion().reset()
    return z.detach(), rec, w


@pytest.mark.parametrize('N,C,H,W,Cout,R,pad,res', [(4, 256, 14, 14, 64, 1, 0, True), (3, 64, 9, 7, 256, 1, 0, True), (2, 128, 12, 12, 128, 3, 1, False),
                                                      (5, 512, 7, 7, 128, 1, 0, True), (2, 64, 20, 20, 64, 3, 1, False), (1, 1024, 5, 5, 256, 1, 0, True)])
def test_conv_f32_dgrad_with_batchnorm_backward_pass1_in_the_epilogue(N, C, H, W, Cout, R, pad, res):
    """lec_conv_f32_dgrad_fused, fold form: the data gradient of a stride-1 layer whose input z = relu(bn(x) [+ r]) is a BatchNorm output
    writes g = mask * (dx [+ dres]) -- bit-equal to the plain data gradient followed by the add and the mask -- and leaves the partial
    sums of pass 1 (sum g, sum g * xhat per channel) in the BatchNorm workspace: equal to float64 sums of that g to fp32 summation
    error.  Mask bits come from a real lec_bn_fwd_f32 forward (its byte layout); dres is absent for single-consumer outputs."""
    z, rec, _ = _bn_record(N, C, H, W, seed=C + R, res=res)
    g = torch.Generator(device='cpu').manual_seed(77)
    Ho, Wo = H + 2 * pad - R + 1, W + 2 * pad - R + 1
    dy = _cl(torch.randn(N, Cout, Ho, Wo, generator=g))
    w = _cl(torch.randn(Cout, C, R, R, generator=g) * 0.1)
    dres = _cl(torch.randn(N, C, H, W, generator=g)) if res else None
    rec['dres'] = dres
    dx0 = ops.conv_f32_dgrad(dy, w, z.shape, 1, pad)
    got = ops.conv_f32_dgrad_fused(dy, w, z.shape, 1, pad, fold=rec)
    n = ops.fusion().ws_owner[1]
    assert ops.fusion().ws_owner[0] == got.data_ptr() and ops.fusion().folded == {got.data_ptr(): n} and 1 <= n <= 512
    want = (dx0 + dres if res else dx0) * (z > 0)
    assert torch.equal(got, want)
    part = ops._bn_workspace(dy.device)[:n * 2 * C * 4].view(torch.float32).view(n, 2, C).double().sum(0)
    gd = want.double(); xh = (rec['x'].double() - rec['mean'].double().view(1, C, 1, 1)) * rec['invstd'].double().view(1, C, 1, 1)
    s_ref = gd.sum(dim=(0, 2, 3)); q_ref = (gd * xh).sum(dim=(0, 2, 3))
    scale = gd.abs().sum(dim=(0, 2, 3)).max().item()
    assert (part[0] - s_ref).abs().max().item() <= 1e-5 * scale and (part[1] - q_ref).abs().max().item() <= 1e-5 * (gd * xh).abs().sum(dim=(0, 2, 3)).max().item()
    ops.fusion().reset()


# ------------------------------------------------------------------------------------------------ round 3: inference forward (evaluation phases)
AFF_CASES = [  # N, Cin, H, W, Cout, R, stride, pad, residual, relu
    (2, 64, 12, 12, 256, 1, 1, 0, True, True), (3, 256, 9, 7, 64, 1, 1, 0, False, True), (2, 128, 12, 12, 128, 3, 2, 1, False, True),
    (2, 4, 32, 32, 64, 7, 2, 3, False, True), (2, 256, 8, 8, 512, 1, 2, 0, False, False), (1, 512, 7, 7, 512, 3, 1, 1, False, True),
    (40, 256, 14, 14, 1024, 1, 1, 0, True, True), (24, 256, 14, 14, 256, 3, 1, 1, False, True), (5, 64, 9, 9, 64, 3, 1, 1, True, True),
]


@pytest.mark.parametrize('N,Cin,H,W,Cout,R,stride,pad,res,relu', AFF_CASES)
def test_conv_f32_fwd_affine_equals_convolution_then_eval_mode_batchnorm(N, Cin, H, W, Cout, R, stride, pad, res, relu):
    """lec_conv_f32_fwd_affine: F.batch_norm(training=False) (+ residual add, + ReLU) in the convolution's epilogue.  Bit-equal to the two-kernel
    form (lec_conv_f32_fwd, then the fused BatchNorm layer in eval mode: same multiply, add, add, max) with the tile walk and with the balanced
    kernel, and equal to float64 F.conv2d + F.batch_norm to fp32 noise."""
    g = torch.Generator(device='cpu').manual_seed(N + Cin + Cout)
    x = _cl(torch.randn(N, Cin, H, W, generator=g)); w = _cl(torch.randn(Cout, Cin, R, R, generator=g) / (Cin * R * R) ** 0.5)
    bn = BatchNormAct2d(Cout, relu=relu).to(DEV).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(Cout, generator=g) + 0.5); bn.bias.copy_(torch.randn(Cout, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(Cout, generator=g) * 0.2); bn.running_var.copy_(torch.rand(Cout, generator=g) + 0.3)
    Ho = (H + 2 * pad - R) // stride + 1; Wo = (W + 2 * pad - R) // stride + 1
    r = _cl(torch.randn(N, Cout, Ho, Wo, generator=g)) if res else None
    scale, shift = bn.eval_affine()
    with torch.no_grad():
        for mode in (0, 2):
            with _Balanced(mode):
                y0 = ops.conv_f32_fwd(x, w, stride, pad)
                want = bn(y0, r) if res else bn(y0)
                got = ops.conv_f32_fwd_affine(x, w, stride, pad, scale, shift, r, relu)
            assert torch.equal(got, want), 'mode %d' % mode
        ref = F.batch_norm(F.conv2d(x.double(), w.double(), None, stride, pad), bn.running_mean.double(), bn.running_var.double(),
                           bn.weight.double(), bn.bias.double(), False, 0.1, bn.eps)
        if res:
            ref = ref + r.double()
        if relu:
            ref = ref.relu()
    assert (got.double() - ref).abs().max().item() <= 3e-5 * ref.abs().max().item()
    # the two vectors are recomputed ON THE STREAM by every call, into the same storage (a captured inference graph holds the two pointers and
    # replays the launch): they follow the parameters without the host noticing a change
    before = scale.clone()
    with torch.no_grad():
        bn.running_var.mul_(2.0)
    s2, h2 = bn.eval_affine()
    assert s2.data_ptr() == scale.data_ptr() and h2.data_ptr() == shift.data_ptr() and not torch.equal(s2, before)
    assert torch.equal(s2, bn.weight / torch.sqrt(bn.running_var + bn.eps)) and torch.equal(h2, bn.bias - bn.running_mean * s2)


@pytest.mark.parametrize('arch,n,hw', [('resnet18', 6, 64), ('resnet50', 10, 96)])
def test_inference_forward_runs_liblecone_kernels_and_matches_stock_torch(arch, n, hw, monkeypatch):
    """An fp32 forward nobody differentiates (torch.no_grad(): the evaluation phases' image embedding, oe_h.py:1989-2011) takes liblecone's
    kernels: in eval mode every convolution + BatchNorm (+ add) (+ ReLU) is ONE launch (lec_conv_f32_fwd_affine), in train mode (the reference
    embeds the 'train' phase's images with batch statistics) the convolutions leave the statistics to the fused BatchNorm.  Against stock
    torch fp32 (MIOpen convolutions, F.batch_norm) on the same weights: to fp32 noise, running statistics updated alike."""
    from learning_embeddings_amd import resnet as R
    import copy
    torch.manual_seed(0)
    net = (resnet18 if arch == 'resnet18' else R.resnet50)(num_classes=10).to(DEV).to(memory_format=torch.channels_last)
    with torch.no_grad():                                      # running statistics / affine parameters away from their init values
        for m in net.modules():
            if isinstance(m, BatchNormAct2d):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.1)
    x = _cl(torch.rand(n, 3, hw, hw))
    calls = {'affine': 0, 'fwd': 0}
    fa, ff = ops.conv_f32_fwd_affine, ops.conv_f32_fwd
    monkeypatch.setattr(ops, 'conv_f32_fwd_affine', lambda *a, **k: (calls.__setitem__('affine', calls['affine'] + 1), fa(*a, **k))[1])
    monkeypatch.setattr(ops, 'conv_f32_fwd', lambda *a, **k: (calls.__setitem__('fwd', calls['fwd'] + 1), ff(*a, **k))[1])
    fs = ops.conv_f32_stem_fwd                                      # (the stem of 64 / 128 / 224-pixel images has its own forward kernel)
    monkeypatch.setattr(ops, 'conv_f32_stem_fwd', lambda *a, **k: (calls.__setitem__('fwd', calls['fwd'] + 1), fs(*a, **k))[1])
    n_conv = sum(isinstance(m, torch.nn.Conv2d) for m in net.modules())
    for train in (False, True):
        ref_net = copy.deepcopy(net); net.train(train); ref_net.train(train)
        calls['affine'] = calls['fwd'] = 0
        with torch.no_grad():
            y = net(x)
            monkeypatch.setattr(R, 'MFMA_F32', False); BatchNormAct2d.fused_enabled = False
            try:
                y_ref = ref_net(x)
            finally:
                monkeypatch.setattr(R, 'MFMA_F32', True); BatchNormAct2d.fused_enabled = True
        assert (calls['fwd'], calls['affine']) == ((n_conv, 0) if train else (0, n_conv)), calls
        assert (y - y_ref).abs().max().item() <= 2e-4 * (1 + y_ref.abs().max().item())
        if train:
            for (k, a), (_, b) in zip(net.named_buffers(), ref_net.named_buffers()):
                if a.dtype.is_floating_point:
                    assert torch.allclose(a, b, rtol=1e-4, atol=1e-5), k
    ops.fusion().reset()


def test_conv_macs_walks_the_modules_on_the_gpu_too():
    """bench.py's analytic flops come from resnet.conv_macs, a forward with hooks on the convolution MODULES: the inference forward bypasses
    the modules (conv_bn calls liblecone directly), so the walk must take the module path (it once counted 0.7 instead of 24.5 GFLOP per image)."""
    from learning_embeddings_amd.resnet import conv_macs
    from learning_embeddings_amd import resnet as R
    net = resnet50(num_classes=10).to(DEV).to(memory_format=torch.channels_last)
    R.library_launches(reset=True)
    assert abs(conv_macs(net, 224) / 1e9 - 4.0872) < 0.05
    assert net.training
    assert sum(R.library_launches().values()) == 0, 'the FLOP-counting walk must not hand a convolution to the library (it ran MIOpen\'s naive kernel 53 times until round 6)'


# ------------------------------------------------------------------------------------------------ round 3: the balanced (stream-K) kernel
class _Balanced:
    """The `schedule` argument of the fp32 forward / data-gradient entry points (LEC_SCHEDULE_*) for the ops called in a block -- through the
    current FusionContext, as a backbone's conv_schedule reaches them (2: the balanced kernel wherever it applies, 0: never)."""
    def __init__(self, mode): self.mode = mode
    def __enter__(self): self.fc = ops.fusion(); self.prev = self.fc.schedule; self.fc.schedule = self.mode
    def __exit__(self, *a): self.fc.schedule = self.prev


SK_CASES = [  # N, Cin, H, W, Cout, R, pad: one workgroup per K chunk (tiny), tiles over 3 - 4 workgroups, whole tiles + split ends, ragged rows / channels
    (2, 64, 12, 12, 256, 1, 0), (1, 512, 7, 7, 512, 3, 1), (64, 256, 14, 14, 256, 3, 1), (32, 64, 56, 56, 256, 1, 0), (3, 256, 9, 7, 128, 1, 0),
    (16, 128, 28, 28, 128, 3, 1), (5, 1024, 5, 5, 2048, 1, 0),
]


@pytest.mark.parametrize('N,Cin,H,W,Cout,R,pad', SK_CASES)
def test_conv_f32_balanced_kernel_equals_float64_on_integers_and_is_run_to_run_identical(N, Cin, H, W, Cout, R, pad):
    """conv_f32_act_sk_kernel (schedule LEC_SCHEDULE_BALANCED: every eligible launch): forward with statistics and the stride-1 data gradient.
    Small-integer operands: every partial sum is exact, so the outputs and the per-channel statistics must EQUAL float64 whatever way the
    K range of a tile was cut over workgroups (a lost, doubled or stale partial is a mismatch, not noise).  Random operands: fp32 noise
    against float64, and bit-identical results over repeated launches (the fix-up adds the partials in workgroup order, not arrival
    order; counters re-arm themselves)."""
    g = torch.Generator(device='cpu').manual_seed(Cin + Cout + R + N)
    for kind in ('int', 'rand'):
        if kind == 'int':
            x = torch.randint(-3, 4, (N, Cin, H, W), generator=g).float(); w = torch.randint(-2, 3, (Cout, Cin, R, R), generator=g).float()
        else:
            x = torch.randn(N, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, R, R, generator=g) / (Cin * R * R) ** 0.5
        x = _cl(x); w = _cl(w)
        yr = F.conv2d(x.double(), w.double(), None, 1, pad)
        dy = _cl(torch.randint(-2, 3, yr.shape, generator=g).float() if kind == 'int' else torch.randn(yr.shape, generator=g))
        dxr = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), 1, pad)
        outs = []
        with _Balanced(2):
            for rep in range(3):
                ops.fusion().reset()
                y = ops.conv_f32_fwd(x, w, 1, pad, want_stats=True)
                k = ops.fusion().ws_owner[1]
                part = ops._bn_workspace(x.device).view(torch.float32)[:k * 2 * Cout].view(k, 2, Cout).clone()
                dx = ops.conv_f32_dgrad(dy, w, x.shape, 1, pad)
                outs.append((y, part, dx))
        assert k == (N * yr.shape[2] * yr.shape[3] + 127) // 128, 'one statistics row per m-tile: the balanced kernel ran'
        y, part, dx = outs[0]
        for o in outs[1:]:
            assert torch.equal(o[0], y) and torch.equal(o[1], part) and torch.equal(o[2], dx), 'repeated launches differ'
        ps = part.double().sum(0)
        if kind == 'int':
            assert torch.equal(y.double(), yr), 'forward'
            assert torch.equal(dx.double(), dxr), 'data gradient'
            assert torch.equal(ps[0], yr.sum(dim=(0, 2, 3))) and torch.equal(ps[1], (yr ** 2).sum(dim=(0, 2, 3)))
        else:
            tol = lambda ref: 2e-5 * ref.abs().max().item()
            assert (y.double() - yr).abs().max().item() <= tol(yr)
            assert (dx.double() - dxr).abs().max().item() <= tol(dxr)
            assert torch.allclose(ps[0], yr.sum(dim=(0, 2, 3)), rtol=1e-4, atol=1e-3 * (N * H * W / 288) ** 0.5)
            with _Balanced(0):
                y0 = ops.conv_f32_fwd(x, w, 1, pad); dx0 = ops.conv_f32_dgrad(dy, w, x.shape, 1, pad)
            assert (y - y0).abs().max().item() <= 4e-6 * yr.abs().max().item() and (dx - dx0).abs().max().item() <= 4e-6 * dxr.abs().max().item()
    ops.fusion().reset()


FULL_CASES = [  # N, Cin, H, W, Cout, R, stride, pad: layer shapes of the bench step at the rows of a pass (256) and of a whole step (512)
    (512, 256, 14, 14, 256, 3, 1, 1), (256, 1024, 14, 14, 256, 1, 1, 0), (256, 128, 56, 56, 128, 3, 2, 1), (512, 512, 7, 7, 512, 3, 1, 1), (256, 256, 14, 14, 1024, 1, 1, 0),
]


@pytest.mark.parametrize('N,Cin,H,W,Cout,R,stride,pad', FULL_CASES)
def test_conv_f32_full_size_layers_are_exact_on_integers_with_either_schedule(N, Cin, H, W, Cout, R, stride, pad):
    """BASELINE-size layers (392 x 2^k tiles: the launches the balanced kernel was built for) on operands from {-1, 0, 1}: every partial sum is
    exactly representable, so forward, data gradient and weight gradient must EQUAL the float64 convolution -- with the tile walk and with the
    balanced kernel (512 workgroups, every tile boundary inside some workgroup's run), whatever the fix-up order or the atomics' order -- and
    the per-tile statistics rows must add up to the exact per-channel sums."""
    g = torch.Generator(device='cpu').manual_seed(N + Cin + Cout + R)
    x = _cl(torch.randint(-1, 2, (N, Cin, H, W), generator=g).float()); w = _cl(torch.randint(-1, 2, (Cout, Cin, R, R), generator=g).float())
    yr = F.conv2d(x.double(), w.double(), None, stride, pad)
    dy = _cl(torch.randint(-1, 2, yr.shape, generator=g).float())
    dxr = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), stride, pad)
    dwr = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), stride, pad)
    rows = {}
    for mode in (0, 1):
        with _Balanced(mode):
            ops.fusion().reset()
            y = ops.conv_f32_fwd(x, w, stride, pad, want_stats=True)
            k = ops.fusion().ws_owner[1]; rows[mode] = k
            s0 = ops._bn_workspace(x.device).view(torch.float32)[:k * 2 * Cout].view(k, 2, Cout)[:, 0].double().sum(0)
            dx = ops.conv_f32_dgrad(dy, w, x.shape, stride, pad)
        assert torch.equal(y.double(), yr), 'forward, mode %d' % mode
        assert torch.equal(s0, yr.sum(dim=(0, 2, 3))), 'statistics, mode %d' % mode
        assert torch.equal(dx.double(), dxr), 'data gradient, mode %d' % mode
        del y, dx
    mtiles = (N * yr.shape[2] * yr.shape[3] + 127) // 128
    assert rows[0] == min(mtiles, 512) and rows[1] == (mtiles if mtiles <= 2048 else min(mtiles, 512))     # the balanced kernel ran where expected
    dw = torch.zeros_like(w)
    ops.conv_f32_wgrad(dy, x, dw, stride, pad)
    assert torch.equal(dw.double(), dwr), 'weight gradient'
    ops.fusion().reset()


@pytest.mark.parametrize('N,Cin,H,W,Cout,R,pad,wgs', [(10, 256, 14, 14, 256, 3, 1, 288), (10, 1024, 14, 14, 256, 1, 0, 128), (10, 512, 7, 7, 512, 3, 1, 288),
                                                        (10, 256, 14, 14, 1024, 1, 0, 0), (256, 256, 14, 14, 256, 3, 1, 512)])
def test_conv_f32_launcher_cuts_small_batches_along_k(N, Cin, H, W, Cout, R, pad, wgs):
    """The launcher's own choice (schedule LEC_SCHEDULE_AUTO): a 10-image batch (the reference's evaluation chunk) gives layer3 / layer4 a few dozen
    tiles -- the balanced kernel cuts them along K into `wgs` workgroups of >= 8 chunks (0: the cut would not add parallel work, the tile walk runs).
    Which kernel ran shows in the number of statistics rows (one per m-tile when balanced); results exact on integers either way."""
    g = torch.Generator(device='cpu').manual_seed(N + Cin + Cout + R)
    x = _cl(torch.randint(-3, 4, (N, Cin, H, W), generator=g).float()); w = _cl(torch.randint(-2, 3, (Cout, Cin, R, R), generator=g).float())
    yr = F.conv2d(x.double(), w.double(), None, 1, pad)
    with _Balanced(1):
        ops.fusion().reset()
        y = ops.conv_f32_fwd(x, w, 1, pad, want_stats=True)
        k = ops.fusion().ws_owner[1]
        part = ops._bn_workspace(x.device).view(torch.float32)[:k * 2 * Cout].view(k, 2, Cout).double().sum(0)
    mtiles = (N * H * W + 127) // 128
    tiles = mtiles * (Cout // 128); nchunks = Cin * R * R // 32
    assert (min(512, tiles * nchunks // 8) if wgs else 0) == wgs            # the launcher's arithmetic, restated
    assert k == (mtiles if wgs else min(mtiles, 512))
    assert torch.equal(y.double(), yr) and torch.equal(part[0], yr.sum(dim=(0, 2, 3))) and torch.equal(part[1], (yr ** 2).sum(dim=(0, 2, 3)))
    ops.fusion().reset()


@pytest.mark.parametrize('N,C,H,W,Cout,R,pad,res', [(4, 256, 14, 14, 64, 1, 0, True), (2, 128, 12, 12, 128, 3, 1, False), (48, 256, 14, 14, 256, 3, 1, False),
                                                      (40, 512, 7, 7, 128, 1, 0, True), (24, 128, 28, 28, 512, 1, 0, True)])
def test_conv_f32_balanced_kernel_with_the_fold_epilogue(N, C, H, W, Cout, R, pad, res):
    """The balanced kernel's fold epilogue (pass 1 of the BatchNorm backward): g bit-equal to the balanced plain data gradient followed by the
    add and the mask, the partial sums (one row per m-tile) equal to float64 sums of that g to fp32 error, repeated launches identical."""
    z, rec, _ = _bn_record(N, C, H, W, seed=C + R, res=res)
    g = torch.Generator(device='cpu').manual_seed(78)
    Ho, Wo = H + 2 * pad - R + 1, W + 2 * pad - R + 1
    dy = _cl(torch.randn(N, Cout, Ho, Wo, generator=g))
    w = _cl(torch.randn(Cout, C, R, R, generator=g) * 0.1)
    rec['dres'] = _cl(torch.randn(N, C, H, W, generator=g)) if res else None
    with _Balanced(2):
        dx0 = ops.conv_f32_dgrad(dy, w, z.shape, 1, pad)
        got = ops.conv_f32_dgrad_fused(dy, w, z.shape, 1, pad, fold=rec)
        n = ops.fusion().ws_owner[1]
        part = ops._bn_workspace(dy.device)[:n * 2 * C * 4].view(torch.float32).view(n, 2, C).clone()
        again = ops.conv_f32_dgrad_fused(dy, w, z.shape, 1, pad, fold=rec)
        part2 = ops._bn_workspace(dy.device)[:n * 2 * C * 4].view(torch.float32).view(n, 2, C)
    assert n == (N * H * W + 127) // 128
    want = (dx0 + rec['dres'] if res else dx0) * (z > 0)
    assert torch.equal(got, want) and torch.equal(again, got) and torch.equal(part2, part)
    gd = want.double(); xh = (rec['x'].double() - rec['mean'].double().view(1, C, 1, 1)) * rec['invstd'].double().view(1, C, 1, 1)
    ps = part.double().sum(0)
    assert (ps[0] - gd.sum(dim=(0, 2, 3))).abs().max().item() <= 1e-5 * gd.abs().sum(dim=(0, 2, 3)).max().item()
    assert (ps[1] - (gd * xh).sum(dim=(0, 2, 3))).abs().max().item() <= 1e-5 * (gd * xh).abs().sum(dim=(0, 2, 3)).max().item()
    ops.fusion().reset()


@pytest.mark.parametrize('N,Cin,H,W,Cout', [(4, 64, 14, 14, 256), (3, 128, 9, 7, 512), (2, 256, 12, 12, 1024), (6, 512, 7, 7, 2048), (2, 64, 20, 20, 64)])
def test_conv_f32_gradients_with_batchnorm_backward_pass2_on_the_operand_load(N, Cin, H, W, Cout):
    """lec_conv_f32_dgrad_fused / lec_conv_f32_wgrad_fused, on-load form (1x1 / stride 1): handed g, the BatchNorm input x and the
    coefficient vectors of lec_bn_bwd_coeffs_f32, both kernels form dy = A g + B x + D per output channel while loading.  Checked (a)
    against the same kernels fed the materialised dy (fp32 fma of the same three terms: equal up to the fma's single rounding, a few
    ulp of the operands), (b) the coefficients against float64 of lec_bn_bwd_f32's formula dx = gamma invstd (g - c1 - xhat c2)."""
    gen = torch.Generator(device='cpu').manual_seed(Cout + Cin)
    M = N * H * W
    g_ = _cl(torch.randn(N, Cout, H, W, generator=gen)); xb = _cl(torch.randn(N, Cout, H, W, generator=gen) * 1.5 + 0.3)
    xin = _cl(torch.randn(N, Cin, H, W, generator=gen)); w = _cl(torch.randn(Cout, Cin, 1, 1, generator=gen) * 0.1)
    gamma = (torch.rand(Cout, generator=gen) + 0.5).to(DEV)
    mean = xb.double().mean(dim=(0, 2, 3)); var = xb.double().var(dim=(0, 2, 3), unbiased=False)
    invstd = (1.0 / torch.sqrt(var + 1e-5))
    mean32, invstd32 = mean.float(), invstd.float()
    # partial rows the way a reduce pass leaves them: here simply two rows that add up to the sums
    xh = (xb.double() - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
    s = g_.double().sum(dim=(0, 2, 3)); q = (g_.double() * xh).sum(dim=(0, 2, 3))
    ws = ops._bn_workspace(g_.device)
    part = ws[:2 * 2 * Cout * 4].view(torch.float32).view(2, 2, Cout)
    part[0, 0] = (s * 0.25).float(); part[1, 0] = (s - (s * 0.25).float().double()).float()
    part[0, 1] = (q * 0.5).float(); part[1, 1] = (q - (q * 0.5).float().double()).float()
    dgamma = torch.empty(Cout, device=DEV); dbeta = torch.empty(Cout, device=DEV); coef = torch.empty(3 * Cout, device=DEV)
    from learning_embeddings_amd._lib import lib, check, dptr, stream_ptr
    check(lib.lec_bn_bwd_coeffs_f32(M, Cout, 2, dptr(gamma), dptr(mean32), dptr(invstd32), dptr(dgamma), dptr(dbeta), dptr(coef), dptr(ws), ws.numel(), 0, stream_ptr()))
    c1, c2 = s / M, q / M
    gs = gamma.double() * invstd32.double()
    want_coef = torch.stack([gs, -gs * invstd32.double() * c2, gs * (invstd32.double() * c2 * mean32.double() - c1)])
    assert (coef.view(3, Cout).double() - want_coef).abs().max().item() <= 2e-6 * want_coef.abs().max().item()
    assert (dbeta.double() - s).abs().max().item() <= 1e-6 * s.abs().max().item() + 1e-6 and (dgamma.double() - q).abs().max().item() <= 1e-6 * q.abs().max().item() + 1e-6
    A, B, D = (coef.view(3, Cout)[i].view(1, Cout, 1, 1) for i in range(3))
    dy = torch.addcmul(torch.addcmul(D.expand_as(g_), xb, B), g_, A).contiguous(memory_format=torch.channels_last)
    # (a) data gradient
    dx_ref = ops.conv_f32_dgrad(dy, w, xin.shape, 1, 0)
    dx = ops.conv_f32_dgrad_fused(g_, w, xin.shape, 1, 0, xf=(xb, coef))
    tol = 4e-6 * (dy.abs().max().item() * w.abs().max().item() * Cout) ** 1.0
    assert (dx - dx_ref).abs().max().item() <= tol, ((dx - dx_ref).abs().max().item(), tol)
    # ... and against float64 of the BatchNorm formula followed by the convolution's data gradient
    dy64 = gs.view(1, -1, 1, 1) * (g_.double() - c1.view(1, -1, 1, 1) - xh * c2.view(1, -1, 1, 1))
    dx64 = torch.einsum('nohw,oi->nihw', dy64, w.double().view(Cout, Cin))
    assert (dx.double() - dx64).abs().max().item() <= 3e-5 * dx64.abs().max().item()
    # (b) weight gradient
    dw_ref = torch.zeros(Cout, Cin, 1, 1, device=DEV).contiguous(memory_format=torch.channels_last); dw = dw_ref.clone()
    ops.conv_f32_wgrad(dy, xin, dw_ref, 1, 0)
    ops.conv_f32_wgrad(g_, xin, dw, 1, 0, xf=(xb, coef))
    dw64 = torch.einsum('nohw,nihw->oi', dy64, xin.double())
    assert (dw.view(Cout, Cin).double() - dw64).abs().max().item() <= 3e-5 * dw64.abs().max().item()
    assert (dw - dw_ref).abs().max().item() <= 3e-5 * dw64.abs().max().item()


@pytest.mark.parametrize('hw,n', [(64, 6), (96, 4)])
def test_resnet50_f32_fused_batchnorm_backward_matches_the_unfused_path(hw, n):
    """The whole fp32 ResNet-50 with the round-3 fusions -- pass 1 of every foldable BatchNorm backward in the epilogue of the data
    gradient that produces its gradient, pass 2 on the operand load of the 1x1 convolution behind it -- against the same network with
    both switched off (separate BatchNorm passes): outputs bit-equal (forward is untouched), input gradient and every parameter
    gradient equal to fp32 summation-order error; and the fused kernels must actually have run."""
    torch.manual_seed(0)
    net = resnet50(num_classes=10).to(DEV).to(memory_format=torch.channels_last).train()
    x0 = _cl(torch.rand(n, 3, hw, hw))
    res, calls, g = {}, {}, None
    orig = ops.conv_f32_dgrad_fused
    keep = (ops.FOLD_BN_BWD_F32, ops.LAZY_BN_PASS2_F32, ops.FOLD_F32_MIN_CHANNELS, ops.LAZY_F32_MIN_ELEMS)
    ops.FOLD_F32_MIN_CHANNELS = 0; ops.LAZY_F32_MIN_ELEMS = 0      # every eligible layer, whatever the size thresholds of the default policy
    try:
        for tag in ('fused', 'plain'):
            ops.FOLD_BN_BWD_F32 = ops.LAZY_BN_PASS2_F32 = tag == 'fused'
            def counted(*a, _t=tag, **k):
                key = (_t, 'xf' if k.get('xf') is not None else '', 'fold' if k.get('fold') is not None else '')
                calls[key] = calls.get(key, 0) + 1
                return orig(*a, **k)
            ops.conv_f32_dgrad_fused = counted
            for p_ in net.parameters():
                p_.grad = torch.zeros_like(p_)
            x = x0.clone().requires_grad_(True)
            WgradOverlap.instance = WgradOverlap()
            y = net(x)
            if g is None:
                g = torch.randn_like(y)
            y.backward(g)
            WgradOverlap.instance.join(); torch.cuda.synchronize()
            res[tag] = (y.detach().clone(), x.grad.clone(), {k: p_.grad.clone() for k, p_ in net.named_parameters()})
            assert not net.fusion.lazy_dx and not net.fusion.folded
    finally:
        WgradOverlap.instance = None
        ops.conv_f32_dgrad_fused = orig
        ops.FOLD_BN_BWD_F32, ops.LAZY_BN_PASS2_F32, ops.FOLD_F32_MIN_CHANNELS, ops.LAZY_F32_MIN_ELEMS = keep
    assert not any(k[0] == 'plain' for k in calls), calls
    n_fold = sum(v for k, v in calls.items() if k[2]); n_xf = sum(v for k, v in calls.items() if k[1])
    # 16 conv1 (15 fold a block output, every conv1 forms its gradient on load), 13 stride-1 conv2 (fold bn1), 16 conv3 (fold bn2, on load)
    assert n_fold >= 15 + 13 + 16 and n_xf >= 16 + 16, calls
    assert torch.equal(res['fused'][0], res['plain'][0])
    cos = lambda a, b: float(a.double().flatten() @ b.double().flatten() / (a.double().norm() * b.double().norm() + 1e-300))
    assert cos(res['fused'][1], res['plain'][1]) > 0.999999
    for k in res['plain'][2]:
        a, b = res['fused'][2][k], res['plain'][2][k]
        if b.norm() > 0:
            assert cos(a, b) > 0.99999, (k, cos(a, b))
            assert (a - b).abs().max().item() <= 2e-3 * b.abs().max().item(), (k, (a - b).abs().max().item(), b.abs().max().item())


@pytest.mark.parametrize('shape', [(6, 40, 24), (4, 64, 64)])
def test_stem_tail_as_one_op_equals_batchnorm_relu_then_maxpool(shape):
    """ops.BNReluPoolFn (lec_bn_relu_maxpool_fwd_f32 / _bwd_f32): the stem's maxpool(relu(bn1(conv1(x)))) with the BatchNorm apply and the ReLU on the
    pooling's loads, and in backward the pooling's input gradient rebuilt on the fly inside both BatchNorm passes.  Against the separate ops
    (BNActFn + MaxPool3x3s2Fn, LEC_FUSE_STEM_POOL off) on the same ResNet-18: output and running statistics bit for bit (forward is the same
    arithmetic), every parameter gradient to summation order."""
    from learning_embeddings_amd import resnet as R
    from learning_embeddings_amd.resnet import WgradOverlap
    n, h, w = shape
    x = torch.rand(n, 3, h, w, device=DEV, generator=torch.Generator(DEV).manual_seed(3)).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(n, 10, device=DEV, generator=torch.Generator(DEV).manual_seed(4))
    calls = {'fused': 0}
    orig = ops.BNReluPoolFn.forward

    def counted(ctx, *a, **k):
        calls['fused'] += 1
        return orig(ctx, *a, **k)
    res = {}
    old_inst, old_flag = WgradOverlap.instance, ops.FUSE_STEM_POOL
    WgradOverlap.instance = WgradOverlap()
    ops.BNReluPoolFn.forward = staticmethod(counted)
    try:
        for tag in ('separate', 'fused'):
            ops.FUSE_STEM_POOL = tag == 'fused'
            torch.manual_seed(0)
            m = R.resnet18(num_classes=10).to(DEV).to(memory_format=torch.channels_last).train()
            for p_ in m.parameters():
                p_.grad = torch.zeros_like(p_)
            feats = {}
            hook = m.maxpool.register_forward_hook(lambda mod, i, o: feats.__setitem__('p', o.detach().clone()))
            y = m(x)
            hook.remove()
            y.backward(gy)
            WgradOverlap.instance.join(); torch.cuda.synchronize()
            res[tag] = (y.detach().clone(), m.bn1.running_mean.clone(), m.bn1.running_var.clone(), {k: p_.grad.double().clone() for k, p_ in m.named_parameters()})
        assert calls['fused'] == 1                                 # (the module hook does not fire on the fused path: the pooling module is not called)
    finally:
        ops.BNReluPoolFn.forward = staticmethod(orig)
        WgradOverlap.instance, ops.FUSE_STEM_POOL = old_inst, old_flag
    ya, rma, rva, ga = res['separate']; yb, rmb, rvb, gb = res['fused']
    assert torch.equal(rma, rmb) and torch.equal(rva, rvb)
    assert torch.equal(ya, yb)                                     # same pooled activation, same network behind it
    for k in ga:
        scale = ga[k].abs().max().item() + 1e-12
        assert (ga[k] - gb[k]).abs().max().item() <= 2e-4 * scale, (k, (ga[k] - gb[k]).abs().max().item(), scale)
    assert gb['bn1.weight'].abs().max() > 0 and gb['conv1.weight'].abs().max() > 0


def test_stem_tail_kernels_against_torch():
    """The fused kernels through the C ABI against plain torch ops in float64: p / argmax semantics (first maximum wins), dx, d gamma, d beta."""
    import torch.nn.functional as F
    g = torch.Generator(DEV).manual_seed(11)
    N, C_, H, W = 3, 16, 12, 20
    x = torch.randn(N, C_, H, W, device=DEV, generator=g).contiguous(memory_format=torch.channels_last)
    gamma = torch.rand(C_, device=DEV, generator=g) + 0.5; beta = torch.randn(C_, device=DEV, generator=g) * 0.3
    xd = x.double().requires_grad_(True); gd = gamma.double().requires_grad_(True); bd = beta.double().requires_grad_(True)
    mean = x.double().mean((0, 2, 3)); var = x.double().var((0, 2, 3), unbiased=False)
    invstd = (1.0 / torch.sqrt(var + 1e-5))
    z = F.relu((xd - mean[None, :, None, None]) * invstd[None, :, None, None] * gd[None, :, None, None] + bd[None, :, None, None])
    # batch statistics depend on x: differentiate through them like F.batch_norm(training=True)
    mean_g = xd.mean((0, 2, 3)); var_g = xd.var((0, 2, 3), unbiased=False)
    zg = F.relu((xd - mean_g[None, :, None, None]) / torch.sqrt(var_g + 1e-5)[None, :, None, None] * gd[None, :, None, None] + bd[None, :, None, None])
    pd = F.max_pool2d(zg, 3, 2, 1)
    dp = torch.randn(N, C_, H // 2, W // 2, device=DEV, generator=g).contiguous(memory_format=torch.channels_last)
    pd.backward(dp.double())
    save_mean = mean.float().contiguous(); save_invstd = invstd.float().contiguous()
    scale = (gamma * save_invstd).contiguous(); shift = (beta - save_mean * scale).contiguous()
    p = torch.empty((N, C_, H // 2, W // 2), device=DEV, memory_format=torch.channels_last)
    arg = torch.empty(N * (H // 2) * (W // 2) * C_, dtype=torch.uint8, device=DEV)
    _lib.check(_lib.lib.lec_bn_relu_maxpool_fwd_f32(_lib.dptr(x), N, H, W, C_, _lib.dptr(scale), _lib.dptr(shift), _lib.dptr(p), _lib.dptr(arg), _lib.stream_ptr()))
    assert torch.allclose(p.double(), pd.detach(), rtol=1e-5, atol=1e-6)
    dx = torch.empty_like(x); dgamma = torch.zeros(C_, device=DEV); dbeta = torch.zeros(C_, device=DEV)
    ws = torch.zeros(_lib.lib.lec_bn_workspace_bytes(C_), dtype=torch.uint8, device=DEV)
    _lib.check(_lib.lib.lec_bn_relu_maxpool_bwd_f32(_lib.dptr(dp), None, _lib.dptr(arg), _lib.dptr(x), N, H, W, C_, _lib.dptr(gamma), _lib.dptr(beta), _lib.dptr(save_mean),
                                                    _lib.dptr(save_invstd), _lib.dptr(dx), _lib.dptr(dgamma), _lib.dptr(dbeta), _lib.dptr(ws), ws.numel(), 0,
                                                    _lib.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.allclose(dx.double(), xd.grad, rtol=1e-4, atol=2e-5), (dx.double() - xd.grad).abs().max()
    assert torch.allclose(dgamma.double(), gd.grad, rtol=1e-4, atol=1e-4) and torch.allclose(dbeta.double(), bd.grad, rtol=1e-4, atol=1e-4)
    # accumulate = 1 ADDS into the parameter gradients; the pooled gradient handed over as two branch gradients (dp = dp_a + dp_b, added on load)
    dp_b = (torch.randn(dp.shape, device=DEV, generator=g) * 0.5).contiguous(memory_format=torch.channels_last); dp_a = (dp - dp_b).contiguous(memory_format=torch.channels_last)
    _lib.check(_lib.lib.lec_bn_relu_maxpool_bwd_f32(_lib.dptr(dp_a), _lib.dptr(dp_b), _lib.dptr(arg), _lib.dptr(x), N, H, W, C_, _lib.dptr(gamma), _lib.dptr(beta), _lib.dptr(save_mean),
                                                    _lib.dptr(save_invstd), _lib.dptr(dx), _lib.dptr(dgamma), _lib.dptr(dbeta), _lib.dptr(ws), ws.numel(), 1,
                                                    _lib.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.allclose(dgamma.double(), 2 * gd.grad, rtol=1e-4, atol=2e-4)
    # arguments the kernels cannot serve are refused, not mis-run
    with pytest.raises(_lib.LeconeError):
        _lib.check(_lib.lib.lec_bn_relu_maxpool_fwd_f32(_lib.dptr(x), N, H + 1, W, C_, _lib.dptr(scale), _lib.dptr(shift), _lib.dptr(p), _lib.dptr(arg), _lib.stream_ptr()))


def test_global_avgpool_backward_is_channels_last_and_equals_the_framework_op():
    """resnet._GlobalAvgPoolFn: same forward bits as flatten(adaptive_avg_pool2d(x, 1)); its backward writes the gradient of a channels_last input in
    channels_last (the BatchNorm backward behind it takes it without a layout conversion) and equals the framework's."""
    from learning_embeddings_amd import resnet as R
    g = torch.Generator(DEV).manual_seed(2)
    x = torch.randn(6, 64, 7, 7, device=DEV, generator=g).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(6, 64, device=DEV, generator=g)
    xa = x.clone().requires_grad_(True); xb = x.clone().requires_grad_(True)
    ya = R._global_avgpool(torch.nn.AdaptiveAvgPool2d((1, 1)), xa); yb = torch.flatten(F.adaptive_avg_pool2d(xb, 1), 1)
    assert torch.equal(ya, yb)
    ya.backward(gy); yb.backward(gy)
    assert xa.grad.is_contiguous(memory_format=torch.channels_last)
    assert torch.allclose(xa.grad, xb.grad, rtol=1e-6, atol=0)


def test_six_forwards_before_any_backward_keep_their_own_fusion_context():
    """the backbone's pool of fusion contexts (hand-off records + BatchNorm workspace of ONE forward / backward pair) used to be capped
    at four and silently handed the OLDEST in-flight context to a fifth forward.  Six training forwards (different batches) before any backward, backwards in a
    scrambled order: every input gradient must equal the one of that forward run alone, the parameter gradients their sum.  Then: a forward whose output is
    dropped without a backward frees its context (no growth), and the cap raises instead of recycling."""
    torch.manual_seed(0)
    net = resnet50(num_classes=10).to(DEV).to(memory_format=torch.channels_last).train()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = 0.0                                        # (running statistics play no part in a training forward; keep both runs from the same state)
    xs = [_cl(torch.rand(4, 3, 64, 64)) for _ in range(6)]
    gs = None
    WgradOverlap.instance = WgradOverlap()
    try:
        def zero():
            for p_ in net.parameters():
                p_.grad = torch.zeros_like(p_)
        # alone, one after the other
        alone_dx, alone_y = [], []
        zero()
        for i, x0 in enumerate(xs):
            x = x0.clone().requires_grad_(True)
            y = net(x)
            if gs is None:
                gs = [torch.randn_like(y) for _ in xs]
            y.backward(gs[i]); WgradOverlap.instance.join()
            alone_dx.append(x.grad.clone()); alone_y.append(y.detach().clone())
        torch.cuda.synchronize()
        alone_p = {k: p_.grad.clone() for k, p_ in net.named_parameters()}
        assert len(net._fusions) == 1
        # all six forwards first
        zero()
        ins = [x0.clone().requires_grad_(True) for x0 in xs]
        outs = [net(x) for x in ins]
        assert len(net._fusions) == 6 and all(c.busy() for c in net._fusions)
        for i in (3, 0, 5, 1, 4, 2):
            outs[i].backward(gs[i]); WgradOverlap.instance.join()
        torch.cuda.synchronize()
        assert not any(c.busy() for c in net._fusions)
        cos = lambda a, b: float(a.double().flatten() @ b.double().flatten() / (a.double().norm() * b.double().norm() + 1e-300))
        for i in range(6):
            assert torch.equal(outs[i].detach(), alone_y[i])
            assert cos(ins[i].grad, alone_dx[i]) > 0.999999, (i, cos(ins[i].grad, alone_dx[i]))
        for k, p_ in net.named_parameters():
            if alone_p[k].norm() > 0:
                assert cos(p_.grad, alone_p[k]) > 0.99999, (k, cos(p_.grad, alone_p[k]))
        # outputs dropped without a backward: the graph is freed, the context is free again
        del outs, ins
        for _ in range(10):
            y = net(xs[0].clone().requires_grad_(True)); del y
        assert len(net._fusions) == 6
        # the cap raises
        net.max_forwards_in_flight = 7
        held = [net(xs[0].clone().requires_grad_(True)) for _ in range(7)]
        with pytest.raises(RuntimeError, match='waiting for their backward'):
            net(xs[0].clone().requires_grad_(True))
        del held
    finally:
        WgradOverlap.instance = None


def test_conv_f32_batches_beyond_two_gib_run_as_image_groups_on_the_own_kernels():
    """lec_conv_f32_* with tensors of 2 GiB and more (such batches -- the reference's own B K-row forwards, oe_h.py:980-985, 1003-1009 -- used to fall
    back to the library silently): the entry points split the batch into groups of images.  A convolution is independent per image, so the result must EQUAL, bit
    for bit, the two halves run as separate calls; the statistics partials of all groups add up to the sums over the whole output; the weight gradient to the sum
    of the halves'."""
    g = torch.Generator(device='cpu').manual_seed(3)
    N, Cin, H, W, Cout = 10, 64, 224, 224, 64                      # input 10 x 12.8 MB ... small; force groups with a large pixel count instead
    N, Cin, H, W, Cout, R, st, pd = 44, 64, 224, 224, 256, 1, 1, 0  # output 44 x 224 x 224 x 256 x 4 B = 2.26 GiB: 41 images per launch
    x = _cl(torch.randn(N, Cin, H, W, generator=g)); w = _cl(torch.randn(Cout, Cin, R, R, generator=g) / 8.0)
    assert N * H * W * Cout * 4 >= (1 << 31)
    y = ops.conv_f32_fwd(x, w, st, pd, want_stats=True)
    k = ops._BN_WS_OWNER[1]; ops._BN_WS_OWNER[0] = 0
    part = ops._bn_workspace(x.device).view(torch.float32)[:k * 2 * Cout].view(k, 2, Cout).double().sum(0)
    ya = ops.conv_f32_fwd(x[:41].contiguous(memory_format=torch.channels_last), w, st, pd)
    yb = ops.conv_f32_fwd(x[41:].contiguous(memory_format=torch.channels_last), w, st, pd)
    assert torch.equal(y[:41], ya) and torch.equal(y[41:], yb)
    assert torch.allclose(part[0], y.double().sum(dim=(0, 2, 3)), rtol=1e-5, atol=1e-2) and torch.allclose(part[1], (y.double() ** 2).sum(dim=(0, 2, 3)), rtol=1e-5, atol=1e-2)
    del ya, yb
    dy = y; dy.normal_(generator=None)
    dx = ops.conv_f32_dgrad(dy, w, x.shape, st, pd)
    dxa = ops.conv_f32_dgrad(dy[:41].contiguous(memory_format=torch.channels_last), w, (41, Cin, H, W), st, pd)
    assert torch.equal(dx[:41], dxa)
    del dx, dxa
    dw = torch.zeros_like(w); ops.conv_f32_wgrad(dy, x, dw, st, pd)
    dwa = torch.zeros_like(w); ops.conv_f32_wgrad(dy[:41].contiguous(memory_format=torch.channels_last), x[:41].contiguous(memory_format=torch.channels_last), dwa, st, pd)
    dwb = torch.zeros_like(w); ops.conv_f32_wgrad(dy[41:].contiguous(memory_format=torch.channels_last), x[41:].contiguous(memory_format=torch.channels_last), dwb, st, pd)
    assert (dw - (dwa + dwb)).abs().max().item() <= 1e-4 * dw.abs().max().item()


def test_resnet_fp32_forward_backward_above_668_rows_launches_no_library_convolution():
    """A 700-row fp32 forward + backward of ResNet-18 at 224 x 224 (the stem's output alone is 2.25 GB): every convolution runs on liblecone's kernels -- the
    library-launch counter of resnet.py stays at zero -- and the result is finite."""
    from learning_embeddings_amd import resnet as R
    torch.manual_seed(0)
    from learning_embeddings_amd import parallel
    net = resnet18(10).to(DEV).to(memory_format=torch.channels_last).train()
    arena = parallel.FlatArena(net.parameters(), DEV)               # (as every trainer / engine has: gradients land in the arena's fp32 slots)
    ov = WgradOverlap(None, arena, side_stream=False); net.wgrad_overlap = ov
    x = _cl(torch.rand(700, 3, 224, 224))
    R.library_launches(reset=True)
    out = net(x)
    out.square().mean().backward()
    torch.cuda.synchronize()
    assert sum(R.library_launches().values()) == 0, R.library_launches()
    assert torch.isfinite(out).all() and all(torch.isfinite(p.grad).all() for p in net.parameters())
