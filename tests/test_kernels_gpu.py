"""Parity of the HIP kernels (through the C ABI) against the reference-generated fixtures and the pinned oracle.
Tolerances: 1e-4 abs on cone energies (north star); gradients relative to the row's gradient scale; integer index
work bit-exact."""
import os
import numpy as np
import pytest
import torch
from conftest import GOLDEN
from oracle import cone_oracle as O

pytestmark = pytest.mark.gpu

from learning_embeddings_amd import ops, _lib  # noqa: E402
from learning_embeddings_amd.hierarchy import NegativeGraph, SyntheticLabelMap  # noqa: E402

DEV = 'cuda'


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def T(a, dtype=torch.float32):
    return torch.tensor(np.asarray(a), dtype=dtype, device=DEV)


def rowrel(g, r):
    g = np.asarray(g, np.float64); r = np.asarray(r, np.float64)
    return np.abs(g - r) / (np.abs(r).max(axis=-1, keepdims=True) + 1e-3)


# ---------------------------------------------------------------------------------------------- F1: cone energy
@pytest.mark.parametrize('D', [2, 10, 128])
def test_cone_energy_vs_reference_fixture(D):
    f = load('F1_cone_energy.npz'); K = float(f['K'])
    x = T(f['x_%d' % D]).requires_grad_(True); y = T(f['y_%d' % D]).requires_grad_(True)
    E = ops.pair_energy(x, y, K)
    (E * T(f['gE_%d' % D])).sum().backward()
    ref, ref64 = f['E_%d' % D], f['E64_%d' % D]
    tol = np.maximum(1e-4, 4 * np.abs(ref - ref64))
    assert (np.abs(E.detach().cpu().numpy() - ref) <= tol).all()
    for g, r, r64 in ((x.grad, f['gx_%d' % D], f['gx64_%d' % D]), (y.grad, f['gy_%d' % D], f['gy64_%d' % D])):
        scale = np.abs(r).max(axis=1, keepdims=True) + 1e-3
        noise = np.abs(r - r64).max(axis=1, keepdims=True) / scale
        assert (np.abs(g.cpu().numpy() - r) / scale <= np.maximum(2e-3, 4 * noise)).all()


@pytest.mark.parametrize('D', [1, 3, 10, 16, 17, 64, 100, 128, 300])
def test_cone_energy_vs_oracle_random(D):
    rs = np.random.RandomState(D)
    P = 1000
    x = rs.randn(P, D).astype(np.float32); y = rs.randn(P, D).astype(np.float32)
    x *= (rs.uniform(0.1, 0.99, (P, 1)) / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
    y *= (rs.uniform(0.1, 1.3, (P, 1)) / np.linalg.norm(y, axis=1, keepdims=True)).astype(np.float32)
    gE = rs.rand(P).astype(np.float32)
    xt = T(x).requires_grad_(True); yt = T(y).requires_grad_(True)
    E = ops.pair_energy(xt, yt, 0.1)
    (E * T(gE)).sum().backward()
    E64 = O.cone_energy(x, y, 0.1, np.float64)
    E32 = O.cone_energy(x, y, 0.1)
    tol = np.maximum(1e-4, 4 * np.abs(E32 - E64))
    assert (np.abs(E.detach().cpu().numpy() - E32) <= tol).all()
    gx, gy = O.cone_energy_grad(x, y, gE, 0.1)
    ok = np.abs(E32 - E64) < 1e-5                                           # well-conditioned rows
    # d(acos)/da = 1/sqrt(1-a^2) amplifies fp32 rounding of `a` near the clamp (1-a^2 >= 2e-5): a few ulp of `a` move the
    # derivative by ~1e-6/(1-a^2) relative.  The reference's own fp32 autograd carries the same noise.
    t = O._cone_terms(x, y, 0.1, np.float64)
    tol = (2e-3 + 2e-6 / np.maximum(1 - t['ac'] ** 2, 2e-5))[:, None]
    assert (rowrel(xt.grad.cpu().numpy(), gx) <= tol)[ok].all()
    assert (rowrel(yt.grad.cpu().numpy(), gy) <= tol)[ok].all()


def test_cone_energy_shapes_and_empty():
    x = torch.rand(4, 7, 10, device=DEV) * 0.3; y = torch.rand(4, 7, 10, device=DEV) * 0.3
    E = ops.pair_energy(x, y, 0.1)
    assert E.shape == (4, 7)
    assert np.abs(E.cpu().numpy() - O.cone_energy(x.cpu().numpy(), y.cpu().numpy(), 0.1)).max() < 1e-4
    assert ops.pair_energy(torch.zeros(0, 10, device=DEV), torch.zeros(0, 10, device=DEV)).shape == (0,)
    # coincident points: 0/0 or tiny/0 at oe_h.py:823 depending on rounding -> NaN or a clamped angle, never a crash
    z = torch.full((2, 10), 0.2, device=DEV)
    e = ops.pair_energy(z, z.clone(), 0.1)
    assert (torch.isnan(e) | ((e >= 0) & (e <= 3.2))).all()


def test_energy_matrix_matches_pairwise():
    rs = np.random.RandomState(0)
    for D, N, M in ((10, 723, 50), (2, 33, 7), (128, 300, 40)):
        lab = (rs.randn(N, D) * 0.2).astype(np.float32); img = (rs.randn(M, D) * 0.3).astype(np.float32)
        E = ops.energy_matrix(T(lab), T(img), 0.1).cpu().numpy()
        want = O.cone_energy(np.repeat(lab[None], M, 0), np.repeat(img[:, None], N, 1), 0.1)
        ref64 = O.cone_energy(np.repeat(lab[None], M, 0), np.repeat(img[:, None], N, 1), 0.1, np.float64)
        assert (np.abs(E - want) <= np.maximum(1e-4, 4 * np.abs(want - ref64))).all()


# ---------------------------------------------------------------------------------------------- F2/F3: projections
def test_label_projection_vs_reference_fixture():
    f = load('F2_embedder.npz'); K = float(f['K'])
    W = T(f['W']).requires_grad_(True)
    out = ops.LabelProjectFn.apply(W, T(f['idx'], torch.int64), K)
    assert np.abs(out.detach().cpu().numpy() - f['out']).max() < 2e-6
    (out * T(f['gout'])).sum().backward()
    assert np.abs(W.grad.cpu().numpy() - f['gW']).max() / np.abs(f['gW']).max() < 1e-5


def test_image_softclip_vs_reference_fixture():
    f = load('F3_image_proj.npz'); K = float(f['K'])
    raw = T(f['raw']).requires_grad_(True)
    out = ops.ImageSoftClipFn.apply(raw, K)
    ref = f['soft_clip']
    assert (np.abs(out.detach().cpu().numpy() - ref) / (1e-6 + np.abs(ref))).max() < 1e-5
    (out * T(f['gout'])).sum().backward()
    assert rowrel(raw.grad.cpu().numpy(), f['graw']).max() < 1e-4


# ---------------------------------------------------------------------------------------------- F5: fused criterion
def _codes(ix, N):
    ix = np.asarray(ix, np.int64)
    return torch.tensor(np.where(ix < N, ix, -1 - (ix - N)), dtype=torch.int32, device=DEV)


def run_joint(W, R, frm, to, neg, alpha, K, weights=None, energy='hyp_cone'):
    N = W.shape[0]
    Wt = T(W).requires_grad_(True)
    Rt = T(R).requires_grad_(True) if R is not None and len(R) else None
    lab, img = {'hyp_cone': (_lib.LABEL_HYP, _lib.IMAGE_SOFTCLIP), 'order': (_lib.LABEL_RAW, _lib.IMAGE_RAW),
                'euc_cone': (_lib.LABEL_SOFTCLIP_K, _lib.IMAGE_SOFTCLIP_K)}[energy]
    loss, e_pos, e_neg = ops.JointLossFn.apply(
        Wt, Rt, _codes(frm, N), _codes(to, N), _codes(neg, N).reshape(len(frm), -1).contiguous(),
        T(weights) if weights is not None else None, 0.0 if K is None else K, alpha, ops.ENERGY[energy], lab, img)
    loss.backward()
    return (float(loss), e_pos.cpu().numpy(), e_neg.cpu().numpy(), Wt.grad.cpu().numpy(),
            Rt.grad.cpu().numpy() if Rt is not None else None)


@pytest.mark.parametrize('tag', ['s3', 'ethec'])
def test_joint_loss_vs_reference_fixture(tag):
    f = load('F5_criterion.npz')
    g = lambda k: f[tag + '_' + k]
    # negatives from the C++ sampler (bit-exact) -> fused kernel -> the reference's loss / energies / gradients
    lm = SyntheticLabelMap(g('levels').tolist(), edges=[tuple(e) for e in g('edges').tolist()])
    ng = NegativeGraph.from_labelmap(lm, n_images=int(g('n_images')), pick_per_level=bool(g('pick_per_level')), seed=0)
    neg = ng.draw_batch(g('from'), g('to'), int(g('Kneg')))
    assert np.array_equal(neg, g('neg'))
    loss, e_pos, e_neg, gW, gR = run_joint(g('W'), g('R'), g('from'), g('to'), neg, float(g('alpha')), float(g('K')))
    assert np.abs(e_pos - g('e_pos')).max() <= 1e-4
    assert np.abs(e_neg - g('e_neg')[..., 0]).max() <= 1e-4
    assert abs(loss - float(g('loss'))) <= 1e-4 * max(1.0, abs(float(g('loss'))))
    assert np.abs(gW - g('gW')).max() / np.abs(g('gW')).max() < 1e-3
    assert np.abs(gR - g('gR')).max() / np.abs(g('gR')).max() < 1e-3


@pytest.mark.parametrize('B,K,D,M', [(1, 0, 10, 4), (7, 1, 2, 5), (64, 5, 10, 40), (33, 37, 16, 64), (16, 256, 10, 32),
                                     (20, 3, 64, 16), (12, 9, 128, 16), (5, 2, 300, 8)])
def test_joint_loss_vs_oracle_random(B, K, D, M):
    rs = np.random.RandomState(B * 1000 + K)
    N = 500
    W = rs.randn(N, D).astype(np.float32)
    W *= (rs.uniform(0.1, 0.6, (N, 1)) / np.linalg.norm(W, axis=1, keepdims=True)).astype(np.float32)
    R = (rs.randn(M, D) * 0.3).astype(np.float32)
    frm = rs.randint(0, N, B); to = np.where(rs.rand(B) < 0.7, N + rs.randint(0, M, B), rs.randint(0, N, B))
    neg = np.where(rs.rand(B, 2 * K) < 0.8, rs.randint(0, N, (B, 2 * K)), N + rs.randint(0, M, (B, 2 * K)))
    # never pair a node with itself (the reference's A has a zero diagonal: cannot be sampled)
    for b in range(B):
        for k in range(2 * K):
            other = frm[b] if k < K else to[b]
            while neg[b, k] == other:
                neg[b, k] = rs.randint(0, N)
        while frm[b] == to[b]:
            frm[b] = rs.randint(0, N)
    w = rs.uniform(0.5, 2.0, B).astype(np.float32)
    alpha = 1.5                                                              # keep many hinges alive
    loss, e_pos, e_neg, gW, gR = run_joint(W, R, frm, to, neg, alpha, 0.1, weights=w)
    o_loss, o_pos, o_neg, o_gW, o_gR = O.joint_loss_fwd_bwd(W, R, frm, to, neg, alpha, 0.1, weights=w)
    assert np.abs(e_pos - o_pos).max() <= 1e-4
    if K:
        assert np.abs(e_neg - o_neg).max() <= 1e-4
    assert abs(loss - o_loss) <= 1e-4 * max(1.0, abs(o_loss))
    assert np.abs(gW - o_gW).max() / (np.abs(o_gW).max() + 1e-12) < 2e-3
    assert np.abs(gR - o_gR).max() / (np.abs(o_gR).max() + 1e-12) < 2e-3


def test_joint_loss_forward_only_and_determinism():
    rs = np.random.RandomState(3)
    N, D, B, K = 100, 10, 50, 5
    W = T((rs.randn(N, D) * 0.1).astype(np.float32))
    frm = torch.tensor(rs.randint(0, 50, B), dtype=torch.int32, device=DEV)
    to = torch.tensor(rs.randint(50, N, B), dtype=torch.int32, device=DEV)
    neg = torch.tensor(rs.randint(0, N, (B, 2 * K)), dtype=torch.int32, device=DEV)
    neg[:, :K] += (neg[:, :K] == frm[:, None]).int(); neg[:, K:] -= (neg[:, K:] == to[:, None]).int()
    neg = neg.clamp_(0, N - 1).contiguous()
    outs = [ops.joint_loss_raw(W, None, frm, to, neg, None, 0.1, 0.5, 0, 1, 1)[0].item() for _ in range(5)]
    assert len(set(outs)) == 1                                               # loss reduction is order-deterministic
    gt = torch.zeros_like(W)
    l2 = ops.joint_loss_raw(W, None, frm, to, neg, None, 0.1, 0.5, 0, 1, 1, gt, None)[0].item()
    assert l2 == outs[0] and gt.abs().sum().item() > 0


# ---------------------------------------------------------------------------------------------- F6: table step
@pytest.mark.parametrize('D', [10, 2])
def test_table_step_vs_reference_fixture(D):
    f = load('F6_table_step.npz'); K = float(f['K'])
    W = T(f['adam_W0_%d' % D]); m = torch.zeros_like(W); v = torch.zeros_like(W)
    for step in range(3):
        ops.table_step_adam(W, T(f['adam_grads_%d' % D][step]), m, v, step + 1, float(f['lr_adam']), K)
        assert np.abs(W.cpu().numpy() - f['adam_W_%d' % D][step]).max() < 2e-6
    assert np.abs(m.cpu().numpy() - f['adam_m_%d' % D]).max() < 1e-6
    assert np.abs(v.cpu().numpy() - f['adam_v_%d' % D]).max() < 1e-6
    Wr = T(f['rsgd_W0_%d' % D])
    ops.table_step_rsgd(Wr, T(f['rsgd_grad_%d' % D]), float(f['lr_rsgd']), K)
    assert np.abs(Wr.cpu().numpy() - f['rsgd_W_%d' % D]).max() < 2e-6


@pytest.mark.parametrize('N,D', [(50000, 10), (1000, 128), (7, 300)])
def test_table_step_vs_oracle(N, D):
    rs = np.random.RandomState(N)
    W = rs.randn(N, D).astype(np.float32)
    W *= (rs.uniform(0.05, 1.2, (N, 1)) / np.linalg.norm(W, axis=1, keepdims=True)).astype(np.float32)
    g = (rs.randn(N, D) * (rs.rand(N, 1) < 0.5)).astype(np.float32)
    m0 = (rs.randn(N, D) * 0.01).astype(np.float32); v0 = (rs.rand(N, D) * 0.01).astype(np.float32)
    Wt, mt, vt = T(W), T(m0), T(v0)
    ops.table_step_adam(Wt, T(g), mt, vt, 4, 3e-3, 0.1)
    Wo, mo, vo = O.table_step_adam(W, g, m0, v0, 4, 3e-3, 0.1)
    assert np.abs(Wt.cpu().numpy() - Wo).max() < 5e-6 and np.abs(mt.cpu().numpy() - mo).max() < 1e-6
    assert np.abs(vt.cpu().numpy() - vo).max() < 1e-6
    n = np.linalg.norm(Wt.cpu().numpy(), axis=1)
    assert n.min() >= O.inner_radius(0.1) - 1e-6 and n.max() <= 1.0          # the clip invariant


def test_adam_flat_vs_torch_adam():
    torch.manual_seed(0)
    n = 1000003
    p = torch.randn(n, device=DEV); g1 = torch.randn(n, device=DEV); g2 = torch.randn(n, device=DEV)
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3)
    m = torch.zeros_like(p); v = torch.zeros_like(p)
    for step, g in enumerate((g1, g2), 1):
        ref.grad = g.clone(); opt.step()
        ops.adam_flat(p, g, m, v, step, 1e-3)
    assert (p - ref.detach()).abs().max().item() < 1e-6


# ---------------------------------------------------------------------------------------------- F7: Euclidean order embeddings (config 1)
def test_order_energy_and_toy_criterion_vs_reference_fixture():
    f = load('F7_order_embedding.npz')
    x = T(f['x']).requires_grad_(True); y = T(f['y']).requires_grad_(True)
    E = ops.pair_energy(x, y, None, 'order')
    (E * T(f['gE'])).sum().backward()
    assert np.abs(E.detach().cpu().numpy() - f['E']).max() < 1e-5
    assert np.abs(x.grad.cpu().numpy() - f['gx']).max() < 1e-5 and np.abs(y.grad.cpu().numpy() - f['gy']).max() < 1e-5
    for tag in ('toy2', 'toy3'):
        g = lambda k: f[tag + '_' + k]
        lm = SyntheticLabelMap(g('levels').tolist(), edges=[tuple(e) for e in g('edges').tolist()])
        ng = NegativeGraph.from_labelmap(lm, pick_per_level=True, labels_only=True, seed=0)
        neg = ng.draw_batch(g('from'), g('to'), 4)
        assert np.array_equal(neg, g('neg'))
        loss, e_pos, e_neg, gW, _ = run_joint(g('W'), None, g('from'), g('to'), neg, 1.0, None, energy='order')
        assert np.abs(e_pos - g('e_pos')).max() < 1e-5 and np.abs(e_neg.reshape(-1) - g('e_neg')).max() < 1e-5
        assert abs(loss - float(g('loss'))) < 1e-4 * abs(float(g('loss')))
        assert np.abs(gW - g('gW')).max() < 1e-4


# ---------------------------------------------------------------------------------------------- F8: multi-level CE (config 4)
@pytest.mark.parametrize('tag,w', [('unw', None), ('w', 'level_weights_w')])
def test_multilevel_ce_vs_reference_fixture(tag, w):
    f = load('F8_multilevel_ce.npz')
    z = T(f['logits']).requires_grad_(True)
    loss = ops.MultiLevelCEFn.apply(z, T(f['level_labels'], torch.int64), f['levels'].tolist(), None if w is None else f[w].tolist())
    loss.backward()
    assert abs(loss.item() - float(f[tag + '_loss'])) < 1e-5 * abs(float(f[tag + '_loss']))
    assert np.abs(z.grad.cpu().numpy() - f[tag + '_glogits']).max() < 1e-6


@pytest.mark.parametrize('tag,w', [('unw', None), ('w', 'level_weights_w')])
def test_multilevel_ce_class_weights_vs_reference_fixture(tag, w):
    """MultiLevelCELoss(labelmap, level_weights, weight=per-class weights) (loss.py:16-25) through the host mirror, against the
    reference's own loss and gradient (fixture F8b)."""
    from learning_embeddings_amd.loss import MultiLevelCELoss
    from learning_embeddings_amd.hierarchy import SyntheticLabelMap
    f = load('F8b_multilevel_ce_class_weights.npz')
    crit = MultiLevelCELoss(SyntheticLabelMap(f['levels'].tolist()), level_weights=None if w is None else f[w].tolist(), weight=torch.from_numpy(f['class_weights']))
    z = T(f['logits']).requires_grad_(True)
    loss = crit(z, None, T(f['level_labels'], torch.int64))
    loss.backward()
    assert abs(loss.item() - float(f[tag + '_loss'])) < 1e-5 * abs(float(f[tag + '_loss']))
    assert np.abs(z.grad.cpu().numpy() - f[tag + '_glogits']).max() < 1e-6


def test_multilevel_ce_batch512():
    rs = np.random.RandomState(0)
    levels = [6, 21, 135, 561]
    z = (rs.randn(512, 723) * 4).astype(np.float32)
    lab = np.stack([rs.randint(0, n, 512) for n in levels], 1)
    zt = T(z).requires_grad_(True)
    loss = ops.MultiLevelCEFn.apply(zt, T(lab, torch.int64), levels, None)
    loss.backward()
    ol, og = O.multilevel_ce(z, lab, levels)
    assert abs(loss.item() - ol) < 1e-5 * ol and np.abs(zt.grad.cpu().numpy() - og).max() < 1e-6


# ---------------------------------------------------------------------------------------------- full-size properties
@pytest.mark.parametrize('B,K,D,N', [(256, 5, 10, 2000), (256, 256, 10, 50000)])
def test_joint_loss_full_size_properties(B, K, D, N):
    """At BASELINE.json's sizes (config 3: B=256, K=5; config 5: K=256, 50k labels) the oracle is too slow to run per
    element, so the kernel is checked through size-independent properties:
      * checksum: loss == sum_b e_pos + sum max(0, alpha - e_neg) from the kernel's own energy outputs;
      * permutation: shuffling the positives (with their negatives) leaves the loss and the gradients unchanged;
      * linearity in the weights: loss(w) with w = 2 doubles loss and gradients;
      * locality: rows that appear in no pair get exactly zero gradient;
      * sampled oracle: 64 random pairs agree with the oracle to 1e-4."""
    rs = np.random.RandomState(B + K)
    M = B
    W = rs.randn(N, D).astype(np.float32); W *= (rs.uniform(0.1, 0.5, (N, 1)) / np.linalg.norm(W, axis=1, keepdims=True)).astype(np.float32)
    R = (rs.randn(M, D) * 0.3).astype(np.float32)
    used = rs.choice(N, N // 2, replace=False)                       # only half of the labels ever appear
    frm = used[rs.randint(0, len(used), B)]; to = N + rs.permutation(M)[:B]
    neg = used[rs.randint(0, len(used), (B, 2 * K))]
    clash = neg[:, :K] == frm[:, None]                                # (u, u) cannot be sampled: A has a zero diagonal
    while clash.any():
        neg[:, :K][clash] = used[rs.randint(0, len(used), int(clash.sum()))]; clash = neg[:, :K] == frm[:, None]
    alpha = 1.2
    def run(order, w=None):
        Wt = T(W).requires_grad_(True); Rt = T(R).requires_grad_(True)
        loss, e_pos, e_neg = ops.JointLossFn.apply(Wt, Rt, _codes(frm[order], N), _codes(to[order], N),
                                                   _codes(neg[order], N).reshape(B, -1).contiguous(),
                                                   None if w is None else T(w[order]), 0.1, alpha, 0, 1, 1)
        loss.backward()
        return loss.item(), e_pos.cpu().numpy(), e_neg.cpu().numpy(), Wt.grad.cpu().numpy(), Rt.grad.cpu().numpy()
    ident = np.arange(B)
    l0, ep, en, gW, gR = run(ident)
    chk = ep.astype(np.float64).sum() + np.maximum(alpha - en.astype(np.float64), 0).sum()
    assert abs(l0 - chk) <= 1e-4 * abs(chk)
    perm = rs.permutation(B)
    l1, ep1, en1, gW1, gR1 = run(perm)
    assert abs(l1 - l0) <= 1e-5 * abs(l0) and np.array_equal(ep1, ep[perm]) and np.array_equal(en1, en[perm])
    assert np.abs(gW1 - gW).max() <= 1e-4 * np.abs(gW).max() and np.abs(gR1 - gR).max() <= 1e-4 * np.abs(gR).max()
    l2, _, _, gW2, gR2 = run(ident, np.full(B, 2.0, np.float32))
    assert abs(l2 - 2 * l0) <= 1e-5 * abs(l0) and np.abs(gW2 - 2 * gW).max() <= 1e-4 * np.abs(gW).max()
    untouched = np.setdiff1d(np.arange(N), used)
    assert (gW[untouched] == 0).all()
    # a sample of pairs against the oracle
    bs = rs.randint(0, B, 64); ks = rs.randint(0, 2 * K, 64)
    xs = np.where(ks < K, frm[bs], neg[bs, ks]); ys = np.where(ks < K, neg[bs, ks], to[bs])
    def emb(ix):
        out = np.zeros((len(ix), D), np.float32); lab = ix < N
        out[lab] = O.embedder_forward(W, ix[lab], 0.1); out[~lab] = O.image_soft_clip(R[ix[~lab] - N], 0.1)
        return out
    want = O.cone_energy(emb(xs), emb(ys), 0.1)
    assert np.abs(en[bs, ks] - want).max() <= 1e-4


def test_table_step_full_size_invariants():
    """config-5 table (50 000 x 10) and a 1M x 128 table: after the step every row norm lies in [r_in, 1 - 1e-5] (the clip
    invariant of oe_h.py:1604-1617), rows with zero gradient and zero moments do not move, and the step is idempotent on a
    zero gradient with zero moments."""
    for N, D in ((50000, 10), (1000000, 128)):
        g = torch.Generator(device='cpu').manual_seed(N)
        W = torch.randn(N, D, generator=g); W = (W / W.norm(dim=1, keepdim=True) * (0.12 + 0.8 * torch.rand(N, 1, generator=g))).to(DEV)
        grad = torch.randn(N, D, generator=g).to(DEV); grad[::2] = 0
        m = torch.zeros_like(W); v = torch.zeros_like(W)
        W0 = W.clone()
        ops.table_step_adam(W, grad, m, v, 1, 1e-3, 0.1)
        n = W.norm(dim=1)
        assert n.min().item() >= O.inner_radius(0.1) - 1e-6 and n.max().item() <= 1.0
        assert torch.equal(W[::2], W0[::2])                              # untouched rows (already inside the shell) are bit-identical
        assert (W[1::2] != W0[1::2]).any()


# ---------------------------------------------------------------------------------------------- F10: Euclidean cones (oe.py)
@pytest.mark.parametrize('D', [2, 10, 128])
def test_euclidean_cone_energy_vs_reference_fixture(D):
    f = load('F10_euclidean_cone.npz'); K = float(f['K'])
    x = T(f['x%d' % D]).requires_grad_(True); y = T(f['y%d' % D]).requires_grad_(True)
    E = ops.pair_energy(x, y, K, 'euc_cone')
    (E * T(f['gE%d' % D])).sum().backward()
    ref, ref64 = f['E%d' % D], f['E64_%d' % D]
    assert (np.abs(E.detach().cpu().numpy() - ref) <= np.maximum(1e-4, 4 * np.abs(ref - ref64))).all()
    for g, r, r64 in ((x.grad, f['gx%d' % D], f['gx64_%d' % D]), (y.grad, f['gy%d' % D], f['gy64_%d' % D])):
        scale = np.abs(r).max(axis=1, keepdims=True) + 1e-3
        noise = np.abs(r - r64).max(axis=1, keepdims=True) / scale            # the reference's own fp32 noise (psi ~ 0 rows)
        assert (np.abs(g.cpu().numpy() - r) / scale <= np.maximum(1e-3, 4 * noise)).all()
    # all-pairs form against the pairwise form
    Em = ops.energy_matrix(x.detach()[:40], y.detach()[:30], K, 'euc_cone').cpu().numpy()
    want = O.euc_cone_energy(np.repeat(f['x%d' % D][None, :40], 30, 0), np.repeat(f['y%d' % D][:30, None], 40, 1), K)
    assert np.abs(Em - want).max() <= 1e-4


def test_euclidean_projections_vs_reference_fixture():
    f = load('F10_euclidean_cone.npz'); K = float(f['K'])
    W = T(f['emb_W']).requires_grad_(True)
    out = ops.LabelProjectFn.apply(W, T(f['emb_idx'], torch.int64), K, _lib.LABEL_SOFTCLIP_K)
    assert np.abs(out.detach().cpu().numpy() - f['emb_out']).max() < 5e-6
    (out * T(f['emb_gout'])).sum().backward()
    assert np.abs(W.grad.cpu().numpy() - f['emb_gW']).max() / np.abs(f['emb_gW']).max() < 1e-5
    raw = T(f['img_raw']).requires_grad_(True)
    o2 = ops.ImageSoftClipFn.apply(raw, K, _lib.IMAGE_SOFTCLIP_K)
    assert np.abs(o2.detach().cpu().numpy() - f['img_out']).max() < 5e-6
    (o2 * T(f['img_gout'])).sum().backward()
    assert rowrel(raw.grad.cpu().numpy(), f['img_graw']).max() < 1e-4


def test_euclidean_joint_loss_vs_reference_fixture():
    f = load('F10_euclidean_cone.npz'); K = float(f['K'])
    g = lambda k: f['c_' + k]
    lm = SyntheticLabelMap(g('levels').tolist(), edges=[tuple(e) for e in g('edges').tolist()])
    ng = NegativeGraph.from_labelmap(lm, n_images=int(g('n_images')), pick_per_level=True, seed=0)
    neg = ng.draw_batch(g('from'), g('to'), int(g('Kneg')))
    assert np.array_equal(neg, g('neg'))                                     # oe.py:755-808 draws like oe_h.py:849-902
    loss, e_pos, e_neg, gW, gR = run_joint(g('W'), g('R'), g('from'), g('to'), neg, float(g('alpha')), K, energy='euc_cone')
    assert np.abs(e_pos - g('e_pos')).max() <= 1e-4
    assert np.abs(e_neg - g('e_neg')[..., 0]).max() <= 1e-4
    assert abs(loss - float(g('loss'))) <= 1e-4 * max(1.0, abs(float(g('loss'))))
    assert np.abs(gW - g('gW')).max() / np.abs(g('gW')).max() < 1e-3
    assert np.abs(gR - g('gR')).max() / np.abs(g('gR')).max() < 1e-3


@pytest.mark.parametrize('B,K,D,M', [(7, 1, 2, 5), (64, 5, 10, 40), (16, 256, 10, 32), (12, 9, 128, 16), (5, 2, 300, 8)])
def test_euclidean_joint_loss_vs_oracle_random(B, K, D, M):
    rs = np.random.RandomState(B * 1000 + K + 7)
    N = 500
    W = rs.randn(N, D).astype(np.float32); R = (rs.randn(M, D) * 0.3).astype(np.float32)
    frm = rs.randint(0, N, B); to = np.where(rs.rand(B) < 0.7, N + rs.randint(0, M, B), rs.randint(0, N, B))
    neg = np.where(rs.rand(B, 2 * K) < 0.8, rs.randint(0, N, (B, 2 * K)), N + rs.randint(0, M, (B, 2 * K)))
    for b in range(B):
        while frm[b] == to[b]:
            frm[b] = rs.randint(0, N)
        for k in range(2 * K):
            other = frm[b] if k < K else to[b]
            while neg[b, k] == other:
                neg[b, k] = rs.randint(0, N)
    w = rs.uniform(0.5, 2.0, B).astype(np.float32)
    loss, e_pos, e_neg, gW, gR = run_joint(W, R, frm, to, neg, 1.6, 3.0, weights=w, energy='euc_cone')
    o_loss, o_pos, o_neg, o_gW, o_gR = O.joint_loss_fwd_bwd(W, R, frm, to, neg, 1.6, 3.0, weights=w, energy='euc_cone')
    assert np.abs(e_pos - o_pos).max() <= 1e-4 and np.abs(e_neg - o_neg).max() <= 1e-4
    assert abs(loss - o_loss) <= 1e-4 * max(1.0, abs(o_loss))
    assert np.abs(gW - o_gW).max() / (np.abs(o_gW).max() + 1e-12) < 2e-3
    assert np.abs(gR - o_gR).max() / (np.abs(o_gR).max() + 1e-12) < 2e-3


# ---------------------------------------------------------------------------------------------- 8f-1: fused scoring + per-level top-k
@pytest.mark.parametrize('energy,K,D,levels,M', [('hyp_cone', 0.1, 10, [6, 21, 135, 561], 300), ('hyp_cone', 0.1, 2, [2, 3, 9], 70),
                                                  ('euc_cone', 3.0, 10, [8, 64, 384], 129), ('order', None, 16, [5, 40], 64),
                                                  ('hyp_cone', 0.1, 100, [4, 30, 200], 65), ('hyp_cone', 0.1, 10, [2, 8, 32, 128, 512, 2048, 8192], 200)])
def test_level_topk_vs_bruteforce(energy, K, D, levels, M):
    rs = np.random.RandomState(D + M)
    N = sum(levels)
    lab = (rs.randn(N, D) * (0.2 if energy == 'hyp_cone' else 1.0)).astype(np.float32)
    img = (rs.randn(M, D) * (0.3 if energy == 'hyp_cone' else 1.0)).astype(np.float32)
    if energy == 'euc_cone':
        lab = O.soft_clip_add(lab, K); img = O.soft_clip_add(img, K)
    starts = np.concatenate([[0], np.cumsum(levels)])
    k = 5
    idx, val = ops.level_topk(T(lab), T(img), starts, k, K, energy)
    idx = idx.cpu().numpy(); val = val.cpu().numpy()
    X = np.repeat(lab[None], M, 0); Y = np.repeat(img[:, None], N, 1)
    fn = {'hyp_cone': lambda dt: O.cone_energy(X, Y, K, dt), 'euc_cone': lambda dt: O.euc_cone_energy(X, Y, K, dt),
          'order': lambda dt: O.order_energy(X, Y, dt)}[energy]
    E = fn(np.float32); E64 = fn(np.float64)
    Em = ops.energy_matrix(T(lab), T(img), K, energy).cpu().numpy()
    for l, n in enumerate(levels):
        s, e = starts[l], starts[l + 1]
        kk = min(k, n)
        order = np.argsort(E[:, s:e], axis=1, kind='stable')[:, :kk] + s
        want = np.take_along_axis(E, order, 1)
        got_i, got_v = idx[:, l, :kk], val[:, l, :kk]
        assert (got_i >= s).all() and (got_i < e).all()
        assert (np.diff(got_v, axis=1) >= 0).all()                               # ascending
        tol = np.maximum(1e-4, 4 * np.abs(E - E64).max())
        assert np.abs(got_v - want).max() <= tol                                 # the k smallest energies, to the parity bar
        assert np.abs(np.take_along_axis(Em, got_i.astype(np.int64), 1) - got_v).max() <= 1e-5   # value belongs to the index
        for i in range(M):
            assert len(set(got_i[i].tolist())) == kk                             # no label twice
        # where the oracle's ranking is unambiguous (gaps above the tolerance) the indices are the same
        srt = np.sort(E[:, s:e], axis=1)[:, :min(kk + 1, n)]
        clear = (np.diff(srt, axis=1) > 2 * tol).all(axis=1) if srt.shape[1] > 1 else np.ones(M, bool)
        assert np.array_equal(got_i[clear], order[clear])
        if n < k:
            assert (idx[:, l, n:] == -1).all() and np.isinf(val[:, l, n:]).all()


def test_level_topk_empty_and_argument_errors():
    lab = torch.rand(10, 4, device=DEV) * 0.3; img = torch.rand(0, 4, device=DEV)
    idx, val = ops.level_topk(lab, img, [0, 4, 10], 3, 0.1)
    assert idx.shape == (0, 2, 3)
    with pytest.raises(_lib.LeconeError):
        ops.level_topk(lab, torch.rand(3, 4, device=DEV), [0, 4, 10], 9, 0.1)    # k > 8
    with pytest.raises(ValueError):
        ops.level_topk(lab, torch.rand(3, 4, device=DEV), [0, 4, 11], 3, 0.1)    # offsets past the table


# ---------------------------------------------------------------------------------------------- MFMA 1x1 convolution (+ BN statistics)
@pytest.mark.parametrize('cin,cout,M', [(64, 256, 32 * 301), (64, 64, 32 * 40), (128, 512, 32 * 97), (128, 256, 32 * 64),
                                        (256, 64, 32 * 129), (256, 128, 32 * 33), (512, 128, 32 * 77), (64, 256, 512 * 56 * 56)])
def test_conv1x1_mfma_vs_fp32_matmul_and_statistics(cin, cout, M):
    """lec_conv1x1_fwd: y = x w^T (bf16 in, fp32 accumulate, bf16 out) against an fp32 matmul, and the per-channel
    sum / sum-of-squares partials it leaves for BatchNorm against sums over the rounded output."""
    assert ops.conv1x1_supported(cin, cout, M) and not ops.conv1x1_supported(cin, cout, M + 1) and not ops.conv1x1_supported(96, 256, M)
    g = torch.Generator(device='cpu').manual_seed(cin * 7 + cout)
    rows = min(M, 32 * 512)
    x = (torch.randn(rows, cin, generator=g) * 0.7).to(DEV).to(torch.bfloat16)
    if M > rows:
        x = x.repeat(M // rows, 1)
    w = (torch.randn(cout, cin, generator=g) * 0.2).to(DEV).to(torch.bfloat16)
    y = ops.conv1x1_rows(x, w, want_stats=True)
    n = ops._BN_WS_OWNER[1]
    assert ops._BN_WS_OWNER[0] == y.data_ptr() and 1 <= n <= 512
    part = ops._bn_workspace(x.device)[:n * 2 * cout * 4].view(torch.float32).view(n, 2, cout).double().sum(0)
    ref = x[:rows].float() @ w.float().t()
    assert (y[:rows].float() - ref).abs().max().item() <= 6e-3 * ref.abs().max().item()      # bf16 rounding of the output
    assert torch.equal(y[:rows], y[-rows:])                                                   # every strip of a long input
    yd = y.float().double()
    assert ((part[0] - yd.sum(0)).abs().max() / (yd.sum(0).abs().max() + 1e-9)).item() < 1e-4
    assert ((part[1] - (yd * yd).sum(0)).abs().max() / (yd * yd).sum(0).abs().max()).item() < 1e-5
    y2 = ops.conv1x1_rows(x, w)                                                               # plain product (the dgrad form)
    assert torch.equal(y, y2)
    y3 = ops.conv1x1_rows(x, w.t().contiguous(), w_transposed=True)                           # weight given as [Cin, Cout]
    assert torch.equal(y, y3)
    ops._BN_WS_OWNER[0] = 0


@pytest.mark.parametrize('N,H,W', [(1, 8, 8), (3, 16, 24), (2, 56, 56), (300, 8, 16)])
def test_conv3x3_c64_wgrad_mfma_vs_torch(N, H, W):
    """lec_conv3x3_c64_wgrad: dw[co][ky][kx][ci] += sum dy * shifted x (bf16 in, fp32 accumulate, float atomics) against autograd
    of an fp32 convolution -- integer-valued data first (exact: bit-equal), then random data; accumulates across calls."""
    g = torch.Generator(device='cpu').manual_seed(N * H + W)
    def nhwc(t):
        return t.to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    def reference(x, dy):
        w = torch.zeros(64, 64, 3, 3, device=DEV, dtype=torch.float64, requires_grad=True)
        torch.nn.functional.conv2d(x.double(), w, padding=1).backward(dy.double())
        return w.grad
    x = nhwc(torch.randint(-3, 4, (N, 64, H, W), generator=g).float()); dy = nhwc(torch.randint(-2, 3, (N, 64, H, W), generator=g).float())
    dw = torch.zeros(64, 64, 3, 3, device=DEV).contiguous(memory_format=torch.channels_last)
    ops.conv3x3_c64_wgrad(dy, x, dw)
    ref = reference(x, dy)
    assert torch.equal(dw.double(), ref)
    ops.conv3x3_c64_wgrad(dy, x, dw)
    assert torch.equal(dw.double(), 2 * ref)
    x = nhwc(torch.randn(N, 64, H, W, generator=g) * 0.7); dy = nhwc(torch.randn(N, 64, H, W, generator=g) * 0.1)
    dw = torch.full((64, 64, 3, 3), 0.25, device=DEV).contiguous(memory_format=torch.channels_last)
    ops.conv3x3_c64_wgrad(dy, x, dw)
    ref = reference(x, dy) + 0.25
    assert ((dw.double() - ref).abs().max() / ref.abs().max()).item() < 2e-5
    with pytest.raises(ValueError):
        ops.conv3x3_c64_wgrad(dy, x, torch.zeros(64, 64, 3, 3, device=DEV))            # not the channels_last layout


@pytest.mark.parametrize('cin,cout,M', [(64, 64, 64), (64, 256, 64 * 5), (256, 64, 64 * 300), (256, 128, 64 * 257), (128, 512, 64 * 513),
                                        (512, 128, 64 * 700), (256, 1024, 64 * 130), (1024, 256, 64 * 70)])
def test_conv1x1_wgrad_mfma_vs_fp32_matmul(cin, cout, M):
    """lec_conv1x1_wgrad: dw += dy^T x (bf16 in, fp32 accumulate, float atomics into the caller's buffer) against an fp32
    matmul -- integer-valued data first (every product and sum exact: the result must be bit-equal), then random data; the
    buffer's previous content is kept (accumulation across calls)."""
    assert ops.conv1x1_wgrad_supported(cin, cout, M) and not ops.conv1x1_wgrad_supported(cin, cout, M + 32) and not ops.conv1x1_wgrad_supported(96, 256, M)
    g = torch.Generator(device='cpu').manual_seed(cin * 3 + cout)
    rows = min(M, 64 * 64)
    xi = torch.randint(-3, 4, (rows, cin), generator=g).float(); dyi = torch.randint(-2, 3, (rows, cout), generator=g).float()
    reps = M // rows; tail = M - reps * rows
    def tile(t):
        return torch.cat([t.repeat(reps, 1), t[:tail]], 0).to(DEV).to(torch.bfloat16)
    x, dy = tile(xi), tile(dyi)
    dw = torch.zeros(cout, cin, device=DEV)
    ops.conv1x1_wgrad_rows(dy, x, dw)
    ref = (dy.float().double().t() @ x.float().double())
    assert torch.equal(dw.double(), ref)                                                       # |sums| < 2^24: exact in fp32
    ops.conv1x1_wgrad_rows(dy, x, dw)
    assert torch.equal(dw.double(), 2 * ref)
    x = tile(torch.randn(rows, cin, generator=g) * 0.7); dy = tile(torch.randn(rows, cout, generator=g) * 0.1)
    dw = torch.full((cout, cin), 0.5, device=DEV)
    ops.conv1x1_wgrad_rows(dy, x, dw)
    ref = dy.float().double().t() @ x.float().double() + 0.5
    assert ((dw.double() - ref).abs().max() / ref.abs().max()).item() < 2e-5
    with pytest.raises(ValueError):
        ops.conv1x1_wgrad_rows(dy, x, dw.half())


@pytest.mark.parametrize('cin,cout,M', [(64, 256, 32 * 70), (128, 512, 32 * 33), (64, 256, 32 * 2100)])
def test_conv1x1_dgrad_with_batchnorm_backward_pass1_in_the_epilogue(cin, cout, M):
    """lec_conv1x1_dgrad_bnfold: g = relu_mask * (dy W + dy2) must be bit-equal to the two-kernel path (data gradient, then pass 1
    of lec_bn_bwd), its partials must sum to (sum g, sum g * xhat), and lec_bn_bwd_prereduced must finish the backward like
    lec_bn_bwd does from the unfused inputs."""
    from learning_embeddings_amd._lib import lib, check, dptr, stream_ptr
    assert ops.conv1x1_dgrad_bnfold_supported(cin, cout, M) and not ops.conv1x1_dgrad_bnfold_supported(256, 64, M)
    g_ = torch.Generator(device='cpu').manual_seed(cin + M)
    gy = (torch.randn(M, cin, generator=g_) * 0.5).to(DEV).to(torch.bfloat16)
    w = (torch.randn(cin, cout, generator=g_) * 0.2).to(DEV).to(torch.bfloat16)          # the layer's forward weight [Cout_fwd = cin][Cin_fwd = cout]
    hw = M // 32
    nhwc = lambda t: t.view(32, hw, 1, cout).permute(0, 3, 1, 2)                         # [N, C, H, W] channels_last view of [M, C] rows
    dy2 = nhwc((torch.randn(M, cout, generator=g_) * 0.5).to(DEV).to(torch.bfloat16))
    xbn = nhwc((torch.randn(M, cout, generator=g_) * 1.3 + 0.4).to(DEV).to(torch.bfloat16))
    mask = torch.randint(0, 256, (M * cout // 8,), generator=g_, dtype=torch.int32).to(torch.uint8).to(DEV)
    mean = (torch.randn(cout, generator=g_) * 0.3).to(DEV); invstd = (torch.rand(cout, generator=g_) + 0.5).to(DEV)
    gamma = (torch.rand(cout, generator=g_) + 0.5).to(DEV)
    entry = {'x': xbn, 'dres': dy2, 'mask': mask, 'mean': mean, 'invstd': invstd}
    g = ops.conv1x1_dgrad_bnfold_rows(gy, w, entry)
    n = ops._BN_WS_OWNER[1]
    assert ops._BN_WS_OWNER[0] == g.data_ptr() and ops._FOLDED == {g.data_ptr(): n}
    ws = ops._bn_workspace(gy.device)
    part = ws[:n * 2 * cout * 4].view(torch.float32).view(n, 2, cout).double().sum(0)
    # the two-kernel path
    dyc = ops.conv1x1_rows(gy, w, w_transposed=True)
    bits = ((mask.view(M, cout // 8, 1).to(torch.int32) >> torch.arange(8, device=DEV, dtype=torch.int32)) & 1).view(M, cout).bool()
    ref = torch.where(bits, dyc.float() + dy2.permute(0, 2, 3, 1).reshape(M, cout).float(), torch.zeros((), device=DEV)).to(torch.bfloat16)
    assert torch.equal(g, ref)
    gd = ref.float().double(); xh = (xbn.permute(0, 2, 3, 1).reshape(M, cout).float().double() - mean.double()) * invstd.double()
    assert ((part[0] - gd.sum(0)).abs().max() / gd.abs().sum(0).max()).item() < 1e-5
    assert ((part[1] - (gd * xh).sum(0)).abs().max() / (gd * xh).abs().sum(0).max()).item() < 1e-5
    # finish the backward from the partials; against lec_bn_bwd on the unfused inputs
    dx = torch.empty_like(xbn); dgam = torch.empty(cout, device=DEV); dbet = torch.empty(cout, device=DEV)
    check(lib.lec_bn_bwd_prereduced(dptr(g), dptr(xbn), M, cout, dptr(gamma), dptr(mean), dptr(invstd), n, dptr(dx), dptr(dgam), dptr(dbet),
                                    dptr(ws), ws.numel(), 0, stream_ptr()))
    dx2 = torch.empty_like(xbn); dres2 = torch.empty_like(xbn); dgam2 = torch.empty_like(dgam); dbet2 = torch.empty_like(dbet)
    check(lib.lec_bn_bwd(dptr(nhwc(dyc)), dptr(dy2), None, dptr(mask), dptr(xbn), M, cout, dptr(gamma), dptr(mean), dptr(invstd), dptr(dx2),
                         dptr(dres2), dptr(dgam2), dptr(dbet2), 1, dptr(ws), ws.numel(), 0, stream_ptr()))
    assert torch.equal(dres2.permute(0, 2, 3, 1).reshape(M, cout), g)
    assert torch.allclose(dgam, dgam2, rtol=1e-4, atol=1e-3) and torch.allclose(dbet, dbet2, rtol=1e-4, atol=1e-3)
    assert (dx.float() - dx2.float()).abs().max().item() <= 2e-2 * dx2.float().abs().max().item()
    assert (dx != dx2).float().mean().item() < 1e-3
    ops._BN_WS_OWNER[0] = 0; ops._FOLDED.clear()


@pytest.mark.parametrize('cin,cout,M', [(64, 256, 32 * 70), (128, 512, 32 * 33), (64, 256, 32 * 2100)])
def test_conv1x1_twice_with_batchnorm_apply_in_the_epilogue(cin, cout, M):
    """lec_conv1x1_stats + lec_bn_fwd_finalize + lec_conv1x1_fwd_bnapply (the convolution run twice, the BatchNorm apply pass inside
    the second run) against lec_conv1x1_fwd + lec_bn_fwd_prestat: y, z = relu(bn(y) + residual), the bitmask, the saved statistics
    and the running statistics must be bit-equal."""
    from learning_embeddings_amd._lib import lib, check, dptr, stream_ptr
    import ctypes as C
    g_ = torch.Generator(device='cpu').manual_seed(cin + M)
    x = (torch.randn(M, cin, generator=g_) * 0.7).to(DEV).to(torch.bfloat16)
    w = (torch.randn(cout, cin, generator=g_) * 0.2).to(DEV).to(torch.bfloat16)
    hw = M // 32
    nhwc = lambda t: t.view(32, hw, 1, cout).permute(0, 3, 1, 2)
    res = nhwc((torch.randn(M, cout, generator=g_) * 0.8).to(DEV).to(torch.bfloat16))
    gam = (torch.rand(cout, generator=g_) + 0.5).to(DEV); bet = torch.randn(cout, generator=g_).to(DEV)
    ws = ops._bn_workspace(x.device)
    outs = []
    for fused in (False, True):
        rm = torch.zeros(cout, device=DEV); rv = torch.ones(cout, device=DEV)
        sm = torch.empty(cout, device=DEV); si = torch.empty(cout, device=DEV)
        z = nhwc(torch.empty(M, cout, device=DEV, dtype=torch.bfloat16)); mask = torch.zeros(M * cout // 8, dtype=torch.uint8, device=DEV)
        if fused:
            y = ops.conv1x1_stats_rows(x, w)
            n = ops._BN_WS_OWNER[1]; ops._DEFERRED.clear(); ops._BN_WS_OWNER[0] = 0
            check(lib.lec_bn_fwd_finalize(M, cout, dptr(gam), dptr(bet), 1e-5, 0.1, dptr(rm), dptr(rv), n, dptr(sm), dptr(si), dptr(ws), ws.numel(), stream_ptr()))
            off = lib.lec_bn_workspace_coeff_offset(cout)
            check(lib.lec_conv1x1_fwd_bnapply(dptr(x), dptr(w), M, cin, cout, C.c_void_p(ws.data_ptr() + off), C.c_void_p(ws.data_ptr() + off + 4 * cout),
                                              dptr(res), dptr(y), dptr(z), dptr(mask), stream_ptr()))
        else:
            y = ops.conv1x1_rows(x, w, want_stats=True)
            n = ops._BN_WS_OWNER[1]; ops._BN_WS_OWNER[0] = 0
            check(lib.lec_bn_fwd_prestat(dptr(nhwc(y)), dptr(res), M, cout, dptr(gam), dptr(bet), 1e-5, 0.1, dptr(rm), dptr(rv), n, dptr(sm), dptr(si),
                                         dptr(z), 1, dptr(mask), dptr(ws), ws.numel(), stream_ptr()))
        outs.append((y.clone(), z.clone(), mask.clone(), sm, si, rm, rv))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    assert outs[0][2].any() and (outs[0][1] == 0).any()


@pytest.mark.parametrize('cin,cout,M', [(64, 256, 64 * 37), (128, 512, 64 * 19), (64, 256, 64 * 1100)])
def test_conv1x1_wgrad_with_batchnorm_backward_pass2_in_its_staging(cin, cout, M):
    """lec_conv1x1_wgrad_bnapply: dx must be bit-equal to pass 2 of lec_bn_bwd (here through lec_bn_bwd_prereduced on the same
    coefficients) and dw must be dx^T x accumulated in fp32."""
    from learning_embeddings_amd._lib import lib, check, dptr, stream_ptr
    import ctypes as C
    assert lib.lec_conv1x1_wgrad_bnapply_supported(cin, cout, M) and not lib.lec_conv1x1_wgrad_bnapply_supported(256, 64, M)
    g_ = torch.Generator(device='cpu').manual_seed(cout + M)
    hw = M // 32
    nhwc = lambda t: t.view(32, hw, 1, t.shape[1]).permute(0, 3, 1, 2)
    g = nhwc((torch.randn(M, cout, generator=g_) * 0.5).to(DEV).to(torch.bfloat16))
    xbn = nhwc((torch.randn(M, cout, generator=g_) * 1.3 + 0.4).to(DEV).to(torch.bfloat16))
    x = (torch.randn(M, cin, generator=g_) * 0.7).to(DEV).to(torch.bfloat16)
    gamma = (torch.rand(cout, generator=g_) + 0.5).to(DEV); mean = (torch.randn(cout, generator=g_) * 0.3).to(DEV)
    invstd = (torch.rand(cout, generator=g_) + 0.5).to(DEV)
    ws = ops._bn_workspace(g.device)
    part = (torch.randn(2, cout, generator=g_) * 0.05 * M).to(DEV)                     # one row of "partials": c1, c2 = part / M
    ws[:2 * cout * 4].view(torch.float32).copy_(part.flatten())
    dx_ref = torch.empty_like(g); dgam = torch.empty(cout, device=DEV); dbet = torch.empty(cout, device=DEV)
    check(lib.lec_bn_bwd_prereduced(dptr(g), dptr(xbn), M, cout, dptr(gamma), dptr(mean), dptr(invstd), 1, dptr(dx_ref), dptr(dgam), dptr(dbet),
                                    dptr(ws), ws.numel(), 0, stream_ptr()))
    off = lib.lec_bn_workspace_coeff_offset(cout)
    dx = torch.empty_like(g); dw = torch.full((cout, cin), 0.5, device=DEV)
    check(lib.lec_conv1x1_wgrad_bnapply(dptr(g), dptr(xbn), dptr(x), M, cin, cout, dptr(gamma), dptr(mean), dptr(invstd),
                                        C.c_void_p(ws.data_ptr() + off), C.c_void_p(ws.data_ptr() + off + 4 * cout), dptr(dx), dptr(dw), stream_ptr()))
    assert torch.equal(dx, dx_ref)
    ref = dx_ref.permute(0, 2, 3, 1).reshape(M, cout).float().double().t() @ x.float().double() + 0.5
    assert ((dw.double() - ref).abs().max() / ref.abs().max()).item() < 2e-5


def test_conv1x1_statistics_feed_batchnorm():
    """conv (MFMA kernel, statistics in the epilogue) -> BatchNorm (no statistics pass) equals conv -> full BatchNorm."""
    g = torch.Generator(device='cpu').manual_seed(5)
    N, H, W, cin, cout = 4, 16, 16, 64, 256
    x = torch.randn(N * H * W, cin, generator=g).to(DEV).to(torch.bfloat16)
    w = (torch.randn(cout, cin, generator=g) * 0.2).to(DEV).to(torch.bfloat16)
    gam = (torch.rand(cout, generator=g) + 0.5).to(DEV); bet = torch.randn(cout, generator=g).to(DEV)
    outs = []
    for fused in (True, False):
        y = ops.conv1x1_rows(x, w, want_stats=fused)
        y4 = y.view(N, H, W, cout).permute(0, 3, 1, 2)
        rm = torch.zeros(cout, device=DEV); rv = torch.ones(cout, device=DEV)
        assert (ops._BN_WS_OWNER[0] == y.data_ptr()) == fused
        z = ops.BNActFn.apply(y4, None, gam, bet, rm, rv, True, 0.1, 1e-5, True)
        assert ops._BN_WS_OWNER[0] == 0
        outs.append((z.float(), rm.clone(), rv.clone()))
    assert (outs[0][0] - outs[1][0]).abs().max().item() <= 2e-2 * outs[1][0].abs().max().item()
    assert torch.allclose(outs[0][1], outs[1][1], rtol=1e-4, atol=1e-6) and torch.allclose(outs[0][2], outs[1][2], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('N,H,W,Cc', [(2, 8, 8, 64), (3, 16, 16, 64), (1, 56, 56, 64), (4, 12, 20, 64), (4, 12, 16, 64),
                                      (2, 8, 8, 128), (3, 28, 28, 128), (5, 7, 13, 128), (9, 4, 8, 128)])
def test_conv3x3_mfma_vs_torch(N, H, W, Cc):
    """lec_conv3x3_c64_fwd (3x3 / stride 1 / pad 1, 64 -> 64, NHWC bf16): forward, statistics partials and the data gradient
    (same kernel on the flipped, transposed weights) against torch in fp32; strips cross image rows and images."""
    g = torch.Generator(device='cpu').manual_seed(N * H)
    x = (torch.randn(N, Cc, H, W, generator=g) * 0.7).to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cc, Cc, 3, 3, generator=g) * 0.05).to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y = ops.conv3x3_c64(x, w, want_stats=True)
    n = ops._BN_WS_OWNER[1]; assert ops._BN_WS_OWNER[0] == y.data_ptr(); ops._BN_WS_OWNER[0] = 0
    ref = torch.nn.functional.conv2d(x.float(), w.float(), padding=1)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert (y.float() - ref).abs().max().item() <= 6e-3 * ref.abs().max().item()
    part = ops._bn_workspace(x.device)[:n * 2 * Cc * 4].view(torch.float32).view(n, 2, Cc).double().sum(0)
    yd = y.float().double()
    assert ((part[0] - yd.sum((0, 2, 3))).abs().max() / (yd.abs().sum((0, 2, 3)).max())).item() < 1e-5
    assert ((part[1] - (yd * yd).sum((0, 2, 3))).abs().max() / (yd * yd).sum((0, 2, 3)).max()).item() < 1e-5
    dy = torch.randn(N, Cc, H, W, generator=g).to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gx = ops.conv3x3_c64(dy, w, w_transposed=True)
    xr = x.float().requires_grad_(True)
    torch.nn.functional.conv2d(xr, w.float(), padding=1).backward(dy.float())
    assert (gx.float() - xr.grad).abs().max().item() <= 6e-3 * xr.grad.abs().max().item()


@pytest.mark.parametrize('B,K,D,N', [(64, 5, 10, 2000), (256, 256, 10, 50000), (32, 16, 128, 5000)])
def test_joint_loss_fp16_label_table_with_fp32_master(B, K, D, N):
    """BASELINE.json config 5 ("fp16+fp32-master", SURVEY.md 8(d) s = 2): the fused loss reads the label rows from a 2-byte fp16
    shadow of the table.  (a) Against the oracle evaluated ON THE fp16-ROUNDED TABLE the kernel is as exact as the fp32 path
    (energies <= 1e-4 abs): the arithmetic is unchanged, only the stored rows are rounded.  (b) Against the oracle on the fp32
    master the drift is the fp16 rounding of the rows: tolerance 5e-3 abs on the energies (rows have norm ~0.1: half-ulp 3e-5
    per element, amplified by the cone angle's conditioning near the apex)."""
    rs = np.random.RandomState(B + K + D)
    W = rs.randn(N, D).astype(np.float32); W = W / np.linalg.norm(W, axis=1, keepdims=True) * (0.1 + 0.3 * rs.rand(N, 1)).astype(np.float32)
    M = 64
    R = (rs.randn(M, D) * 0.3).astype(np.float32)
    frm = rs.randint(0, N, B); to = N + rs.randint(0, M, B)
    neg = np.concatenate([N + rs.randint(0, M, (B, K)), rs.randint(0, N, (B, K))], axis=1)
    dev = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(DEV) if dt is None else torch.from_numpy(np.ascontiguousarray(a).astype(dt)).to(DEV)
    code = lambda a: dev(np.where(a >= N, -1 - (a - N), a), np.int32)
    Wd = dev(W); Wh = Wd.to(torch.float16)
    gt = torch.zeros_like(Wd); gf = torch.zeros(M, D, device=DEV)
    loss, e_pos, e_neg = ops.joint_loss_raw(Wd, dev(R), code(frm), code(to), code(neg), None, 0.1, 0.01, _lib.ENERGY_HYP_CONE, _lib.LABEL_HYP,
                                            _lib.IMAGE_SOFTCLIP, gt, gf, table_f16=Wh)
    W16 = Wh.float().cpu().numpy()
    o16 = O.joint_loss_fwd_bwd(W16, R, frm, to, neg, 0.01, 0.1)
    o32 = O.joint_loss_fwd_bwd(W, R, frm, to, neg, 0.01, 0.1)
    assert np.abs(e_pos.cpu().numpy() - o16[1]).max() <= 1e-4 and np.abs(e_neg.cpu().numpy() - o16[2]).max() <= 1e-4
    assert abs(loss.item() - o16[0]) <= 1e-4 * max(1.0, abs(o16[0]))
    gs = np.abs(o16[3]).max() + 1e-12
    assert np.abs(gt.cpu().numpy() - o16[3]).max() <= 5e-3 * gs
    assert np.abs(e_pos.cpu().numpy() - o32[1]).max() <= 5e-3 and np.abs(e_neg.cpu().numpy() - o32[2]).max() <= 5e-3


def test_table_step_refreshes_the_fp16_shadow():
    """lec_table_step_adam_f16: the fp32 master, the Adam moments are bit-identical to the plain table step, and the shadow
    is the updated, clipped master rounded to fp16."""
    rs = np.random.RandomState(3)
    N, D = 5000, 10
    W = rs.randn(N, D).astype(np.float32); W = W / np.linalg.norm(W, axis=1, keepdims=True) * (0.05 + 0.9 * rs.rand(N, 1)).astype(np.float32)
    G = rs.randn(N, D).astype(np.float32)
    a = [torch.from_numpy(W.copy()).to(DEV), torch.from_numpy(G).to(DEV), torch.zeros(N, D, device=DEV), torch.zeros(N, D, device=DEV)]
    b = [t.clone() for t in a]
    sh = torch.zeros(N, D, dtype=torch.float16, device=DEV)
    for step in (1, 2):
        ops.table_step_adam(a[0], a[1], a[2], a[3], step, 1e-2, 0.1)
        ops.table_step_adam(b[0], b[1], b[2], b[3], step, 1e-2, 0.1, table_f16=sh)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert torch.equal(sh, b[0].to(torch.float16))


@pytest.mark.gpu
def test_joint_loss_skips_node_codes_outside_the_tables():
    """A stale or corrupt node code (label row >= n_labels, image row >= n_feat) reaching lec_joint_loss_fwd_bwd through the C ABI must not
    become an out-of-bounds read or atomic: the tables sit inside larger buffers whose guard rows hold a sentinel, and a launch with bad
    codes leaves every guard row -- values and gradients -- untouched and the energies of the pairs it did not corrupt are those of the
    clean launch.  (The bad code is treated as a zero row: its own energy -- and so the loss -- may come out NaN, which is the loud
    outcome a corrupt batch should have; the reference raises IndexError in nn.Embedding for the same input.)"""
    torch.manual_seed(0)
    N, M, D, B, K = 40, 12, 10, 8, 3
    bigW = torch.full((N + 16, D), 7.0, device=DEV); bigG = torch.full((N + 16, D), 7.0, device=DEV)
    bigF = torch.full((M + 16, D), 7.0, device=DEV); bigGF = torch.full((M + 16, D), 7.0, device=DEV)
    W = bigW[:N]; F_ = bigF[:M]
    W.copy_(torch.randn(N, D, device=DEV) * 0.1); F_.copy_(torch.randn(M, D, device=DEV))
    frm = torch.randint(0, N, (B,), device=DEV, dtype=torch.int32)
    to = (-1 - torch.randint(0, M, (B,), device=DEV)).to(torch.int32)
    neg = torch.randint(0, N, (B, 2 * K), device=DEV, dtype=torch.int32)
    def run(neg_):
        gW = bigG[:N]; gF = bigGF[:M]; gW.zero_(); gF.zero_()
        out = ops.joint_loss_raw(W, F_, frm, to, neg_.contiguous(), None, 0.1, 0.01, 0, 1, 1, grad_table=gW, grad_feat=gF)
        torch.cuda.synchronize()
        return [o.clone() for o in out], gW.clone(), gF.clone()
    (loss0, ep0, en0), gW0, gF0 = run(neg)
    bad = neg.clone()
    bad[1, 0] = N + 5                                   # label row past the table
    bad[2, 4] = -1 - (M + 3)                            # image row past the features
    bad[5, 2] = 2 ** 30                                 # far out
    (loss1, ep1, en1), gW1, gF1 = run(bad)
    for big in (bigW, bigG, bigF, bigGF):
        n = N if big.shape[0] == N + 16 else M
        assert (big[n:] == 7.0).all(), 'a guard row was written'
    assert torch.equal(ep0, ep1)                         # positives untouched
    clean = [b for b in range(B) if b not in (1, 2, 5)]
    assert torch.equal(en0[clean], en1[clean])


@pytest.mark.gpu
def test_order_embeddings_embedder_soft_clip_runs_the_kernel_and_matches_the_oracle():
    """order_embeddings.Embedder with K (order_embeddings.py:179-199; off in config 1): forward = direction * (norm + K) through
    lec_image_softclip_fwd, its gradient through lec_image_softclip_bwd, against the oracle's restatement and autograd of the formula."""
    from learning_embeddings_amd import order_embeddings as oe1
    from learning_embeddings_amd.hierarchy import SyntheticLabelMap
    torch.manual_seed(0)
    lm = SyntheticLabelMap([2, 4, 8])
    emb = oe1.Embedder(10, lm, K=3.0).to(DEV)
    idx = torch.randint(0, lm.n_classes, (4, 7), device=DEV)
    out = emb(idx)
    W = emb.embeddings.weight.detach()
    ref = O.soft_clip_add(W.cpu().numpy()[idx.cpu().numpy().reshape(-1)], 3.0).reshape(4, 7, 10)
    assert np.abs(out.detach().cpu().numpy() - ref).max() < 5e-6
    g = torch.randn_like(out)
    out.backward(g)
    x = W[idx].clone().requires_grad_(True)
    y = torch.nn.functional.normalize(x, dim=-1) * (x.norm(dim=-1, keepdim=True) + 3.0)
    y.backward(g)
    gW = torch.zeros_like(W).index_add_(0, idx.reshape(-1), x.grad.reshape(-1, 10))
    assert (emb.embeddings.weight.grad - gW).abs().max().item() < 5e-6


@pytest.mark.parametrize('B,K,D,M,chunk', [(32, 5, 10, 96, 32), (16, 40, 10, 300, 64), (24, 9, 128, 50, 16), (8, 256, 10, 240, 100)])
def test_joint_loss_row_windows_add_up_to_the_whole_batch_launch(B, K, D, M, chunk):
    """lec_joint_loss_fwd_bwd_window (the chunked step of config 5): one launch per window of feature rows evaluates the pairs whose image row lies in the
    window (label-label pairs ride with the first).  Over the windows: every energy equals the whole-batch launch's BIT FOR BIT, the loss values add up to
    its loss, the gradients to its gradients; and rows OUTSIDE a launch's window are never read into a result -- they hold NaN while that launch runs."""
    rs = np.random.RandomState(B + K + D)
    N = 400
    W = rs.randn(N, D).astype(np.float32)
    W *= (rs.uniform(0.1, 0.6, (N, 1)) / np.linalg.norm(W, axis=1, keepdims=True)).astype(np.float32)
    R = (rs.randn(M, D) * 0.3).astype(np.float32)
    # at most one image end point per pair (what a pick_per_level trainer draws): (label, image) positives; u-fixed negatives corrupt `to` with a label or an
    # image, v-fixed negatives corrupt `from` with a label
    frm = rs.randint(0, N, B); to = N + rs.permutation(M)[:B]
    neg = np.empty((B, 2 * K), dtype=np.int64)
    neg[:, :K] = np.where(rs.rand(B, K) < 0.5, rs.randint(0, N, (B, K)), N + rs.randint(0, M, (B, K)))
    neg[:, K:] = rs.randint(0, N, (B, K))
    to[:B // 4] = rs.randint(0, N, B // 4)                                     # a few label-label positives: their u-fixed negatives may still be images
    for b in range(B):
        for k in range(2 * K):
            other = frm[b] if k < K else to[b]
            while neg[b, k] == other:
                neg[b, k] = rs.randint(0, N)
        while frm[b] == to[b]:
            frm[b] = rs.randint(0, N)
    code = lambda a: torch.tensor(np.where(a < N, a, -1 - (a - N)), dtype=torch.int32, device=DEV).contiguous()
    Wt, Rt = T(W), T(R)
    cf, ct, cn = code(frm), code(to), code(neg)
    alpha = 1.5
    gW0 = torch.zeros_like(Wt); gR0 = torch.zeros_like(Rt)
    l0, p0, n0 = ops.joint_loss_raw(Wt, Rt, cf, ct, cn, None, 0.1, alpha, 0, 1, 1, gW0, gR0)
    gW = torch.zeros_like(Wt); gR = torch.zeros_like(Rt)
    out = (torch.full((B,), float('nan'), device=DEV), torch.full((B, 2 * K), float('nan'), device=DEV))
    total = 0.0
    for lo in range(0, M, chunk):
        hi = min(lo + chunk, M)
        Rw = torch.full_like(Rt, float('nan')); Rw[lo:hi] = Rt[lo:hi]        # only this window's rows exist
        l_c, _, _ = ops.joint_loss_raw(Wt, Rw, cf, ct, cn, None, 0.1, alpha, 0, 1, 1, gW, gR, window=(lo, hi, lo == 0), out=out)
        assert torch.isfinite(l_c).all()
        total += float(l_c.item())
    assert torch.equal(out[0], p0) and torch.equal(out[1], n0)
    assert abs(total - float(l0.item())) <= 1e-5 * max(1.0, abs(float(l0.item())))
    assert torch.isfinite(gW).all() and torch.isfinite(gR).all()
    assert (gW - gW0).abs().max().item() <= 1e-4 * gW0.abs().max().item() + 1e-7
    assert (gR - gR0).abs().max().item() <= 1e-4 * gR0.abs().max().item() + 1e-7
    # and with the fp16 shadow of the table
    W16 = Wt.to(torch.float16)
    l1, p1, n1 = ops.joint_loss_raw(Wt, Rt, cf, ct, cn, None, 0.1, alpha, 0, 1, 1, table_f16=W16)
    out2 = (torch.zeros(B, device=DEV), torch.zeros(B, 2 * K, device=DEV))
    for lo in range(0, M, chunk):
        ops.joint_loss_raw(Wt, Rt, cf, ct, cn, None, 0.1, alpha, 0, 1, 1, table_f16=W16, window=(lo, min(lo + chunk, M), lo == 0), out=out2)
    assert torch.equal(out2[0], p1) and torch.equal(out2[1], n1)


def test_joint_loss_window_rejects_image_image_pairs_that_straddle_two_windows():
    """The windowed entry's contract (include/lecone.h): a pair of TWO image rows in DIFFERENT windows cannot be evaluated by any single launch (the
    other row is not there) -- the owning launch must answer NaN (that pair's energy and its loss), never a finite energy computed from a zero row; the same pair
    with both rows inside ONE window is evaluated normally, bit for bit like the whole-batch launch."""
    rs = np.random.RandomState(3)
    N, M, D, B, K = 50, 64, 10, 8, 3
    W = rs.randn(N, D).astype(np.float32); W *= (rs.uniform(0.1, 0.6, (N, 1)) / np.linalg.norm(W, axis=1, keepdims=True)).astype(np.float32)
    R = (rs.randn(M, D) * 0.3).astype(np.float32)
    Wt, Rt = T(W), T(R)
    frm = torch.tensor(rs.randint(0, N, B), dtype=torch.int32, device=DEV)
    to = torch.tensor(-1 - np.arange(B), dtype=torch.int32, device=DEV)                   # image rows 0 .. B-1 (first window)
    neg = torch.tensor(rs.randint(0, N, (B, 2 * K)), dtype=torch.int32, device=DEV)
    neg[:, :K] += (neg[:, :K] == frm[:, None]).int()
    neg[2, K + 1] = -1 - 40                                                               # v-fixed negative of positive 2: (image row 40, image row 2)
    l0, p0, n0 = ops.joint_loss_raw(Wt, Rt, frm, to, neg, None, 0.1, 1.5, 0, 1, 1)
    assert torch.isfinite(l0).all() and torch.isfinite(n0).all()
    # one window over everything: fine, identical
    out = (torch.zeros(B, device=DEV), torch.zeros(B, 2 * K, device=DEV))
    l1, _, _ = ops.joint_loss_raw(Wt, Rt, frm, to, neg, None, 0.1, 1.5, 0, 1, 1, window=(0, M, True), out=out)
    assert torch.equal(out[1], n0) and abs(float(l1.item()) - float(l0.item())) <= 1e-5 * abs(float(l0.item()))
    # two windows [0, 32) | [32, 64): rows 2 and 40 never meet
    out = (torch.zeros(B, device=DEV), torch.zeros(B, 2 * K, device=DEV))
    la, _, _ = ops.joint_loss_raw(Wt, Rt, frm, to, neg, None, 0.1, 1.5, 0, 1, 1, window=(0, 32, True), out=out)
    lb, _, _ = ops.joint_loss_raw(Wt, Rt, frm, to, neg, None, 0.1, 1.5, 0, 1, 1, window=(32, 64, False), out=out)
    assert torch.isfinite(la).all(), 'the first window owns none of the offending pair'
    assert torch.isnan(lb).all() and torch.isnan(out[1][2, K + 1]), 'the launch owning the larger row must poison the pair, not invent an energy'
    ok = torch.ones_like(out[1], dtype=torch.bool); ok[2, K + 1] = False
    assert torch.equal(out[1][ok], n0[ok])


def test_joint_loss_fixed_point_hand_off_equals_the_ticket_form_and_carries_nan():
    """Round 5: where the loss is bounded at launch (no per-positive weights, cone energies) every block hands its partial over as ONE 64-bit integer atomic
    (count in the top bits, llrint(partial * 2^F) below: integer addition does not depend on the order of arrival).  Against the ticket form (selected by passing
    unit weights): the same loss to float rounding, run-to-run identical bits; a NaN energy (a NaN image embedding: torch.sum keeps NaN, oe_h.py:843-846) still makes the
    loss NaN, and the launch after it is clean again (the accumulator and the flag are re-armed)."""
    rs = np.random.RandomState(11)
    N, M, D = 300, 64, 10
    W = rs.randn(N, D).astype(np.float32); W *= (rs.uniform(0.1, 0.6, (N, 1)) / np.linalg.norm(W, axis=1, keepdims=True)).astype(np.float32)
    R = (rs.randn(M, D) * 0.3).astype(np.float32)
    Wt, Rt = T(W), T(R)
    for B, K in ((64, 5), (256, 64), (32, 256)):
        frm = torch.tensor(rs.randint(0, N, B), dtype=torch.int32, device=DEV)
        to = torch.tensor(-1 - rs.randint(0, M, B), dtype=torch.int32, device=DEV)
        neg = torch.tensor(rs.randint(0, N, (B, 2 * K)), dtype=torch.int32, device=DEV)
        neg[:, :K] += (neg[:, :K] == frm[:, None]).int()                          # never (u, u): x == y is 0 / 0 in the reference too
        neg[:, :K] -= 2 * (neg[:, :K] >= N).int()
        neg = neg.contiguous()
        ones = torch.ones(B, device=DEV)
        fx = [ops.joint_loss_raw(Wt, Rt, frm, to, neg, None, 0.1, 0.5, 0, 1, 1)[0].item() for _ in range(4)]
        tk = ops.joint_loss_raw(Wt, Rt, frm, to, neg, ones, 0.1, 0.5, 0, 1, 1)[0].item()
        assert len(set(fx)) == 1 and np.isfinite(fx[0])
        assert abs(fx[0] - tk) <= 2e-6 * abs(tk), (B, K, fx[0], tk)
        Rz = Rt.clone(); Rz[int(-1 - to[0].item())] = float('nan')                # a NaN image embedding (a diverged CNN): its pairs' energies are NaN
        assert np.isnan(ops.joint_loss_raw(Wt, Rz, frm, to, neg, None, 0.1, 0.5, 0, 1, 1)[0].item())
        assert np.isnan(ops.joint_loss_raw(Wt, Rz, frm, to, neg, ones, 0.1, 0.5, 0, 1, 1)[0].item())
        assert ops.joint_loss_raw(Wt, Rt, frm, to, neg, None, 0.1, 0.5, 0, 1, 1)[0].item() == fx[0]
