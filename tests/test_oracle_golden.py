"""The oracle (oracle/cone_oracle.py) pinned against the fixtures generated from the reference (tests/golden/make_golden.py)."""
import json, os, random
import numpy as np
import pytest
from oracle import cone_oracle as O
from conftest import GOLDEN


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b) / (1e-6 + np.maximum(np.abs(a), np.abs(b)))


@pytest.mark.parametrize('D', [2, 10, 128])
def test_F1_cone_energy(D):
    f = load('F1_cone_energy.npz'); K = float(f['K'])
    x, y = f['x_%d' % D], f['y_%d' % D]
    E = O.cone_energy(x, y, K)
    # north-star tolerance on the cone energy: 1e-4 abs.  Rows where the reference's own float32 result is further than
    # that from the reference evaluated in float64 (near-coincident points: catastrophic cancellation in oe_h.py:823)
    # get 4x the reference's own float32 noise instead.
    tol = np.maximum(1e-4, 4 * np.abs(f['E_%d' % D] - f['E64_%d' % D]))
    assert (np.abs(E - f['E_%d' % D]) <= tol).all()
    assert (np.abs(O.cone_energy(x, y, K, np.float64) - f['E64_%d' % D]) <= 1e-9).all()
    gx, gy = O.cone_energy_grad(x, y, f['gE_%d' % D], K)
    # reference autograd is fp32 and d(acos)/da reaches ~224 at the clamp: compare relative to the row's gradient scale,
    # again floored by the reference's own fp32-vs-fp64 deviation on that row
    for g, r, r64 in ((gx, f['gx_%d' % D], f['gx64_%d' % D]), (gy, f['gy_%d' % D], f['gy64_%d' % D])):
        scale = np.abs(r).max(axis=1, keepdims=True) + 1e-3
        noise = np.abs(r - r64).max(axis=1, keepdims=True) / scale
        assert (np.abs(g - r) / scale <= np.maximum(1e-3, 4 * noise)).all()
        live = np.abs(r64).max(axis=1) > 0                                 # fp64 analytic vs fp64 autograd where masks agree
        same = (np.abs(f['E_%d' % D] - f['E64_%d' % D]) < 1e-4) & live
        assert (np.abs(g - r64)[same] / scale[same]).max() < 1e-3


def test_F2_embedder():
    f = load('F2_embedder.npz'); K = float(f['K'])
    assert abs(O.inner_radius(K) - float(f['inner_radius'])) < 1e-15
    assert abs(float(O.inner_radius_h(K)) - float(f['inner_radius_h'])) < 1e-7
    out = O.embedder_forward(f['W'], f['idx'], K)
    assert np.abs(out - f['out']).max() < 2e-6
    gW = O.embedder_backward(f['W'], f['idx'], f['gout'], K)
    assert np.abs(gW - f['gW']).max() / np.abs(f['gW']).max() < 1e-5
    n0 = np.linalg.norm(f['W_init'], axis=1)                              # a1: init rows at r_in + U[0,0.05)
    assert n0.min() >= O.inner_radius(K) - 1e-6 and n0.max() < O.inner_radius(K) + 0.05 + 1e-6


def test_F3_image_projection():
    f = load('F3_image_proj.npz'); K = float(f['K'])
    assert relerr(O.image_soft_clip(f['raw'], K), f['soft_clip']).max() < 1e-5
    g = O.image_soft_clip_backward(f['raw'], f['gout'], K)
    assert (np.abs(g - f['graw']) / (np.abs(f['graw']).max(axis=1, keepdims=True) + 1e-6)).max() < 1e-4
    assert np.abs(O.featnet_forward(f['fn_in'], f['fn_w'], f['fn_b'], K) - f['fn_out']).max() < 2e-6


def test_F4_mt19937_kats():
    f = json.load(open(os.path.join(GOLDEN, 'F4_sampler.json')))
    r = O.MT19937(0)
    assert [r.u32() for _ in range(16)] == f['mt19937_seed0_u32']
    assert f['mt19937_seed0_u32'][:4] == [3626764237, 1654615998, 3255389356, 3823568514]   # SURVEY.md section 7 KAT
    r = O.MT19937(0)
    assert [r.randbelow(2000) for _ in range(16)] == f['choice_range2000']
    assert f['choice_range2000'][:6] == [1729, 788, 1552, 1823, 861, 82]
    r = O.MT19937(12345)
    assert [[n, r.randbelow(n)] for n, _ in f['randbelow_mixed']] == f['randbelow_mixed']
    # and against this interpreter's own `random` (the third-party algorithm itself)
    for seed in (0, 1, 2**40 + 17):
        random.seed(seed); r = O.MT19937(seed)
        assert [random.getrandbits(32) for _ in range(700)] == [r.u32() for _ in range(700)]


def test_F4_dense_sampler():
    f = json.load(open(os.path.join(GOLDEN, 'F4_sampler.json')))
    for case in f['cases']:
        levels = case['levels']; N = sum(levels); M = case['n_images']
        leaf = [N - levels[-1] + (j % levels[-1]) for j in range(M)]
        A = O.dense_negative_adjacency(N, case['edges'], leaf)
        s = O.DenseSampler(A, levels, pick_per_level=case['pick_per_level'], seed=0)
        s.levels_to_hide = case['levels_to_hide']
        got = [s.draw(side, ix, lvl) for side, ix, lvl in case['calls']]
        assert got == case['out'], (case['hierarchy'], case['pick_per_level'], case['levels_to_hide'])


def test_F4b_lazy_sampler_config5_hierarchy():
    """The oracle's matrix-free mode (a row / column of A written out per draw) against the reference's run over the REAL dense 54 096^2 matrix of
    config 5's hierarchy: every scripted case (level slots 0..8, both sides, labels and images, hidden-level remaps in CPython's set order) and the
    first 24 positives x 512 draws of the engine's first batch.  Also: lazy == dense on a hierarchy small enough to hold both."""
    f = json.load(open(os.path.join(GOLDEN, 'F4b_sampler_s5.json')))
    z = np.load(os.path.join(GOLDEN, 'F4b_sampler_s5_step0.npz'))
    levels = f['levels']; N = sum(levels)
    edges = [(sum(levels[:l - 1]) + (c * levels[l - 1]) // levels[l], sum(levels[:l]) + c) for l in range(1, len(levels)) for c in range(levels[l])]
    leaf = (N - levels[-1]) + z['image_leaf'].astype(np.int64)
    assert np.array_equal(z['image_leaf'], (np.arange(f['n_images'], dtype=np.int64) * levels[-1]) // f['n_images'])
    s = O.LazyDenseSampler(levels, edges, leaf)
    for case in f['cases']:
        s.pick_per_level = case['pick_per_level']; s.levels_to_hide = case['levels_to_hide']; s.seed(0)
        got = [s.draw(side, ix, lvl) for side, ix, lvl in case['calls']]
        assert got == case['out'], (case['pick_per_level'], case['levels_to_hide'])
    s.pick_per_level = True; s.levels_to_hide = []; s.seed(0)
    nb = 24
    assert np.array_equal(s.draw_batch(z['pos_from'][:nb], z['pos_to'][:nb], int(z['K'])), z['neg'][:nb])
    # lazy rows / columns == the dense matrix's
    lv = [3, 7, 20]; n = sum(lv)
    ed = [(sum(lv[:l - 1]) + (c * lv[l - 1]) // lv[l], sum(lv[:l]) + c) for l in range(1, 3) for c in range(lv[l])]
    lf = [n - lv[-1] + (j * 7) % lv[-1] for j in range(31)]
    A = O.dense_negative_adjacency(n, ed, lf)
    lz = O.LazyDenseSampler(lv, ed, lf)
    for ix in range(n + 31):
        assert np.array_equal(lz._row(ix), A[ix, :]) and np.array_equal(lz._col(ix), A[:, ix])


@pytest.mark.parametrize('tag', ['s3', 'ethec'])
def test_F5_criterion(tag):
    f = load('F5_criterion.npz')
    g = lambda k: f[tag + '_' + k]
    levels = g('levels').tolist(); N = sum(levels); M = int(g('n_images')); Kn = int(g('Kneg'))
    leaf = [N - levels[-1] + (j % levels[-1]) for j in range(M)]
    A = O.dense_negative_adjacency(N, g('edges').tolist(), leaf)
    s = O.DenseSampler(A, levels, pick_per_level=bool(g('pick_per_level')), seed=0)
    neg = s.draw_batch(g('from'), g('to'), Kn)
    assert np.array_equal(neg, g('neg'))                                  # bit-exact negative selection
    loss, e_pos, e_neg, gW, gR = O.joint_loss_fwd_bwd(g('W'), g('R'), g('from'), g('to'), neg, float(g('alpha')), float(g('K')))
    assert np.abs(e_pos - g('e_pos')).max() <= 1e-4
    assert np.abs(e_neg - g('e_neg')[..., 0]).max() <= 1e-4
    assert abs(loss - float(g('loss'))) <= 1e-4 * max(1.0, abs(float(g('loss'))))
    assert np.abs(gW - g('gW')).max() / np.abs(g('gW')).max() < 1e-3
    assert np.abs(gR - g('gR')).max() / np.abs(g('gR')).max() < 1e-3


@pytest.mark.parametrize('D', [10, 2])
def test_F6_table_step(D):
    f = load('F6_table_step.npz'); K = float(f['K'])
    W = f['adam_W0_%d' % D]; m = np.zeros_like(W); v = np.zeros_like(W)
    for step in range(3):
        W, m, v = O.table_step_adam(W, f['adam_grads_%d' % D][step], m, v, step + 1, float(f['lr_adam']), K)
        assert np.abs(W - f['adam_W_%d' % D][step]).max() < 2e-6
    assert np.abs(m - f['adam_m_%d' % D]).max() < 1e-6 and np.abs(v - f['adam_v_%d' % D]).max() < 1e-6
    Wr = O.table_step_rsgd(f['rsgd_W0_%d' % D], f['rsgd_grad_%d' % D], float(f['lr_rsgd']), K)
    assert np.abs(Wr - f['rsgd_W_%d' % D]).max() < 2e-6


@pytest.mark.parametrize('tag', ['toy2', 'toy3'])
def test_F7_order_embedding(tag):
    f = load('F7_order_embedding.npz')
    assert np.abs(O.order_energy(f['x'], f['y']) - f['E']).max() < 1e-5
    gx, gy = O.order_energy_grad(f['x'], f['y'], f['gE'])
    assert np.abs(gx - f['gx']).max() < 1e-5 and np.abs(gy - f['gy']).max() < 1e-5
    g = lambda k: f[tag + '_' + k]
    levels = g('levels').tolist(); N = sum(levels)
    A = O.dense_negative_adjacency(N, g('edges').tolist())
    s = O.DenseSampler(A, levels, pick_per_level=True, seed=0, labels_only=True)
    neg = s.draw_batch(g('from'), g('to'), 4)
    assert np.array_equal(neg, g('neg'))
    loss, e_pos, e_neg, gW, _ = O.joint_loss_fwd_bwd(g('W'), None, g('from'), g('to'), neg, 1.0, None, energy='order')
    assert np.abs(e_pos - g('e_pos')).max() < 1e-5 and np.abs(e_neg.reshape(-1) - g('e_neg')).max() < 1e-5
    assert abs(loss - float(g('loss'))) < 1e-4 * abs(float(g('loss')))
    assert np.abs(gW - g('gW')).max() < 1e-4


@pytest.mark.parametrize('tag,w', [('unw', None), ('w', 'level_weights_w')])
def test_F8_multilevel_ce(tag, w):
    f = load('F8_multilevel_ce.npz')
    loss, g = O.multilevel_ce(f['logits'], f['level_labels'], f['levels'].tolist(), None if w is None else f[w])
    assert abs(loss - float(f[tag + '_loss'])) < 1e-5 * abs(float(f[tag + '_loss']))
    assert np.abs(g - f[tag + '_glogits']).max() < 1e-6


@pytest.mark.parametrize('tag,w', [('unw', None), ('w', 'level_weights_w')])
def test_F8b_multilevel_ce_with_class_weights(tag, w):
    f = load('F8b_multilevel_ce_class_weights.npz')
    loss, g = O.multilevel_ce(f['logits'], f['level_labels'], f['levels'].tolist(), None if w is None else f[w], class_weights=f['class_weights'])
    assert abs(loss - float(f[tag + '_loss'])) < 1e-5 * abs(float(f[tag + '_loss']))
    assert np.abs(g - f[tag + '_glogits']).max() < 1e-6


def test_F9_ethec_hierarchy():
    f = json.load(open(os.path.join(GOLDEN, 'F9_ethec_hierarchy.json')))
    assert f['levels'] == [6, 21, 135, 561] and len(f['edges']) == 717
    A = O.dense_negative_adjacency(sum(f['levels']), f['edges'])
    assert int((~A).sum()) - sum(f['levels']) == 1974                      # transitive-closure edge count


@pytest.mark.parametrize('D', [2, 10, 128])
def test_F10_euclidean_cone_energy(D):
    """network/oe.py:721-739 (the Euclidean-cone sibling)."""
    f = load('F10_euclidean_cone.npz'); K = float(f['K'])
    x, y = f['x%d' % D], f['y%d' % D]
    E = O.euc_cone_energy(x, y, K)
    assert (np.abs(E - f['E%d' % D]) <= np.maximum(1e-4, 4 * np.abs(f['E%d' % D] - f['E64_%d' % D]))).all()
    assert (np.abs(O.euc_cone_energy(x, y, K, np.float64) - f['E64_%d' % D]) <= 1e-9).all()
    gx, gy = O.euc_cone_energy_grad(x, y, f['gE%d' % D], K)
    for g, r, r64 in ((gx, f['gx%d' % D], f['gx64_%d' % D]), (gy, f['gy%d' % D], f['gy64_%d' % D])):
        scale = np.abs(r).max(axis=1, keepdims=True) + 1e-3
        noise = np.abs(r - r64).max(axis=1, keepdims=True) / scale            # psi ~ 0 rows: 1 - K^2/|x|^2 cancels in fp32
        assert (np.abs(g - r) / scale <= np.maximum(1e-3, 4 * noise)).all()
        assert (np.abs(g - r64) / (np.abs(r64).max(axis=1, keepdims=True) + 1e-3)).max() < 1e-6


def test_F10_euclidean_projections_and_criterion():
    f = load('F10_euclidean_cone.npz'); K = float(f['K'])
    rows = f['emb_W'][f['emb_idx']]
    assert np.abs(O.soft_clip_add(rows, K) - f['emb_out']).max() < 2e-6                      # oe.py:65-80
    gW = np.zeros(f['emb_W'].shape); np.add.at(gW, f['emb_idx'], O.soft_clip_add_backward(rows, f['emb_gout'], K))
    assert np.abs(gW - f['emb_gW']).max() / np.abs(f['emb_gW']).max() < 1e-5
    assert np.abs(O.soft_clip_add(f['img_raw'], K) - f['img_out']).max() < 2e-6              # oe.py:235-240
    assert np.abs(O.soft_clip_add_backward(f['img_raw'], f['img_gout'], K) - f['img_graw']).max() < 1e-5
    loss, e_pos, e_neg, gW, gR = O.joint_loss_fwd_bwd(f['c_W'], f['c_R'], f['c_from'], f['c_to'], f['c_neg'],
                                                      float(f['c_alpha']), K, energy='euc_cone')   # oe.py:810-873
    assert np.abs(e_pos - f['c_e_pos']).max() < 1e-5 and np.abs(e_neg - f['c_e_neg'][..., 0]).max() < 1e-5
    assert abs(loss - float(f['c_loss'])) < 1e-4 * abs(float(f['c_loss']))
    assert np.abs(gW - f['c_gW']).max() / np.abs(f['c_gW']).max() < 1e-5
    assert np.abs(gR - f['c_gR']).max() / np.abs(f['c_gR']).max() < 1e-5
    # the sampler stream is the one oe_h.py uses: the dense-matrix restatement reproduces the negatives
    A = O.dense_negative_adjacency(int(f['c_W'].shape[0]), f['c_edges'],
                                   image_leaf=[int(f['c_W'].shape[0]) - int(f['c_levels'][-1]) + (j % int(f['c_levels'][-1]))
                                               for j in range(int(f['c_n_images']))])
    smp = O.DenseSampler(A, f['c_levels'].tolist(), n_labels=int(f['c_W'].shape[0]), pick_per_level=True, seed=0)
    assert np.array_equal(smp.draw_batch(f['c_from'], f['c_to'], int(f['c_Kneg'])), f['c_neg'])
