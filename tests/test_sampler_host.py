"""Host-side tests (no GPU): the C++ sampler behind the C ABI, bit-exact against the reference fixtures and the oracle."""
import json, os, random
import numpy as np
import pytest
from conftest import GOLDEN
from oracle import cone_oracle as O
from learning_embeddings_amd import _lib
from learning_embeddings_amd.hierarchy import NegativeGraph, SyntheticLabelMap, image_parents_by_leaf, SYNTHETIC


def test_mt19937_kats():
    f = json.load(open(os.path.join(GOLDEN, 'F4_sampler.json')))
    g = NegativeGraph([2, 8], SyntheticLabelMap([2, 8]).edges, seed=0)
    assert [g.next_u32() for _ in range(16)] == f['mt19937_seed0_u32']
    for seed in (1, 12345, 2**40 + 17):
        g.seed(seed); random.seed(seed)
        assert [g.next_u32() for _ in range(1300)] == [random.getrandbits(32) for _ in range(1300)]


def test_F4_reference_stream():
    f = json.load(open(os.path.join(GOLDEN, 'F4_sampler.json')))
    for case in f['cases']:
        lm = SyntheticLabelMap(case['levels'], edges=[tuple(e) for e in case['edges']])
        g = NegativeGraph.from_labelmap(lm, n_images=case['n_images'], pick_per_level=case['pick_per_level'], seed=0)
        if case['levels_to_hide']:
            g.set_levels_to_hide(case['levels_to_hide'])
        got = [g.draw(side, ix, lvl) for side, ix, lvl in case['calls']]
        assert got == case['out'], (case['hierarchy'], case['pick_per_level'], case['levels_to_hide'])


def _s5_graph(ppl=True):
    f = json.load(open(os.path.join(GOLDEN, 'F4b_sampler_s5.json')))
    z = np.load(os.path.join(GOLDEN, 'F4b_sampler_s5_step0.npz'))
    lm = SyntheticLabelMap(f['levels'])
    g = NegativeGraph.from_labelmap(lm, image_leaf=lm.level_start[-1] + z['image_leaf'].astype(np.int64), pick_per_level=ppl, seed=0)
    return f, z, lm, g


def test_F4b_reference_stream_config5_hierarchy():
    """Config 5's DAG (8 levels, 50 000 labels, 4 096 images spread over the leaves as engine.StepEngine spreads them): the reference's own
    sample_negative_edge over the real 54 096^2 dense matrix (tests/golden/make_golden_sampler_s5.py), every level slot 0..8 on both sides,
    label and image end points, pick_per_level on / off, hidden-level remaps incl. the ones CPython's set iterates out of ascending order."""
    f, z, lm, _ = _s5_graph()
    assert lm.levels == SYNTHETIC['S5']
    for case in f['cases']:
        g = _s5_graph(case['pick_per_level'])[3]
        g.set_levels_to_hide(case['levels_to_hide'])
        got = [g.draw(side, ix, lvl) for side, ix, lvl in case['calls']]
        assert got == case['out'], (case['pick_per_level'], case['levels_to_hide'])


def test_F4b_reference_stream_config5_first_batch():
    """The whole first batch of config 5 (B = 256 positives x 2K = 512 draws, random.seed(0)) as the reference's loop oe_h.py:940-957 drew it,
    and the state of the MT19937 stream after it."""
    f, z, lm, g = _s5_graph()
    neg = g.draw_batch(z['pos_from'], z['pos_to'], int(z['K']))
    assert neg.shape == (256, 512) and np.array_equal(neg, z['neg'])
    assert [g.next_u32() for _ in range(4)] == z['stream_after'].tolist()


def test_hidden_level_slot_order_is_cpythons_set_order():
    """oe_h.py:854 indexes list(set(range(L+1)) - set(hidden)): the sampler reproduces CPython's iteration order of that set (checked against
    this interpreter, which is the one the reference's fixtures were made with), for every hide set of 8 levels and samples of deeper ones."""
    import itertools
    assert list(set(list(range(9))) - {0, 4, 5, 6, 7}) == [8, 1, 2, 3]              # the case that is not ascending (F4b holds its stream)
    for L in (2, 4, 8, 9, 16, 40):
        g = NegativeGraph.from_labelmap(SyntheticLabelMap([2] * L))
        assert g.visible_slots() == list(range(L + 1))
        if L <= 9:
            sets = [h for r in range(1, L + 1) for h in itertools.combinations(range(L + 1), r)]
        else:
            rs = random.Random(L)
            sets = [tuple(rs.sample(range(L + 1), rs.randint(1, L))) for _ in range(1500)]
        for hide in sets:
            g.set_levels_to_hide(list(hide))
            assert g.visible_slots() == list(set(list(range(L + 1))) - set(hide)), (L, hide)


@pytest.mark.parametrize('tag', ['s3', 'ethec'])
def test_F5_batch_negatives(tag):
    f = np.load(os.path.join(GOLDEN, 'F5_criterion.npz'))
    k = lambda n: f[tag + '_' + n]
    lm = SyntheticLabelMap(k('levels').tolist(), edges=[tuple(e) for e in k('edges').tolist()])
    g = NegativeGraph.from_labelmap(lm, n_images=int(k('n_images')), pick_per_level=bool(k('pick_per_level')), seed=0)
    neg = g.draw_batch(k('from'), k('to'), int(k('Kneg')))
    assert np.array_equal(neg, k('neg'))


def test_from_dense_matches_csr_and_oracle():
    lm = SyntheticLabelMap([3, 7, 20])
    M = 37
    leaf = [lm.level_start[-1] + (j % lm.levels[-1]) for j in range(M)]
    A = O.dense_negative_adjacency(lm.n_classes, sorted(lm.edges), leaf)
    for ppl in (False, True):
        g1 = NegativeGraph.from_dense(A, lm.levels, pick_per_level=ppl, seed=3)
        g2 = NegativeGraph.from_labelmap(lm, n_images=M, pick_per_level=ppl, seed=3)
        s = O.DenseSampler(A, lm.levels, pick_per_level=ppl, seed=3)
        rs = np.random.RandomState(0)
        for _ in range(500):
            side, node, lvl = int(rs.randint(2)), int(rs.randint(lm.n_classes + M)), int(rs.randint(9))
            try:
                want = s.draw(side, node, lvl)
            except IndexError:
                with pytest.raises(IndexError):
                    g1.draw(side, node, lvl)
                with pytest.raises(IndexError):
                    g2.draw(side, node, lvl)
                continue
            assert g1.draw(side, node, lvl) == want and g2.draw(side, node, lvl) == want
    assert g2.tc_edges == int((~A).sum()) - A.shape[0]


def test_labels_only_mode_and_general_dag():
    # a DAG that is not level-adjacent and has two parents for one node
    levels = [2, 3, 4]
    edges = [(0, 2), (1, 3), (1, 4), (2, 5), (2, 6), (3, 6), (4, 7), (0, 8), (3, 8)]
    A = O.dense_negative_adjacency(9, edges)
    s = O.DenseSampler(A, levels, pick_per_level=True, seed=0, labels_only=True)
    g = NegativeGraph(levels, edges, pick_per_level=True, labels_only=True, seed=0)
    rs = np.random.RandomState(1)
    for _ in range(400):
        side, node, lvl = int(rs.randint(2)), int(rs.randint(9)), int(rs.randint(7))
        try:
            want = s.draw(side, node, lvl)
        except IndexError:
            with pytest.raises(IndexError):
                g.draw(side, node, lvl)
            continue
        assert g.draw(side, node, lvl) == want


def test_big_hierarchy_no_dense_matrix():
    """S5 (50 000 labels) + 100 000 images: the dense matrix would be 22 GB; the CSR sampler builds in milliseconds.
    Property checks: a drawn negative is never the node itself nor a TC neighbour, and lies in the level window."""
    lm = SyntheticLabelMap(SYNTHETIC['S5'])
    M = 100000
    g = NegativeGraph.from_labelmap(lm, n_images=M, pick_per_level=True, seed=0)
    par = lm.parents()
    def anc(v):
        out = set()
        while v in par:
            v = par[v][0]; out.add(v)
        return out
    N = lm.n_classes; L = len(lm.levels)
    rs = np.random.RandomState(5)
    for _ in range(2000):
        j = int(rs.randint(M)); img = N + j
        leaf = lm.level_start[-1] + j % lm.levels[-1]
        a = anc(leaf) | {leaf}
        lvl = int(rs.randint(L))
        r = g.draw(1, img, lvl)                          # corrupt the label end of (label, image)
        assert lm.level_start[lvl] <= r < lm.level_stop[lvl] and r not in a
        u = sorted(a)[lvl]
        r2 = g.draw(0, u, lvl)                           # corrupt the `to` end of (u, image): a label of level lvl
        assert lm.level_start[lvl] <= r2 < lm.level_stop[lvl] and r2 != u and u not in anc(r2)
        r3 = g.draw(0, u, L)                             # slot L, u is a label -> images only
        assert r3 >= N and u not in (anc(lm.level_start[-1] + (r3 - N) % lm.levels[-1]) | {lm.level_start[-1] + (r3 - N) % lm.levels[-1]})


def test_errors_are_loud():
    g = NegativeGraph([1, 1], [(0, 1)], pick_per_level=True, seed=0)
    with pytest.raises(IndexError):
        g.draw(0, 0, 0)                                  # level 0 holds only the node itself -> empty
    with pytest.raises(_lib.LeconeError):
        g.draw(0, 99, 0)
    with pytest.raises(_lib.LeconeError):
        NegativeGraph([2, 2], [(0, 7)])
    with pytest.raises(_lib.LeconeError):
        NegativeGraph([2, 2], [(0, 2), (2, 0)])          # cycle
