"""Host logic that runs without a GPU: Embedder init parity (a1), labelmaps, graph construction, DP plumbing over gloo."""
import json, os, socket, sys
import numpy as np
import pytest
import torch
import torch.multiprocessing as mp
from conftest import GOLDEN, ROOT
from oracle import cone_oracle as O
from learning_embeddings_amd.hierarchy import SyntheticLabelMap, NegativeGraph, SYNTHETIC
from learning_embeddings_amd import oe_h, embed_toy, parallel
from learning_embeddings_amd.oe_h_trainer import DiGraph, transitive_closure, create_combined_graphs, EmbeddingMetrics, GlobalBatchSampler


def test_embedder_init_matches_reference_seed0():
    f = np.load(os.path.join(GOLDEN, 'F2_embedder.npz'))
    torch.manual_seed(0)
    emb = oe_h.Embedder(10, SyntheticLabelMap.ethec(), None, K=0.1)      # same RNG consumption order as oe_h.py:62-73
    assert np.array_equal(emb.embeddings.weight.detach().numpy(), f['W_init'])
    assert abs(emb.inner_radius - float(f['inner_radius'])) < 1e-15
    assert abs(float(emb.inner_radius_h) - float(f['inner_radius_h'])) < 1e-7


def test_toygraph_matches_reference_fixture():
    f = np.load(os.path.join(GOLDEN, 'F7_order_embedding.npz'))
    for tag, (lv, b) in (('toy2', (3, 2)), ('toy3', (4, 3))):
        g = embed_toy.ToyGraph(levels=lv, branching_factor=b)
        assert g.levels == f[tag + '_levels'].tolist()
        assert sorted(g.edges) == [tuple(e) for e in f[tag + '_edges'].tolist()]


def test_ethec_labelmap_fixture():
    import json
    pkg = os.path.join(os.path.dirname(GOLDEN), '..', 'learning_embeddings_amd', 'data', 'ethec_hierarchy.json')
    assert json.load(open(pkg)) == json.load(open(os.path.join(GOLDEN, 'F9_ethec_hierarchy.json')))   # the package's copy IS the fixture
    lm = SyntheticLabelMap.ethec()
    assert lm.levels == [6, 21, 135, 561] and lm.n_classes == 723 and len(lm.edges) == 717
    assert lm.level_start == [0, 6, 27, 162] and lm.level_stop == [6, 27, 162, 723]


def _fake_loaders(lm, n_train, n_val):
    par = lm.parents()
    def rec(j):
        leaf = lm.level_start[-1] + j % lm.levels[-1]; chain = [leaf]
        while chain[-1] in par:
            chain.append(par[chain[-1]][0])
        chain = chain[::-1]
        return [c - lm.level_start[l] for l, c in enumerate(chain)]
    def loader(lo, hi, bs=4):
        out = []
        for s in range(lo, hi, bs):
            js = list(range(s, min(s + bs, hi)))
            out.append({'level_labels': np.array([rec(j) for j in js]), 'image_filename': ['img_%06d' % j for j in js],
                        'path_to_image': [torch.full((3, 8, 8), float(j)) for j in js]})
        return out
    return {'train': loader(0, n_train), 'val': loader(n_train, n_train + n_val), 'test': loader(n_train + n_val, n_train + 2 * n_val)}


def test_create_combined_graphs_matches_dense_construction():
    lm = SyntheticLabelMap([2, 4, 8])
    dl = _fake_loaders(lm, 24, 6)
    gd = create_combined_graphs(dl, lm, pick_per_level=True)
    N = lm.n_classes
    assert gd['G_train_tc'].size() == len(transitive_closure_edges(lm)) + 24 * 3
    assert gd['mapping_node_to_ix']['img_000000'] == N and gd['mapping_ix_to_node'][N + 23] == 'img_000023'
    leaf = [lm.level_start[-1] + j % lm.levels[-1] for j in range(24)]
    A = O.dense_negative_adjacency(N, sorted(lm.edges), leaf)
    s = O.DenseSampler(A, lm.levels, pick_per_level=True, seed=0)
    g = gd['G_train_neg']; g.seed(0)
    rs = np.random.RandomState(0)
    for _ in range(300):
        side, node, lvl = int(rs.randint(2)), int(rs.randint(N + 24)), int(rs.randint(8))
        try:
            want = s.draw(side, node, lvl)
        except IndexError:
            with pytest.raises(IndexError):
                g.draw(side, node, lvl)
            continue
        assert g.draw(side, node, lvl) == want


def transitive_closure_edges(lm):
    G = DiGraph(); G.add_edges_from(sorted(lm.edges))
    return transitive_closure(G).edges()


def test_embedding_metrics_threshold_sweep_equals_bruteforce():
    rs = np.random.RandomState(0)
    p = torch.tensor(rs.rand(40) * 0.5); n = torch.tensor(rs.rand(200) * 0.8 + 0.1)
    m = EmbeddingMetrics(p, n, 0.0, 'val')
    best = m.calculate_metrics()
    brute = max((m.calculate_best(t) for t in np.unique(np.concatenate((p.numpy(), n.numpy())))), key=lambda r: r[0])
    assert abs(best[0] - brute[0]) < 1e-12


def test_global_batch_sampler_shards_partition_the_global_batch():
    for world in (1, 2, 4):
        parts = [list(GlobalBatchSampler(64, 4, shuffle=True, seed=3, rank=r, world=world)) for r in range(world)]
        ref = GlobalBatchSampler(64, 4, shuffle=True, seed=3, rank=0, world=world).global_batches()
        for i, gb in enumerate(ref):
            assert sum((parts[r][i] for r in range(world)), []) == gb


# ------------------------------------------------------------------------------------------------ world_size-2 gloo
def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _dp_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from learning_embeddings_amd import parallel as par
    from learning_embeddings_amd.hierarchy import SyntheticLabelMap as LM, NegativeGraph as NG
    r, lr_, w = par.init_process_group('gloo')
    torch.manual_seed(0)                                           # identical replicas
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    arena = par.FlatArena(net.parameters(), torch.device('cpu'))
    table_grad = torch.zeros(4, 3)
    red = par.GradientReducer(arena, bucket_mb=1e-5, extra=[table_grad])     # tiny buckets: several all-reduces
    X = torch.arange(48, dtype=torch.float32).reshape(8, 6) / 10.0
    lo, hi = par.shard_range(8)
    arena.zero_grad()
    net(X[lo:hi]).pow(2).sum().backward()                          # SUM loss over the shard
    table_grad += float(rank + 1)
    red.finish()
    buckets = red.time_buckets(reps=2)                             # bench.py's per-bucket timing leaves the gradients as they were
    assert len(buckets) == len(red.buckets) + 1 and all(b['ms'] >= 0 for b in buckets)
    # sampler: replicated mode -> shard of the global stream
    lm = LM([2, 4, 8])
    g = NG.from_labelmap(lm, n_images=16, pick_per_level=True, seed=0)
    def positives(s):
        b = np.arange(8); j = (s * 8 + b) % 16
        return (6 + j % 8).astype(np.int32), (14 + j).astype(np.int32)
    pf = par.NegativePrefetcher(g, positives, 3, mode='replicated')
    shards = [pf.next() for _ in range(3)]
    pf.close()
    q.put((rank, arena.grad.clone().numpy(), table_grad.numpy(), [s[2] for s in shards], [s[0] for s in shards]))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_dp_two_ranks_gloo_sum_allreduce_and_replicated_sampler(world):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs: p.join(60)
    # single-process reference: gradient of the SUM loss over the whole batch
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    arena = parallel.FlatArena(net.parameters(), torch.device('cpu'))
    X = torch.arange(48, dtype=torch.float32).reshape(8, 6) / 10.0
    net(X).pow(2).sum().backward()
    for r in range(world):
        assert np.allclose(res[r][1], arena.grad.numpy(), rtol=1e-5, atol=1e-6)      # SUM (not mean) of shard grads
        assert np.allclose(res[r][2], world * (world + 1) / 2)                         # extras ride along: 1 + 2 (+ 3 + 4)
    # negatives: concatenating the rank shards reproduces the single-process global stream bit for bit
    lm = SyntheticLabelMap([2, 4, 8])
    g = NegativeGraph.from_labelmap(lm, n_images=16, pick_per_level=True, seed=0)
    for s in range(3):
        b = np.arange(8); j = (s * 8 + b) % 16
        want = g.draw_batch((6 + j % 8).astype(np.int32), (14 + j).astype(np.int32), 3)
        got = np.concatenate([res[r][3][s] for r in range(world)])
        assert np.array_equal(got, want)


def test_pair_dataset_loads_image_files_and_collates(tmp_path):
    """ETHECHierarchyWithImages (oe_h.py:583-736) on real files: resize to 224, [0,1] float CHW, train-time transform only
    in __getitem__, get_image() without it; my_collate keeps python lists (oe_h.py:435-444)."""
    from PIL import Image
    lm = SyntheticLabelMap([2, 4])
    G = DiGraph()
    paths = {}
    for j in range(3):
        arr = (np.random.RandomState(j).rand(40, 60, 3) * 255).astype(np.uint8)
        p = os.path.join(str(tmp_path), 'im%d.png' % j); Image.fromarray(arr).save(p); paths['im%d' % j] = p
        G.add_edge(2 + j, 'im%d' % j); G.add_edge(0, 'im%d' % j)
    G.add_edge(0, 2)
    loaders = [{'image_filename': list(paths), 'path_to_image': list(paths.values())}]
    ds = oe_h.ETHECHierarchyWithImages(G, lm, imageless_dataloaders=loaders, transform=lambda t: t.flip(-1))
    assert len(ds) == G.size() == 7
    item = ds[0]
    assert item['original_to'] == 'im0' and item['from'] == 2 and item['to'].shape == (3, 224, 224)
    plain = ds.get_image('im0')
    assert plain.dtype == torch.float32 and 0.0 <= plain.min() and plain.max() <= 1.0
    # channel order is the reference's: cv2.imread gives B, G, R and ToPILImage does not swap (oe_h.py:700-712, 1463-1471), so tensor
    # channel 0 is BLUE.  A three-colour file decides it: left third pure red, middle pure green, right pure blue.
    tri = np.zeros((30, 90, 3), dtype=np.uint8); tri[:, :30, 0] = 255; tri[:, 30:60, 1] = 255; tri[:, 60:, 2] = 255
    p3 = os.path.join(str(tmp_path), 'tri.png'); Image.fromarray(tri).save(p3)
    ds.image_to_loc['tri'] = p3
    t3 = ds.get_image('tri')                                            # [C, 224, 224]; columns ~ [0, 74) red, [75, 149) green, [150, 224) blue
    assert t3[:, 112, 30].tolist() == [0.0, 0.0, 1.0]                    # red pixel   -> (B, G, R) = (0, 0, 1)
    assert t3[:, 112, 112].tolist() == [0.0, 1.0, 0.0]                   # green pixel -> (0, 1, 0)
    assert t3[:, 112, 190].tolist() == [1.0, 0.0, 0.0]                   # blue pixel  -> (1, 0, 0)
    assert torch.equal(item['to'], plain.flip(-1))                       # train transform applied only in __getitem__
    batch = oe_h.my_collate([ds[i] for i in range(len(ds))])
    assert isinstance(batch['from'], list) and batch['status'].tolist() == [1] * 7
    lab_edge = [i for i in range(len(ds)) if not isinstance(ds[i]['original_to'], str)]
    assert len(lab_edge) == 1 and ds[lab_edge[0]]['to'] == 2
    ds.set_levels_to_hide([1])                                           # hides every edge touching level 1 (labels 2..5)
    assert all(not (isinstance(u, int) and 2 <= u < 6) and not (isinstance(v, int) and 2 <= v < 6) for u, v in ds.edge_list)


# ---------------------------------------------------------------------------------------------------- F11: pair dataset + graph build
def _f11():
    return json.load(open(os.path.join(GOLDEN, 'F11_pair_dataset.json')))


def _graph_from(nodes, edges):
    g = DiGraph()
    for n in nodes:
        g.add_node(n)
    for u, v in edges:
        g.add_edge(u, v)
    return g


def test_create_combined_graphs_matches_reference_fixture():
    """oe_h.py:506-580 on the fixture's imageless loaders: node <-> index mapping (labels keep their id, images numbered in
    first-appearance order), every graph's edge set, the insertion-ordered graphs' edge ORDER, and the negative structure
    (dense A = 1 - TC - I in the reference, CSR here): draws from both are bit-identical."""
    f = _f11()
    lm = SyntheticLabelMap(f['levels'])
    loaders = {s: [{'level_labels': np.asarray(b['level_labels']), 'image_filename': b['image_filename']} for b in bl] for s, bl in f['loaders'].items()}
    gd = create_combined_graphs(loaders, lm, pick_per_level=True)
    n_nodes = len(f['mapping_ix_to_node'])
    assert [gd['mapping_ix_to_node'][i] for i in range(n_nodes)] == f['mapping_ix_to_node']
    assert gd['mapping_node_to_ix'] == {n: i for i, n in enumerate(f['mapping_ix_to_node'])}
    tup = lambda es: [tuple(e) for e in es]
    for key, name in (('graph', 'graph_edges'), ('G_train', 'G_train_edges'), ('G_val', 'G_val_edges'), ('G_test', 'G_test_edges'),
                      ('G_train_skeleton_full', 'G_train_skeleton_full_edges')):
        assert gd[key].edges() == tup(f[name]), key                      # insertion-ordered in both code bases
    assert sorted(gd['graph_tc'].edges()) == sorted(tup(f['graph_tc_edges']))
    # the closure's edge ORDER is networkx-version dependent in the reference (descendants() is a set in networkx 3): pin the set
    key = lambda e: (str(type(e[0])), str(e[0]), str(type(e[1])), str(e[1]))
    assert sorted(gd['G_train_tc'].edges(), key=key) == sorted(tup(f['G_train_tc_edges']), key=key)
    assert gd['G_train_tc'].size() == len(f['G_train_tc_edges'])
    # negative structure: same candidates / same stream as the reference's dense matrix
    A = np.asarray(f['neg_adjacency'], dtype=bool)
    dense = NegativeGraph.from_dense(A, lm.levels, pick_per_level=True, seed=0)
    csr = gd['G_train_neg']; csr.seed(0)
    N = lm.n_classes
    frm = np.array([0, 3, 9, 1, 5, 13, 2, 6], dtype=np.int32); to = np.array([N + 0, N + 3, N + 8, N + 4, N + 9, N + 5, 7, N + 11], dtype=np.int32)
    assert np.array_equal(csr.draw_batch(frm, to, 5), dense.draw_batch(frm, to, 5))


@pytest.mark.parametrize('half_half', [False, True])
def test_pair_dataset_matches_reference_fixture(half_half):
    """oe_h.py:583-736: edge lists, __len__, __getitem__ (half_half's map_ranges index mapping included) and
    set_levels_to_hide filtering, item by item against the reference's own dataset on the same graph."""
    from learning_embeddings_amd.oe_h_trainer import ETHECHierarchyWithImages
    f = _f11()
    lm = SyntheticLabelMap(f['levels'])
    g = _graph_from(f['G_train_tc_nodes'], f['G_train_tc_edges'])
    assert g.edges() == [tuple(e) for e in f['G_train_tc_edges']]
    d = ETHECHierarchyWithImages(g, lm, imageless_dataloaders=None, half_half=half_half)
    for hide in ([], [1], [0, 2], [0, 1, 2]):
        rec = f['dataset']['half_half=%s hide=%s' % (half_half, hide)]
        if hide or half_half:
            d.set_levels_to_hide(hide)
        assert len(d) == rec['len']
        if half_half:
            assert d.edge_list_ll == [tuple(e) for e in rec['edge_list_ll']] and d.edge_list_li == [tuple(e) for e in rec['edge_list_li']]
        else:
            assert d.edge_list == [tuple(e) for e in rec['edge_list']]
        for i, want in enumerate(rec['items']):
            if want == 'IndexError':
                with pytest.raises(IndexError):
                    d[i]
            else:
                it = d[i]
                assert [it['original_from'], it['original_to'], it['status']] == want and it['from'] == want[0] and it['to'] == want[1]


# ---------------------------------------------------------------------------------------------------- graph interchange (oe_h.py:563-571, 2250-2297)
class _PickledGraph:
    """Stands in for a gpickled networkx.DiGraph of a reference-written folder: anything with nodes() / edges()."""
    def __init__(self, g):
        self._n = list(g.nodes()); self._e = [tuple(e) for e in g.edges()]
    def nodes(self): return self._n
    def edges(self): return self._e


def _same_graph_dict(a, b, lm, n_img):
    from learning_embeddings_amd.oe_h_trainer import _GRAPH_KEYS
    for key in _GRAPH_KEYS:
        assert list(a[key].nodes()) == list(b[key].nodes()), key
        assert a[key].edges() == b[key].edges(), key
    assert a['mapping_ix_to_node'] == b['mapping_ix_to_node'] and a['mapping_node_to_ix'] == b['mapping_node_to_ix']
    N = lm.n_classes
    ga, gb = a['G_train_neg'], b['G_train_neg']; ga.seed(3); gb.seed(3)
    frm = np.array([0, 1, 3, 5, 2, 4, 6, 1], dtype=np.int32); to = np.array([N + 0, N + 5, N + 2, N + 7, 9, N + 11, N + 3, N + 1], dtype=np.int32)
    assert np.array_equal(ga.draw_batch(frm, to, 6), gb.draw_batch(frm, to, 6))


def test_save_and_load_combined_graphs_round_trip_and_reference_written_folder(tmp_path):
    """save_combined_graphs -> load_combined_graphs gives back the same graph_dict (node and edge ORDER, index mappings, negative sampler
    stream); a folder as the reference writes it (gpickled graph objects + the dense neg_adjacency.npy, no neg_structure.npz) loads to
    the same thing."""
    import pickle
    from learning_embeddings_amd.oe_h_trainer import save_combined_graphs, load_combined_graphs, GRAPH_FILES, _GRAPH_KEYS
    lm = SyntheticLabelMap([2, 4, 8])
    gd = create_combined_graphs(_fake_loaders(lm, 24, 6), lm, pick_per_level=True)
    own = tmp_path / 'own'
    save_combined_graphs(gd, str(own))
    assert sorted(os.listdir(own)) == sorted(list(GRAPH_FILES) + ['neg_structure.npz', 'neg_adjacency.npy'])
    back = load_combined_graphs(str(own), pick_per_level=True)
    _same_graph_dict(gd, back, lm, 24)
    # the reference's own format
    ref = tmp_path / 'ref'; os.makedirs(ref)
    for fname, key in zip(GRAPH_FILES, _GRAPH_KEYS):
        with open(ref / fname, 'wb') as f:
            pickle.dump(_PickledGraph(gd[key]), f)
    np.save(ref / 'neg_adjacency.npy', gd['G_train_neg'].to_dense())
    with pytest.raises(ValueError):
        load_combined_graphs(str(ref))                            # the dense matrix alone does not carry the level sizes
    back2 = load_combined_graphs(str(ref), labelmap=lm, pick_per_level=True)
    _same_graph_dict(gd, back2, lm, 24)
    # reference_compatible=True: what the reference's nx.read_gpickle (pickle.load) expects -- real networkx.DiGraph objects
    # (oe_h.py:2257-2263 calls .size() / .nodes() on them) and its dense neg_adjacency.npy
    nx = pytest.importorskip('networkx')
    rc = tmp_path / 'rc'
    save_combined_graphs(gd, str(rc), reference_compatible=True)
    for fname, key in zip(GRAPH_FILES, _GRAPH_KEYS):
        with open(rc / fname, 'rb') as f:
            G = pickle.load(f)
        assert isinstance(G, nx.DiGraph) and G.size() == gd[key].size()
        assert list(G.nodes()) == list(gd[key].nodes()) and list(G.edges()) == [tuple(e) for e in gd[key].edges()]
    assert np.array_equal(np.load(rc / 'neg_adjacency.npy'), gd['G_train_neg'].to_dense())
    _same_graph_dict(gd, load_combined_graphs(str(rc), pick_per_level=True), lm, 24)


def test_closure_graph_from_the_samplers_csr_equals_transitive_closure():
    """closure_graph (transitive closure read back from the C++ sampler's CSR through lec_sampler_tc_export) against the python
    transitive closure of the same skeleton: same edge set, skeleton edges first in their order."""
    from learning_embeddings_amd.oe_h_trainer import closure_graph
    lm = SyntheticLabelMap([3, 6, 12])
    gd = create_combined_graphs(_fake_loaders(lm, 30, 6), lm, pick_per_level=False)
    sk = gd['G_train_skeleton_full']
    got = closure_graph(sk, gd['G_train_neg'], gd['mapping_ix_to_node'], gd['mapping_node_to_ix'])
    want = transitive_closure(sk)
    key = lambda e: (str(type(e[0])), str(e[0]), str(type(e[1])), str(e[1]))
    assert sorted(got.edges(), key=key) == sorted(want.edges(), key=key)
    assert list(got.nodes()) == list(sk.nodes())
    for u in sk.nodes():
        n_sk = len(list(sk.successors(u)))
        assert list(got.successors(u))[:n_sk] == list(sk.successors(u))


# ---------------------------------------------------------------------------------------------------- F12: evaluation phase
def test_embedding_metrics_match_reference_fixture_including_nan_energies():
    """EmbeddingMetrics (oe_h.py:447-503) against the reference's own outputs (fixture F12, tests/golden/make_golden_eval.py): the 'val'
    threshold sweep (the reference: one pass per candidate threshold in a process pool; here sort + prefix sums) and the fixed-threshold
    branch, on energies with ties, zeros and NaN (the energies of the label row the reference's chunk loop leaves zero), and on the very
    energies its check_graph_embedding fed it.  Exact: same float64 results."""
    fx = json.load(open(os.path.join(GOLDEN, 'F12_eval_phase.json')))
    z = np.load(os.path.join(GOLDEN, 'F12_eval_phase.npz'))
    for i, case in enumerate(fx['embedding_metrics']):
        p, n = torch.from_numpy(z['em%d_pos' % i]), torch.from_numpy(z['em%d_neg' % i])
        got = EmbeddingMetrics(p, n, 0.0, 'val').calculate_metrics()
        assert np.array_equal(np.asarray(got, dtype=np.float64), np.asarray(case['val'])), (i, got, case['val'])
        got = EmbeddingMetrics(p, n, 0.35, 'test').calculate_metrics()
        assert np.array_equal(np.asarray(got, dtype=np.float64), np.asarray(case['fixed_0.35'])), (i, got)
    assert np.isnan(z['neg_e']).sum() == 41                      # the zero row as an apex: one NaN per other node
    got = EmbeddingMetrics(torch.from_numpy(z['pos_e']), torch.from_numpy(z['neg_e']), 0.0, 'val').calculate_metrics()
    assert np.array_equal(np.asarray(got, dtype=np.float64), np.asarray(fx['reconstruction']))


def test_bench_self_launch_starts_the_ranks_as_children(tmp_path):
    """`python bench.py --gpus N` without a torchrun environment (the driver's plain command): bench.self_launch starts N ranks through
    `python -m torch.distributed.run` as CHILD processes (never an exec), every rank sees RANK / WORLD_SIZE / MASTER_*, rank 0's single
    line reaches stdout, and a failing rank makes the launcher's exit code non-zero.  Run here with a stand-in script (no GPU)."""
    import subprocess
    script = tmp_path / 'rank_script.py'
    script.write_text('import os, sys, json\n'
                      'r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])\n'
                      'assert os.environ["MASTER_ADDR"] == "127.0.0.1" and "LOCAL_RANK" in os.environ\n'
                      'if "--fail" in sys.argv and r == 1: sys.exit(3)\n'
                      'if r == 0: print(json.dumps({"n_gpus": w, "argv": sys.argv[1:]}), flush=True)\n')
    code = ('import sys; sys.path.insert(0, %r); import bench; sys.exit(bench.self_launch(2, script=%r, argv=sys.argv[1:], extra_env={"LEC_BENCH_NO_GPU_PROBE": "1"}))'
            % (ROOT, str(script)))
    ok = subprocess.run([sys.executable, '-c', code, '--gpus', '2', '--steps', '3'], capture_output=True, text=True, timeout=300)
    assert ok.returncode == 0, ok.stderr[-2000:]
    lines = [l for l in ok.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1 and json.loads(lines[0]) == {'n_gpus': 2, 'argv': ['--gpus', '2', '--steps', '3']}
    bad = subprocess.run([sys.executable, '-c', code, '--fail'], capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0


def test_bench_launcher_parent_counts_gpus_without_torch_and_never_imports_it(tmp_path):
    """the parent of `python bench.py --gpus N` only spawns children -- it must not initialise HIP.  It counts GPUs from the visibility
    variables or the KFD topology in sysfs (count_gpus_no_hip), picks gloo when the ranks outnumber the devices, and 'torch' is not in sys.modules at the spawn
    (self_launch asserts it; the dry-launch record says so)."""
    import subprocess
    topo = tmp_path / 'nodes'
    for i, simd in enumerate([0, 0, 1024, 1024, 1024]):          # two CPU nodes, three GPUs
        (topo / str(i)).mkdir(parents=True)
        (topo / str(i) / 'properties').write_text('cpu_cores_count %d\nsimd_count %d\nmem_banks_count 1\n' % (64 if simd == 0 else 0, simd))
    code = ('import sys, json; sys.path.insert(0, %r); import bench\n'
            'print(json.dumps({"n": bench.count_gpus_no_hip(), "torch": "torch" in sys.modules}))\n'
            'sys.exit(bench.self_launch(4, script="x.py", argv=["--gpus", "4"]))' % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES', 'LEC_DIST_BACKEND')}
    env.update(LEC_KFD_TOPOLOGY=str(topo), LEC_BENCH_DRY_LAUNCH='1')
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    recs = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{')]
    assert recs[0] == {'n': 3, 'torch': False}
    assert recs[1]['torch_imported'] is False and recs[1]['dist_backend'] == 'gloo'          # 4 ranks on 3 devices: they share, gloo reduces
    assert recs[1]['dry_launch'][1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node' in recs[1]['dry_launch']
    env['HIP_VISIBLE_DEVICES'] = '0,1,2,3,4,5,6,7'                  # the visibility variable wins over the topology
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120, env=env)
    recs = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{')]
    assert recs[0]['n'] == 8 and recs[1]['dist_backend'] is None and recs[1]['torch_imported'] is False


def test_bench_watchdog_exits_nonzero_when_a_region_never_finishes():
    """#5(d): a rank stuck in a region (a collective that never completes) exits 3 by itself; a region that finishes disarms the watchdog."""
    import subprocess
    code = ('import sys, time; sys.path.insert(0, %r); import bench\n'
            'd = bench.start_watchdog(0.3, "ok region"); d(); time.sleep(0.6)\n'
            'bench.start_watchdog(0.3, "stuck region"); time.sleep(30)' % ROOT)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3 and 'WATCHDOG' in r.stderr and 'stuck region' in r.stderr and 'ok region' not in r.stderr


def test_bench_py_calls_only_names_it_defines():
    """bench.py is the driver's contract and most of it only runs on a GPU box: every plain-name call in it must resolve to a function the file defines, a name
    it binds (assignment, import, argument, loop / with / comprehension target) or a builtin -- a helper lost in an edit is caught here, not at round end."""
    import ast, builtins
    src = open(os.path.join(ROOT, 'bench.py')).read()
    tree = ast.parse(src)
    bound = set(dir(builtins))
    for n in ast.walk(tree):
        if isinstance(n, (ast.FunctionDef, ast.ClassDef)):
            bound.add(n.name)
            if isinstance(n, ast.FunctionDef):
                for a in n.args.args + n.args.kwonlyargs + ([n.args.vararg] if n.args.vararg else []) + ([n.args.kwarg] if n.args.kwarg else []):
                    bound.add(a.arg)
        elif isinstance(n, ast.Lambda):
            for a in n.args.args:
                bound.add(a.arg)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            for a in n.names:
                bound.add((a.asname or a.name).split('.')[0])
        elif isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store):
            bound.add(n.id)
        elif isinstance(n, ast.ExceptHandler) and n.name:
            bound.add(n.name)
    missing = sorted({n.func.id for n in ast.walk(tree) if isinstance(n, ast.Call) and isinstance(n.func, ast.Name) and n.func.id not in bound})
    assert not missing, missing


def test_conv_macs_matches_the_analytic_resnet_figures():
    """resnet.conv_macs (the denominator of bench.py's whole-CNN roofline line): SURVEY.md 8(d) quotes 1.8136 GMAC (ResNet-18) and 4.0872 GMAC
    (ResNet-50) per forward at 224 x 224, convolutions + the classifier's 1000-way fc excluded / included as the walk finds it."""
    from learning_embeddings_amd.resnet import resnet18, resnet50, conv_macs
    for net, gmac in ((resnet18(num_classes=10), 1.8136), (resnet50(num_classes=10), 4.0872)):
        m = conv_macs(net, 224)
        assert abs(m / 1e9 - gmac) < 0.01 * gmac, (m, gmac)
