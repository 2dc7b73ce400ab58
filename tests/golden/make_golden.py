#!/usr/bin/env python3
"""Generate the golden fixtures F1..F9 by IMPORTING the reference.

This script runs ONLY in the build container, where /root/reference exists.  It is the
"generating script committed next to the vectors": its outputs (small .npz/.json files in
this directory) are data -- inputs and the reference's outputs on them.  Nothing of the
reference's source travels.  The GPU box never runs this file.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz, *.json

Reference entry points exercised (file:line under /root/reference):
  F1  network/oe_h.py:811-833        EuclideanConesWithImagesHypernymLoss.E_operator (+autograd)
  F2  network/oe_h.py:51-110         Embedder.__init__/forward/soft_clip (+autograd)
  F3  network/oe_h.py:323-328        FeatCNN18.soft_clip ; network/oe_h.py:168-224 FeatNet.forward
  F4  network/oe_h.py:849-902        sample_negative_edge (python `random`, seed 0)
  F5  network/oe_h.py:904-967        criterion.forward (train) + backward
  F6  network/oe_h.py:1766-1771      lambda-rescale -> Adam -> soft_clip ; :1757-1762 RSGD variant
  F7  network/order_embeddings.py:818-824, 840-923   Euclidean order-embedding energy + forward
  F8  network/loss.py:5-38           MultiLevelCELoss
  F9  data/db.py:1117-3512,3565      ETHEC label hierarchy as integer data
"""
import sys, types, importlib, os, json, random
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


def import_reference():
    import matplotlib
    sys.path[:0] = [REF, os.path.join(REF, 'network')]

    class _Any:
        def __init__(s, *a, **k): pass
        def __call__(s, *a, **k): return _Any()
        def __getattr__(s, n): return _Any()

    def stub(name, **attrs):
        m = types.ModuleType(name); m.__dict__.update(attrs); sys.modules[name] = m; return m

    tv = stub('torchvision', __version__='stub')
    tv.datasets = stub('torchvision.datasets', CIFAR10=object, CIFAR100=object, ImageFolder=_Any)
    tv.models = stub('torchvision.models')
    tv.transforms = stub('torchvision.transforms', Compose=_Any, ToPILImage=_Any, Resize=_Any,
                         ToTensor=_Any, RandomHorizontalFlip=_Any)
    stub('tensorboardX', SummaryWriter=_Any); stub('git', Repo=_Any); stub('cv2')
    sk = stub('skimage'); sk.io = stub('skimage.io'); sk.transform = stub('skimage.transform')
    sk.segmentation = stub('skimage.segmentation', mark_boundaries=_Any)
    lm = stub('lime'); lm.lime_image = stub('lime.lime_image', LimeImageExplainer=_Any)
    matplotlib.use('Agg'); matplotlib.use = lambda *a, **k: None
    mods = {}
    for name in ['data.db', 'network.loss', 'network.oe_h', 'network.order_embeddings',
                 'network.order_embeddings_h', 'network.embed_toy']:
        mods[name.split('.')[-1]] = importlib.import_module(name)
    return mods


# ----------------------------------------------------------------------------- helpers (inputs only)
class SynthLabelMap:
    """Duck-typed labelmap (embed_toy.py:33-62 / db.py:3461-3478 contract); deterministic tree:
    child c of level l has parent floor(c * n_{l-1} / n_l)   (SURVEY.md 8d)."""
    def __init__(self, levels):
        self.levels = list(levels)
        self.level_names = ['l%d' % i for i in range(len(levels))]
        self.n_classes = sum(levels)
        self.classes = ['%s_%d' % (self.level_names[l], i) for l in range(len(levels)) for i in range(levels[l])]
        self.level_start, self.level_stop = [], []
        s = 0
        for n in levels:
            self.level_start.append(s); s += n; self.level_stop.append(s)
        self.edges = set()
        for l in range(1, len(levels)):
            for c in range(levels[l]):
                p = (c * levels[l - 1]) // levels[l]
                self.edges.add((self.level_start[l - 1] + p, self.level_start[l] + c))


def label_parents(levels, edges):
    par = {}
    for u, v in edges:
        par.setdefault(v, []).append(u)
    return par


def build_joint_graph(levels, edges, n_images):
    """Label DAG + images.  Image j hangs under leaf (j mod n_leaf) and all its ancestors
    (oe_h.py:520-531 adds one edge per level; :539 takes the transitive closure).
    Returns (n_labels, image names, dense negative adjacency A as oe_h.py:554-561, node<->ix maps)."""
    N = sum(levels)
    leaf_start = N - levels[-1]
    par = label_parents(levels, edges)
    anc = {}

    def ancestors(v):
        if v in anc:
            return anc[v]
        out = set()
        for p in par.get(v, []):
            out.add(p); out |= ancestors(p)
        anc[v] = out
        return out

    names = ['img_%06d' % j for j in range(n_images)]
    n = N + n_images
    A = np.ones((n, n), dtype=bool)
    for v in range(N):
        for a in ancestors(v):
            A[a, v] = 0
    for j in range(n_images):
        leaf = leaf_start + (j % levels[-1])
        A[leaf, N + j] = 0
        for a in ancestors(leaf):
            A[a, N + j] = 0
    np.fill_diagonal(A, 0)
    node_to_ix = {i: i for i in range(N)}
    node_to_ix.update({names[j]: N + j for j in range(n_images)})
    ix_to_node = {v: k for k, v in node_to_ix.items()}
    return N, names, A, node_to_ix, ix_to_node


def t2n(t):
    return t.detach().cpu().numpy()


def main():
    import torch
    torch.set_num_threads(4)
    M = import_reference()
    oe_h, oe, oeh, db, lossm, toy = (M['oe_h'], M['order_embeddings'], M['order_embeddings_h'],
                                     M['db'], M['loss'], M['embed_toy'])
    Kc = 0.1
    r_in = 2 * Kc / (1 + np.sqrt(1 + 4 * Kc * Kc))

    # ------------------------------------------------------------------ F9 ETHEC hierarchy
    lm = db.ETHECLabelMapMerged()
    ethec_edges = []
    for l, d in enumerate([lm.child_of_family_ix, lm.child_of_subfamily_ix, lm.child_of_genus_ix]):
        for p in sorted(d):
            for c in d[p]:
                ethec_edges.append([int(p + lm.level_start[l]), int(c + lm.level_start[l + 1])])
    with open(os.path.join(HERE, 'F9_ethec_hierarchy.json'), 'w') as f:
        json.dump({'levels': [int(x) for x in lm.levels], 'level_names': lm.level_names,
                   'edges': ethec_edges}, f)
    print('F9: levels', lm.levels, 'edges', len(ethec_edges))
    ethec = SynthLabelMap(lm.levels)
    ethec.edges = set(map(tuple, ethec_edges))
    ethec.level_names = list(lm.level_names)

    crit0 = oe_h.EuclideanConesWithImagesHypernymLoss(ethec, 5, {}, 0.01, True, K=Kc, use_CNN=True)

    # ------------------------------------------------------------------ F1 cone energy
    out = {}
    for D in (2, 10, 128):
        g = torch.Generator().manual_seed(100 + D)
        n = 96
        def ball(nrm):
            v = torch.randn(n, D, generator=g)
            return v / v.norm(dim=1, keepdim=True) * nrm.unsqueeze(1)
        xn = r_in + (0.999 - r_in) * torch.rand(n, generator=g)
        yn = r_in + (0.999 - r_in) * torch.rand(n, generator=g)
        x, y = ball(xn), ball(yn)
        # edge rows -------------------------------------------------------------
        y[0] = x[0] * (0.9 / x[0].norm())                   # y on the ray through x, further out: inside cone -> E = 0
        x[0] = x[0] * (0.3 / x[0].norm())
        y[1] = -x[1]                                        # antipodal: acos arg saturates at the lower clamp
        x[2] = x[2] * (0.01 / x[2].norm())                  # tiny apex norm: psi argument clamps at 1-1e-5
        y[3] = x[3] + 1e-4 * torch.randn(D, generator=g)    # near-coincident points
        x[4] = x[4] * (0.99999 / x[4].norm())               # apex at the rim
        y[5] = y[5] * (1.3 / y[5].norm())                   # image-like point outside the ball (FeatCNN18 output is not clipped)
        x[6] = x[6] * (r_in / x[6].norm())                  # apex exactly at the inner radius
        y[7] = x[7] * 0.5                                   # y between origin and x: behind the apex
        x = x.clone().requires_grad_(True); y = y.clone().requires_grad_(True)
        E = crit0.E_operator(x, y)
        gE = torch.rand(n, generator=g) + 0.5
        (E * gE).sum().backward()
        # the same reference function evaluated in float64 on the same float32 inputs: per-row conditioning yardstick
        x64 = x.detach().double().requires_grad_(True); y64 = y.detach().double().requires_grad_(True)
        E64 = crit0.E_operator(x64, y64)
        (E64 * gE.double()).sum().backward()
        out.update({'x_%d' % D: t2n(x), 'y_%d' % D: t2n(y), 'E_%d' % D: t2n(E), 'gE_%d' % D: t2n(gE),
                    'gx_%d' % D: t2n(x.grad), 'gy_%d' % D: t2n(y.grad), 'E64_%d' % D: t2n(E64),
                    'gx64_%d' % D: t2n(x64.grad), 'gy64_%d' % D: t2n(y64.grad)})
    out['K'] = np.float64(Kc)
    np.savez_compressed(os.path.join(HERE, 'F1_cone_energy.npz'), **out)
    print('F1 done; any NaN:', any(np.isnan(v).any() for v in out.values()))

    # ------------------------------------------------------------------ F2 Embedder
    torch.manual_seed(0)
    emb = oe_h.Embedder(10, ethec, None, K=Kc)
    W0 = t2n(emb.embeddings.weight).copy()
    g = torch.Generator().manual_seed(7)
    idx = torch.randint(0, ethec.n_classes, (200,), generator=g)
    with torch.no_grad():
        W = emb.embeddings.weight
        W[idx[0]] = W[idx[0]] / W[idx[0]].norm() * 20.0      # tanh saturates to 1.0f -> upper clip
        W[idx[1]] = 0.0                                      # zero row: +1e-15 path, lower clip
        W[idx[2]] = W[idx[2]] / W[idx[2]].norm() * 1e-9      # tiny row
        W[idx[3]] = W[idx[3]] / W[idx[3]].norm() * 6.0       # near-saturated
        W[idx[4]] = W[idx[4]] / W[idx[4]].norm() * 16.0      # clamp(.,15) active
    W1 = t2n(emb.embeddings.weight).copy()
    o = emb(idx)
    go = torch.randn(o.shape, generator=g)
    (o * go).sum().backward()
    np.savez_compressed(os.path.join(HERE, 'F2_embedder.npz'), W_init=W0, W=W1, idx=t2n(idx), out=t2n(o),
                        gout=t2n(go), gW=t2n(emb.embeddings.weight.grad), K=np.float64(Kc),
                        inner_radius=np.float64(emb.inner_radius), inner_radius_h=t2n(emb.inner_radius_h))
    print('F2 done; init row norms', np.linalg.norm(W0, axis=1).min(), np.linalg.norm(W0, axis=1).max())

    # ------------------------------------------------------------------ F3 FeatCNN18.soft_clip, FeatNet.forward
    class _Shell:  # FeatCNN18.soft_clip only reads self.inner_radius (oe_h.py:323-328)
        inner_radius = r_in
    g = torch.Generator().manual_seed(11)
    raw = (torch.randn(64, 10, generator=g) * torch.logspace(-3, 1, 64).unsqueeze(1)).requires_grad_(True)
    sc = oe_h.FeatCNN18.soft_clip(_Shell(), raw)
    gsc = torch.randn(sc.shape, generator=g)
    (sc * gsc).sum().backward()
    torch.manual_seed(1)
    fn = oe_h.FeatNet(None, input_dim=32, output_dim=10, K=Kc)
    fin = torch.randn(48, 32, generator=g).requires_grad_(True)
    fo = fn(fin)
    gfo = torch.randn(fo.shape, generator=g)
    (fo * gfo).sum().backward()
    np.savez_compressed(os.path.join(HERE, 'F3_image_proj.npz'), raw=t2n(raw), soft_clip=t2n(sc), gout=t2n(gsc),
                        graw=t2n(raw.grad), fn_w=t2n(fn.fc1.weight), fn_b=t2n(fn.fc1.bias), fn_in=t2n(fin),
                        fn_out=t2n(fo), fn_gout=t2n(gfo), fn_gin=t2n(fin.grad), fn_gw=t2n(fn.fc1.weight.grad),
                        fn_gb=t2n(fn.fc1.bias.grad), K=np.float64(Kc))
    print('F3 done')

    # ------------------------------------------------------------------ F4 sampler
    F4 = {'mt19937_seed0_u32': [], 'cases': []}
    random.seed(0)
    F4['mt19937_seed0_u32'] = [random.getrandbits(32) for _ in range(16)]
    random.seed(0)
    F4['choice_range2000'] = [random.choice(range(2000)) for _ in range(16)]
    random.seed(12345)
    F4['randbelow_mixed'] = [[n, random.randrange(n)] for n in (1, 2, 3, 7, 64, 65, 1000, 4096, 100000, 2**31 - 1, 2**32 - 5) for _ in range(3)]
    hier = {'S1': ([2, 8], 12), 'S3': ([8, 64, 384, 1544], 400), 'ETHEC': (lm.levels, 300)}
    for name, (levels, n_img) in hier.items():
        lmap = ethec if name == 'ETHEC' else SynthLabelMap(levels)
        N, names, A, n2i, i2n = build_joint_graph(lmap.levels, lmap.edges, n_img)
        L = len(lmap.levels)
        for ppl in (True, False):
            for hide in ([], [1], [0, 2]) if L >= 3 else ([],):
                crit = oe_h.EuclideanConesWithImagesHypernymLoss(lmap, 5, {}, 0.01, ppl, K=Kc, use_CNN=True)
                crit.set_negative_graph(A, n2i, i2n)
                crit.set_levels_to_hide(hide)
                rs = np.random.RandomState(hash((name, ppl, tuple(hide))) % (2**31) if False else (len(name) * 1000 + int(ppl) * 10 + len(hide)))
                calls, outs = [], []
                random.seed(0)
                for c in range(240):
                    side = int(rs.randint(2))             # 0: u fixed (corrupt v) ; 1: v fixed (corrupt u)
                    is_img = rs.rand() < 0.4
                    node = names[int(rs.randint(n_img))] if is_img else int(rs.randint(N))
                    level_id = int(rs.randint(0, 2 * (L + 1)))
                    if name == 'S1' and not is_img and side == 1 and node < 2 and ppl and (level_id % (L + 1)) == 0:
                        pass                              # candidates non-empty anyway (the other root)
                    try:
                        r = crit.sample_negative_edge(u=node, v=None, level_id=level_id) if side == 0 else \
                            crit.sample_negative_edge(u=None, v=node, level_id=level_id)
                    except IndexError:
                        r = -1                            # empty candidate list: python raises
                    calls.append([side, n2i[node], level_id]); outs.append(int(r))
                F4['cases'].append({'hierarchy': name, 'levels': [int(x) for x in lmap.levels],
                                    'edges': sorted([list(map(int, e)) for e in lmap.edges]), 'n_images': n_img,
                                    'pick_per_level': ppl, 'levels_to_hide': hide,
                                    'calls': calls, 'out': outs})
    with open(os.path.join(HERE, 'F4_sampler.json'), 'w') as f:
        json.dump(F4, f)
    print('F4 done:', len(F4['cases']), 'cases; empties:', sum(o == -1 for c in F4['cases'] for o in c['out']))

    # ------------------------------------------------------------------ F5 full criterion forward/backward (train)
    F5 = {}
    for tag, levels, n_img, B, Kneg, D, alpha, ppl in (('s3', [8, 64, 384, 1544], 256, 48, 5, 10, 0.01, True),
                                                        ('ethec', lm.levels, 128, 32, 3, 2, 0.05, False)):
        lmap = ethec if tag == 'ethec' else SynthLabelMap(levels)
        N, names, A, n2i, i2n = build_joint_graph(lmap.levels, lmap.edges, n_img)
        torch.manual_seed(0)
        model = oe_h.Embedder(D, lmap, None, K=Kc)
        g = torch.Generator().manual_seed(5)
        R = (torch.randn(n_img, D, generator=g) * 0.3).requires_grad_(True)   # raw "CNN outputs", one per image

        class Net(torch.nn.Module):                                           # stand-in CNN: identity + FeatCNN18.soft_clip
            inner_radius = r_in
            def forward(self, x):
                return oe_h.FeatCNN18.soft_clip(self, x)

        class DL:
            def get_image(self, fname):
                return R[n2i[fname] - N]

        crit = oe_h.EuclideanConesWithImagesHypernymLoss(lmap, Kneg, {}, alpha, ppl, K=Kc, use_CNN=True)
        crit.set_negative_graph(A, n2i, i2n); crit.set_dataloader(DL())
        rs = np.random.RandomState(3)
        of, ot = [], []
        leaf_start = N - lmap.levels[-1]
        par = label_parents(lmap.levels, lmap.edges)
        for b in range(B):
            if b % 4 == 3:                                                    # (label,label) TC edge
                v = int(rs.randint(lmap.level_start[1], N)); u = par[v][0]
                if rs.rand() < 0.5 and u in par: u = par[u][0]
                of.append(int(u)); ot.append(int(v))
            else:                                                             # (label,image) edge, label level cycles
                j = int(rs.randint(n_img)); lab = leaf_start + (j % lmap.levels[-1])
                for _ in range((len(lmap.levels) - 1) - (b % len(lmap.levels))):
                    lab = par[lab][0]
                of.append(int(lab)); ot.append(names[j])
        inputs_from = list(of)
        inputs_to = [R[n2i[t] - N] if isinstance(t, str) else t for t in ot]
        random.seed(0)
        loss, e_pos, e_neg = crit(model, Net(), inputs_from, inputs_to, of, ot, torch.ones(B), 'train')
        loss.backward()
        # re-derive the negative node lists the call consumed (same stream, same order; oe_h.py:940-957)
        random.seed(0)
        neg = np.zeros((B, 2 * Kneg), dtype=np.int64)
        for b in range(B):
            for p in range(Kneg):
                neg[b, p] = crit.sample_negative_edge(u=of[b], v=None, level_id=p)
                neg[b, p + Kneg] = crit.sample_negative_edge(u=None, v=ot[b], level_id=p)
        F5.update({tag + '_levels': np.array(lmap.levels), tag + '_edges': np.array(sorted(lmap.edges)),
                   tag + '_n_images': np.int64(n_img), tag + '_W': t2n(model.embeddings.weight), tag + '_R': t2n(R),
                   tag + '_from': np.array([n2i[x] for x in of]), tag + '_to': np.array([n2i[x] for x in ot]),
                   tag + '_neg': neg, tag + '_loss': t2n(loss), tag + '_e_pos': t2n(e_pos), tag + '_e_neg': t2n(e_neg),
                   tag + '_gW': t2n(model.embeddings.weight.grad), tag + '_gR': t2n(R.grad),
                   tag + '_alpha': np.float64(alpha), tag + '_K': np.float64(Kc), tag + '_Kneg': np.int64(Kneg),
                   tag + '_pick_per_level': np.bool_(ppl)})
        print('F5', tag, 'loss', float(loss), 'e_pos', tuple(e_pos.shape), 'e_neg', tuple(e_neg.shape))
    np.savez_compressed(os.path.join(HERE, 'F5_criterion.npz'), **F5)

    # ------------------------------------------------------------------ F6 table step (Adam variant and RSGD variant)
    F6 = {}
    s3 = SynthLabelMap([8, 64, 384, 1544])
    for D in (10, 2):
        torch.manual_seed(0)
        model = oe_h.Embedder(D, s3, None, K=Kc)
        W = model.embeddings.weight
        g = torch.Generator().manual_seed(21)
        with torch.no_grad():
            W[5] = W[5] / W[5].norm() * 0.9999                # close to the rim
            W[6] = W[6] / W[6].norm() * (r_in * 0.5)           # inside the inner radius -> lower clip
            W[7] = W[7] / W[7].norm() * 1.5                    # outside the ball -> upper clip; lambda_x negative
        W_before = t2n(W).copy()

        class T:                                               # the three methods only touch these attributes
            embedding_dim = D
            class criterion: inner_radius = r_in
        T.soft_clip = oe_h.JointEmbeddings.soft_clip; T.lambda_x = oe_h.JointEmbeddings.lambda_x
        T.mob_add = oe_h.JointEmbeddings.mob_add; T.exp_map_x = oe_h.JointEmbeddings.exp_map_x
        tr = T()
        opt = torch.optim.Adam([W], lr=1e-2)
        grads, afters = [], []
        for step in range(3):
            gr = torch.randn(W.shape, generator=g) * (torch.rand(W.shape[0], 1, generator=g) < 0.6).float()
            grads.append(t2n(gr).copy())
            opt.zero_grad(); W.grad = gr.clone()
            W.grad.data *= (1.0 / tr.lambda_x(W.data)) ** 2    # oe_h.py:1768
            opt.step()                                         # :1769
            W.data = tr.soft_clip(W.data)                      # :1771
            afters.append(t2n(W).copy())
        F6.update({'adam_W0_%d' % D: W_before, 'adam_grads_%d' % D: np.stack(grads), 'adam_W_%d' % D: np.stack(afters),
                   'adam_m_%d' % D: t2n(opt.state[W]['exp_avg']), 'adam_v_%d' % D: t2n(opt.state[W]['exp_avg_sq'])})
        # RSGD variant (oe_h.py:1761-1762; always-on in order_embeddings_h.py:764-775)
        Wr = torch.tensor(W_before)
        lr = 0.05
        gr = torch.randn(Wr.shape, generator=g)
        gg = gr.clone() * (1.0 / tr.lambda_x(Wr)) ** 2
        Wr_after = tr.exp_map_x(Wr.clone(), -lr * gg)
        F6.update({'rsgd_W0_%d' % D: W_before, 'rsgd_grad_%d' % D: t2n(gr), 'rsgd_W_%d' % D: t2n(Wr_after)})
    F6['lr_adam'] = np.float32(1e-2); F6['lr_rsgd'] = np.float32(0.05); F6['K'] = np.float64(Kc)
    np.savez_compressed(os.path.join(HERE, 'F6_table_step.npz'), **F6)
    print('F6 done')

    # ------------------------------------------------------------------ F7 Euclidean order embeddings on ToyGraph (config 1)
    tg = toy.ToyGraph(levels=3, branching_factor=2)            # levels [2,4] ... reference formula b**i, i=1..levels-1
    tg2 = toy.ToyGraph(levels=4, branching_factor=3)           # [3,9,27]
    F7 = {}
    for tag, g_ in (('toy2', tg), ('toy3', tg2)):
        N = g_.n_classes
        import networkx as nx
        G = nx.DiGraph(); G.add_edges_from(g_.edges)
        Gtc = nx.transitive_closure(G)
        A = np.ones((N, N), dtype=bool)
        for u, v in Gtc.edges(): A[u, v] = 0
        np.fill_diagonal(A, 0)
        ident = {i: i for i in range(N)}
        crit = oe.OrderEmbeddingLoss(g_, neg_to_pos_ratio=4, alpha=1.0, pick_per_level=True)
        crit.set_negative_graph(A, ident, ident)
        torch.manual_seed(0)
        model = oe.Embedder(embedding_dim=6, labelmap=g_)
        edges = sorted(Gtc.edges())
        rs = np.random.RandomState(1)
        sel = [edges[i] for i in rs.randint(0, len(edges), size=16)]
        frm = [int(e[0]) for e in sel]; to = [int(e[1]) for e in sel]
        random.seed(0)
        _, _, loss, e_pos, e_neg = crit(model, frm, to, torch.ones(len(frm)), 'train', 4)
        loss.backward()
        random.seed(0)
        neg = np.zeros((len(frm), 8), dtype=np.int64)
        for b in range(len(frm)):
            for p in range(4):
                neg[b, p] = crit.sample_negative_edge(u=frm[b], v=None, level_id=p)
                neg[b, p + 4] = crit.sample_negative_edge(u=None, v=to[b], level_id=p)
        F7.update({tag + '_levels': np.array(g_.levels), tag + '_edges': np.array(sorted(g_.edges)),
                   tag + '_W': t2n(model.embeddings.weight), tag + '_from': np.array(frm), tag + '_to': np.array(to),
                   tag + '_neg': neg, tag + '_loss': t2n(loss), tag + '_e_pos': t2n(e_pos), tag + '_e_neg': t2n(e_neg),
                   tag + '_gW': t2n(model.embeddings.weight.grad)})
    g = torch.Generator().manual_seed(2)
    x = torch.randn(64, 6, generator=g).requires_grad_(True); y = torch.randn(64, 6, generator=g).requires_grad_(True)
    E = oe.OrderEmbeddingLoss.E_operator(x, y); gE = torch.rand(64, generator=g)
    (E * gE).sum().backward()
    F7.update({'x': t2n(x), 'y': t2n(y), 'E': t2n(E), 'gE': t2n(gE), 'gx': t2n(x.grad), 'gy': t2n(y.grad)})
    np.savez_compressed(os.path.join(HERE, 'F7_order_embedding.npz'), **F7)
    print('F7 done')

    # ------------------------------------------------------------------ F8 MultiLevelCELoss (config 4)
    g = torch.Generator().manual_seed(4)
    Bc = 24
    logits = (torch.randn(Bc, lm.n_classes, generator=g) * 3).requires_grad_(True)
    lvl = torch.stack([torch.randint(0, n, (Bc,), generator=g) for n in lm.levels], dim=1)
    F8 = {}
    for tag, w in (('unw', None), ('w', [1.0, 0.5, 2.0, 4.0])):
        logits.grad = None
        crit = lossm.MultiLevelCELoss(lm, level_weights=w)
        ls = crit(logits, None, lvl)
        ls.backward()
        F8.update({tag + '_loss': t2n(ls), tag + '_glogits': t2n(logits.grad).copy()})
    F8.update({'logits': t2n(logits), 'level_labels': t2n(lvl), 'levels': np.array(lm.levels),
               'level_weights_w': np.array([1.0, 0.5, 2.0, 4.0], dtype=np.float32)})
    np.savez_compressed(os.path.join(HERE, 'F8_multilevel_ce.npz'), **F8)
    print('F8 done')


if __name__ == '__main__':
    main()
