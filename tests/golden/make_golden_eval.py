#!/usr/bin/env python3
"""Generate fixture F12 (the evaluation phase) by IMPORTING the reference -- build container only.

    python tests/golden/make_golden_eval.py        # rewrites tests/golden/F12_eval_phase.npz / .json

Reference entry points exercised (file:line under /root/reference), each run AS IS on synthetic inputs:
  network/oe_h.py:1971-2178  JointEmbeddings.calculate_classification_metrics(phase) for phase in train / val / test
                             (its chunk loops `[ix:min(ix+bs, len-1)]` leave the LAST image row and the LAST label row zero, :1997-2011;
                             the zero label row's energies are NaN, which torch.topk(largest=False) ranks last)
  network/oe_h.py:2180-2247  JointEmbeddings.check_graph_embedding() (same chunk rule on the label rows, :2230-2234)
  network/oe_h.py:447-503    EmbeddingMetrics.calculate_metrics(): the 'val' threshold sweep and the fixed-threshold branch, also on
                             energies that contain NaN
The trainer object is created WITHOUT its constructor (which wants images on disk, TensorBoard, GitPython): the two methods read a
dozen attributes, set here.  Images are in-memory tensors behind `criterion.dataloader.get_image`; the image network is a linear
stand-in followed by the reference's own FeatCNN18.soft_clip; the label table holds rows of norm 0.3 .. 1.5 so that the cones are
narrow and few energies clamp to zero.  Inputs (loaders, table, stand-in weights, images) and the reference's outputs are stored.
The seed is searched so that no two energies that decide a top-5 set or the best threshold lie closer than 1e-4: the fixture then
pins the RESULT, not a tie-break.
"""
import json, os, sys, tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, SynthLabelMap  # noqa: E402

LEVELS = [6, 12, 24]
COUNTS = {'train': 52, 'val': 26, 'test': 27}
D = 10
KC = 0.1


def loaders(lm, bs=7):
    import torch
    L = len(lm.levels)
    out, k = {}, 0
    for split, n in COUNTS.items():
        recs = []
        for j in range(n):
            leaf = (j * 5 + k) % lm.levels[-1]                 # every leaf (hence every label) occurs in every split
            chain = [leaf]
            for l in range(L - 1, 0, -1):
                chain.append((chain[-1] * lm.levels[l - 1]) // lm.levels[l])
            recs.append(('%s_img_%03d.jpg' % (split, j), chain[::-1]))
        k += 1
        out[split] = [{'image': None, 'labels': None, 'level_labels': torch.tensor([c for _, c in recs[i:i + bs]], dtype=torch.long),
                       'image_filename': [f for f, _ in recs[i:i + bs]]} for i in range(0, n, bs)]
    return out


def build(oe_h, seed):
    import torch, networkx as nx
    lm = SynthLabelMap(LEVELS)
    dl = loaders(lm)
    nx.write_gpickle = lambda *a, **k: None
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            gd = oe_h.create_combined_graphs(dl, lm)
        finally:
            os.chdir(cwd)
    g = torch.Generator().manual_seed(seed)
    names = [f for s in ('train', 'val', 'test') for b in dl[s] for f in b['image_filename']]
    images = torch.rand(len(names), 3, 8, 8, generator=g)
    W = torch.randn(lm.n_classes, D, generator=g)
    W = W / W.norm(dim=1, keepdim=True) * (0.3 + 1.2 * torch.rand(lm.n_classes, 1, generator=g))
    lin_w = torch.randn(D, 192, generator=g) * 0.06
    lin_b = torch.randn(D, generator=g) * 0.1
    crit = oe_h.EuclideanConesWithImagesHypernymLoss(lm, 5, {}, 0.01, pick_per_level=True, K=KC, use_CNN=True)

    class DS:
        def get_image(self, fname):
            return images[names.index(fname)]
    crit.set_dataloader(DS())
    model = oe_h.Embedder(D, lm, None, K=KC)
    model.embeddings.weight.data = W.clone()

    class Net(torch.nn.Module):
        K = KC
        inner_radius = 2 * KC / (1 + np.sqrt(1 + 4 * KC * KC))
        def forward(self, x):
            return oe_h.FeatCNN18.soft_clip(self, x.flatten(1) @ lin_w.t() + lin_b)
    tr = object.__new__(oe_h.JointEmbeddings)
    tr.graph_dict, tr.embedding_dim, tr.use_CNN, tr.criterion, tr.img_feat_net, tr.model = gd, D, True, crit, Net(), model
    tr.device, tr.labelmap, tr.n_proc = torch.device('cpu'), lm, 2
    tr.check_graph_embedding_neg_graph, tr.epoch, tr.levels_to_hide_for_epoch, tr.hide_levels = None, 0, {}, False
    return tr, dl, names, images, W, lin_w, lin_b, lm


def margins_ok(tr, lm, names, images, lin_w, lin_b):
    """No near-ties where they would decide the result: the 6 smallest energies per (image, level) pairwise >= 1e-4 apart and > 1e-4
    (the reference's own embeddings, zero rows included)."""
    import torch
    with torch.no_grad():
        lab = tr.model(torch.arange(lm.n_classes)); lab[-1] = 0
        for phase in ('train', 'val', 'test'):
            imgs = [n for n in tr.graph_dict['G_' + phase] if type(n) == str]
            rep = torch.stack([tr.img_feat_net(images[names.index(n)][None])[0] for n in imgs]); rep[-1] = 0
            for i in range(len(imgs)):
                e = tr.criterion.E_operator(lab, rep[i:i + 1].repeat(lm.n_classes, 1))
                for l in range(len(lm.levels)):
                    v = e[lm.level_start[l]:lm.level_stop[l]]
                    v = torch.sort(v[~torch.isnan(v)])[0][:6]
                    if v[0] < 1e-4 or (v[1:] - v[:-1]).min() < 1e-4:
                        return False
    return True


def scalarise(m):
    out = {}
    for k, v in m.items():
        if isinstance(v, dict):
            out[str(k)] = scalarise(v)
        else:
            out[str(k)] = float(v)
    return out


def main():
    import torch
    mods = import_reference()
    oe_h = mods['oe_h']
    for seed in range(1000):
        tr, dl, names, images, W, lin_w, lin_b, lm = build(oe_h, seed)
        if margins_ok(tr, lm, names, images, lin_w, lin_b):
            break
    else:
        raise SystemExit('no seed without near-ties found')
    fx = {'seed': seed, 'levels': LEVELS, 'D': D, 'K_cone': KC, 'names': names,
          'loaders': {s: [{'level_labels': b['level_labels'].tolist(), 'image_filename': b['image_filename']} for b in bl] for s, bl in dl.items()},
          'classification': {}, 'image_is_a_member_of': None}
    for phase in ('train', 'val', 'test'):
        with torch.no_grad():
            m = tr.calculate_classification_metrics(phase)              # k = [1, 3, 5], the reference's default
        fx['classification'][phase] = scalarise(m)
        if phase == 'train':
            fx['image_is_a_member_of'] = {str(k): [int(x) for x in v] for k, v in tr.image_is_a_member_of.items()}
            img_rep_train = tr.img_rep[0].numpy().copy()
    with torch.no_grad():
        best = tr.check_graph_embedding()
    fx['reconstruction'] = [float(x) for x in best]
    # what the reference fed its EmbeddingMetrics (recomputed the way :2230-2241 does), for the host-side CPU test
    with torch.no_grad():
        le = torch.zeros(len(tr.nodes_in_G), D)
        le[:len(tr.nodes_in_G) - 1] = tr.model(torch.tensor(tr.nodes_in_G[:len(tr.nodes_in_G) - 1]))
        pos_e = tr.criterion.E_operator(le[tr.pos_u_list], le[tr.pos_v_list]).numpy()
        neg_e = tr.criterion.E_operator(le[tr.neg_u_list], le[tr.neg_v_list]).numpy()
    check = oe_h.EmbeddingMetrics(torch.from_numpy(pos_e), torch.from_numpy(neg_e), 0.0, 'val', n_proc=2).calculate_metrics()
    assert np.array_equal(np.asarray(check), np.asarray(best)), 'the recomputed energies are not the ones the reference used'
    # EmbeddingMetrics alone on scripted energies (ties, zeros, NaN in either list), both branches
    rs = np.random.RandomState(7)
    em_cases = []
    for case in range(4):
        p = np.abs(rs.randn(60)).astype(np.float32) * 0.4; n = (np.abs(rs.randn(300)) * 0.8 + 0.1).astype(np.float32)
        p[::7] = 0.0; n[::11] = 0.0; n[5] = p[3]
        if case % 2:
            n[::13] = np.nan
        if case == 3:
            p[::17] = np.nan
        val = oe_h.EmbeddingMetrics(torch.from_numpy(p), torch.from_numpy(n), 0.0, 'val', n_proc=2).calculate_metrics()
        fixed = oe_h.EmbeddingMetrics(torch.from_numpy(p), torch.from_numpy(n), 0.35, 'test').calculate_metrics()
        em_cases.append({'val': [float(x) for x in val], 'fixed_0.35': [float(x) for x in fixed]})
        fx.setdefault('_em_inputs', []).append((p, n))
    em_inputs = fx.pop('_em_inputs')
    fx['embedding_metrics'] = em_cases
    np.savez_compressed(os.path.join(HERE, 'F12_eval_phase.npz'), images=images.numpy(), W=W.numpy(), lin_w=lin_w.numpy(), lin_b=lin_b.numpy(),
                        pos_e=pos_e, neg_e=neg_e, img_rep_train=img_rep_train,
                        **{'em%d_pos' % i: p for i, (p, n) in enumerate(em_inputs)}, **{'em%d_neg' % i: n for i, (p, n) in enumerate(em_inputs)})
    with open(os.path.join(HERE, 'F12_eval_phase.json'), 'w') as f:
        json.dump(fx, f, separators=(',', ':'))
    print('F12 written: seed %d; val m-f1 %.4f hit@5 %.4f; reconstruction f1 %.4f at threshold %.4f; NaN negatives %d'
          % (seed, fx['classification']['val']['m-f1'], fx['classification']['val']['hit@5'], best[0], best[1], int(np.isnan(neg_e).sum())))


if __name__ == '__main__':
    main()
