#!/usr/bin/env python3
"""Generate fixture F10 (the Euclidean entailment-cone sibling) by IMPORTING the reference.

Runs ONLY in the build container (needs /root/reference); the output `F10_euclidean_cone.npz` is data: inputs and the
reference's outputs on them.

    python tests/golden/make_golden_oe.py

Reference entry points exercised (file:line under /root/reference):
  network/oe.py:721-739    EuclideanConesWithImagesHypernymLoss.E_operator (+autograd), K = 3.0
  network/oe.py:51-80      Embedder.forward / soft_clip (+autograd)
  network/oe.py:225-240    FeatCNN18.soft_clip
  network/oe.py:810-873    criterion.forward (train) + backward
"""
import importlib, os, random, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                                            # noqa: E402  (stubs + synthetic-graph helpers)


def main():
    import torch
    torch.set_num_threads(4)
    mg.import_reference()
    oe = importlib.import_module('network.oe')
    K = 3.0
    out = {'K': np.float64(K)}
    crit0 = oe.EuclideanConesWithImagesHypernymLoss(mg.SynthLabelMap([2, 8]), 1, {}, 0.01, False, K=K)
    emb0 = oe.Embedder(2, mg.SynthLabelMap([2, 8]), None, K=K)

    # ---- E_operator + autograd.  Points are soft-clipped first (|p| >= K, as the trainer guarantees); edge rows:
    # y on the cone axis (theta = -1 -> E = 0), y == x (normalize(0) = 0), y behind the apex, |x| only just above K.
    for D in (2, 10, 128):
        g = torch.Generator().manual_seed(100 + D)
        P = 96
        x = emb0.soft_clip(torch.randn(P, D, generator=g) * torch.rand(P, 1, generator=g) * 2.0)
        y = emb0.soft_clip(torch.randn(P, D, generator=g) * torch.rand(P, 1, generator=g) * 4.0)
        y[0] = x[0] * 1.7                                            # on the axis, outward
        y[1] = x[1].clone()                                          # coincident
        y[2] = x[2] * 0.2                                            # towards the origin (theta = +1)
        x[3] = x[3] / x[3].norm() * (K + 1e-4)                       # psi ~ 0: widest cone
        y[4] = x[4] * 1.5 + 0.01 * torch.randn(D, generator=g)       # well inside the cone -> E = 0, zero gradient
        x = x.clone().requires_grad_(True); y = y.clone().requires_grad_(True)
        E = crit0.E_operator(x, y)
        gE = torch.rand(P, generator=g) + 0.5
        E.backward(gE)
        x64 = x.detach().double().requires_grad_(True); y64 = y.detach().double().requires_grad_(True)
        E64 = crit0.E_operator(x64, y64); E64.backward(gE.double())
        out.update({'x%d' % D: mg.t2n(x), 'y%d' % D: mg.t2n(y), 'E%d' % D: mg.t2n(E), 'gE%d' % D: mg.t2n(gE),
                    'gx%d' % D: mg.t2n(x.grad), 'gy%d' % D: mg.t2n(y.grad), 'E64_%d' % D: mg.t2n(E64),
                    'gx64_%d' % D: mg.t2n(x64.grad), 'gy64_%d' % D: mg.t2n(y64.grad)})
        print('E_operator D=%d: E in [%.4f, %.4f], zeros %d' % (D, float(E.min()), float(E.max()), int((E == 0).sum())))

    # ---- Embedder.forward (gather + soft_clip) and FeatCNN18.soft_clip with autograd
    lmap = mg.SynthLabelMap([8, 64, 384, 1544])
    torch.manual_seed(0)
    model = oe.Embedder(10, lmap, None, K=K)
    g = torch.Generator().manual_seed(7)
    idx = torch.randint(0, lmap.n_classes, (200,), generator=g)
    idx[:4] = idx[4:8]                                               # duplicates: dense gradient accumulates
    o = model(idx)
    go = torch.randn(o.shape, generator=g)
    o.backward(go)
    out.update({'emb_W': mg.t2n(model.embeddings.weight), 'emb_idx': mg.t2n(idx), 'emb_out': mg.t2n(o), 'emb_gout': mg.t2n(go),
                'emb_gW': mg.t2n(model.embeddings.weight.grad)})
    raw = (torch.randn(64, 10, generator=g) * 0.5).requires_grad_(True)
    sc = oe.FeatCNN18.soft_clip(emb0, raw)                           # same formula, `self.K` is all it reads
    gsc = torch.randn(sc.shape, generator=g)
    sc.backward(gsc)
    out.update({'img_raw': mg.t2n(raw), 'img_out': mg.t2n(sc), 'img_gout': mg.t2n(gsc), 'img_graw': mg.t2n(raw.grad)})

    # ---- full criterion.forward (train) + backward on a scripted batch (same recipe as F5 's3')
    n_img, B, Kneg, D, alpha, ppl = 256, 48, 5, 10, 1.6, True
    N, names, A, n2i, i2n = mg.build_joint_graph(lmap.levels, lmap.edges, n_img)
    model.embeddings.weight.grad = None
    R = (torch.randn(n_img, D, generator=torch.Generator().manual_seed(5)) * 0.3).requires_grad_(True)

    class Net(torch.nn.Module):                                      # stand-in CNN: identity + FeatCNN18.soft_clip
        K = 3.0
        def forward(self, x):
            return oe.FeatCNN18.soft_clip(self, x)

    class DL:
        def get_image(self, fname):
            return R[n2i[fname] - N]

    crit = oe.EuclideanConesWithImagesHypernymLoss(lmap, Kneg, {}, alpha, ppl, K=K, use_CNN=True)
    crit.set_negative_graph(A, n2i, i2n); crit.set_dataloader(DL())
    rs = np.random.RandomState(3)
    of, ot = [], []
    leaf_start = N - lmap.levels[-1]
    par = mg.label_parents(lmap.levels, lmap.edges)
    for b in range(B):
        if b % 4 == 3:
            v = int(rs.randint(lmap.level_start[1], N)); u = par[v][0]
            if rs.rand() < 0.5 and u in par: u = par[u][0]
            of.append(int(u)); ot.append(int(v))
        else:
            j = int(rs.randint(n_img)); lab = leaf_start + (j % lmap.levels[-1])
            for _ in range((len(lmap.levels) - 1) - (b % len(lmap.levels))):
                lab = par[lab][0]
            of.append(int(lab)); ot.append(names[j])
    inputs_to = [R[n2i[t] - N] if isinstance(t, str) else t for t in ot]
    random.seed(0)
    loss, e_pos, e_neg = crit(model, Net(), list(of), inputs_to, of, ot, torch.ones(B), 'train')
    loss.backward()
    random.seed(0)
    neg = np.zeros((B, 2 * Kneg), dtype=np.int64)
    for b in range(B):
        for p in range(Kneg):
            neg[b, p] = crit.sample_negative_edge(u=of[b], v=None, level_id=p)
            neg[b, p + Kneg] = crit.sample_negative_edge(u=None, v=ot[b], level_id=p)
    out.update({'c_levels': np.array(lmap.levels), 'c_edges': np.array(sorted(lmap.edges)), 'c_n_images': np.int64(n_img),
                'c_W': mg.t2n(model.embeddings.weight), 'c_R': mg.t2n(R),
                'c_from': np.array([n2i[x] for x in of]), 'c_to': np.array([n2i[x] for x in ot]), 'c_neg': neg,
                'c_loss': mg.t2n(loss), 'c_e_pos': mg.t2n(e_pos), 'c_e_neg': mg.t2n(e_neg),
                'c_gW': mg.t2n(model.embeddings.weight.grad), 'c_gR': mg.t2n(R.grad),
                'c_alpha': np.float64(alpha), 'c_Kneg': np.int64(Kneg)})
    print('criterion: loss', float(loss), 'live negatives', int((mg.t2n(e_neg) < alpha).sum()))
    np.savez_compressed(os.path.join(HERE, 'F10_euclidean_cone.npz'), **out)
    print('F10 done')


if __name__ == '__main__':
    main()
