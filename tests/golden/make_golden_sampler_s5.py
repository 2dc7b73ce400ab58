"""Generates F4b: the reference's negative-index stream on config 5's hierarchy (BUILD CONTAINER ONLY: imports /root/reference).

  python tests/golden/make_golden_sampler_s5.py        (about 3 GB of memory for the dense matrix, a few minutes)

What the reference computes here (network/oe_h.py, nothing else):
  :799-809   set_negative_graph(A, node->ix, ix->node)   A = dense bool [N+M, N+M], 1 = not a transitive-closure edge, diagonal 0
  :849-902   sample_negative_edge(u | v, level_id)       np.where over a row / column of A, level window, random.choice
  :940-957   the criterion's loop over a batch           per positive b, per pass p < K: u fixed (slot 2K b + p), then v fixed (slot 2K b + p + K)

Hierarchy S5: 8 levels [2, 8, 32, 128, 512, 2048, 8192, 39078] = 50 000 labels, child c of level l under parent
floor(c n_{l-1} / n_l); M = 4 096 images, image j under leaf (j n_leaf) // M (engine.StepEngine spreads the images like this when
there are far fewer images than leaves: `j mod n_leaf` would hang every image under the first root and leave slot L without candidates).

Outputs (data only: inputs + the indices the reference returned):
  F4b_sampler_s5.json        scripted calls: every level_id slot 0..8 (and wrapped ids up to 2 (L+1)) on both sides, label and image end points,
                             pick_per_level on / off, hidden-level remaps -- including hide sets whose remaining slots CPython's set iterates
                             out of ascending order (oe_h.py:854 `list(set(range(L+1)) - set(hidden))[i]`: {8, 1, 2, 3} for hidden {0,4,5,6,7})
  F4b_sampler_s5_step0.npz   the whole first batch of config 5 as engine.StepEngine('cfg5') builds it: B = 256 positives x 2K = 512 negatives,
                             random.seed(0), through the loop of :940-957
"""
import json, os, random, sys, time
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, SynthLabelMap      # noqa: E402

S5 = [2, 8, 32, 128, 512, 2048, 8192, 39078]
M = 4096
B, K = 256, 256


def build(levels, n_images):
    lmap = SynthLabelMap(levels)
    N, L = lmap.n_classes, len(levels)
    par = {}
    for u, v in lmap.edges:
        par[v] = u                                           # a tree: one parent per label
    anc = [[] for _ in range(N)]
    for v in range(N):                                       # parents are numbered below their children (level order)
        if v in par:
            anc[v] = [par[v]] + anc[par[v]]
    leaf0, nleaf = lmap.level_start[-1], levels[-1]
    img_leaf = (np.arange(n_images, dtype=np.int64) * nleaf) // n_images
    n = N + n_images
    A = np.ones((n, n), dtype=bool)
    for v in range(N):
        if anc[v]:
            A[anc[v], v] = 0
    for j in range(n_images):
        leaf = leaf0 + int(img_leaf[j])
        A[leaf, N + j] = 0
        A[anc[leaf], N + j] = 0
    np.fill_diagonal(A, 0)
    names = ['img_%06d' % j for j in range(n_images)]
    node_to_ix = {i: i for i in range(N)}
    node_to_ix.update({names[j]: N + j for j in range(n_images)})
    ix_to_node = {v: k for k, v in node_to_ix.items()}
    # the engine's positives of step 0 (engine.StepEngine.positives): image b with its ancestor at level b mod L
    b = np.arange(B)
    leaf = leaf0 + img_leaf[b]
    chain = np.array([[int(l)] + anc[int(l)] for l in leaf])[:, ::-1]            # [B, L], root first
    pos_from = chain[b, b % L].astype(np.int64)
    pos_to = (N + b).astype(np.int64)
    return lmap, A, names, node_to_ix, ix_to_node, img_leaf, pos_from, pos_to


def main():
    t0 = time.time()
    oe_h = import_reference()['oe_h']
    lmap, A, names, n2i, i2n, img_leaf, pos_from, pos_to = build(S5, M)
    N, L = lmap.n_classes, len(S5)
    print('dense A built: %d x %d, %.1f s' % (A.shape[0], A.shape[1], time.time() - t0))

    out = {'levels': S5, 'n_images': M, 'image_leaf_rule': '(j * n_leaf) // n_images', 'cases': []}
    hides = [[], [1], [0, 2], [0, 4, 5, 6, 7], [0, 2, 3, 4, 5, 6, 7], [2, 3, 4, 5, 6, 7], [8], [3, 8]]
    for ppl in (True, False):
        for hide in hides:
            if not ppl and hide not in ([], [1]):
                continue
            crit = oe_h.EuclideanConesWithImagesHypernymLoss(lmap, K, {}, 0.01, ppl, K=0.1, use_CNN=True)
            crit.set_negative_graph(A, n2i, i2n)
            crit.set_levels_to_hide(hide)
            rs = np.random.RandomState(5000 + 100 * int(ppl) + 7 * len(hide) + sum(hide))
            calls, outs = [], []
            random.seed(0)
            # every slot id 0 .. 2 (L+1) - 1 on both sides for a label of every level and for images, then random calls
            script = []
            for level_id in range(2 * (L + 1)):
                for side in (0, 1):
                    lv = level_id % L
                    script.append((side, int(lmap.level_start[lv] + rs.randint(S5[lv])), level_id))
                    script.append((side, names[int(rs.randint(M))], level_id))
            for _ in range(120):
                is_img = rs.rand() < 0.4
                node = names[int(rs.randint(M))] if is_img else int(rs.randint(N))
                script.append((int(rs.randint(2)), node, int(rs.randint(0, 3 * (L + 1)))))
            for side, node, level_id in script:
                try:
                    r = crit.sample_negative_edge(u=node, v=None, level_id=level_id) if side == 0 else \
                        crit.sample_negative_edge(u=None, v=node, level_id=level_id)
                except IndexError:
                    r = -1                                    # random.choice on an empty list raises
                calls.append([side, n2i[node], level_id]); outs.append(int(r))
            out['cases'].append({'pick_per_level': ppl, 'levels_to_hide': hide, 'calls': calls, 'out': outs})
            print('case ppl=%s hide=%s: %d calls, %d empty, %.1f s' % (ppl, hide, len(calls), sum(o == -1 for o in outs), time.time() - t0))
    with open(os.path.join(HERE, 'F4b_sampler_s5.json'), 'w') as f:
        json.dump(out, f)

    # the engine's own first batch through the loop of oe_h.py:940-957 (the sampler calls of criterion.forward, nothing else of it:
    # the embeddings of 131 072 pairs are not needed for the index stream)
    crit = oe_h.EuclideanConesWithImagesHypernymLoss(lmap, K, {}, 0.01, True, K=0.1, use_CNN=True)
    crit.set_negative_graph(A, n2i, i2n)
    neg = np.zeros((B, 2 * K), dtype=np.int32)
    random.seed(0)
    for b in range(B):
        u, v = int(pos_from[b]), names[int(pos_to[b]) - N]
        for p in range(K):
            neg[b, p] = crit.sample_negative_edge(u=u, v=None, level_id=p)
            neg[b, p + K] = crit.sample_negative_edge(u=None, v=v, level_id=p)
        if b % 32 == 31:
            print('step-0 batch: positive %d / %d, %.1f s' % (b + 1, B, time.time() - t0))
    after = [random.getrandbits(32) for _ in range(4)]       # where the MT19937 stream stands after the batch
    np.savez_compressed(os.path.join(HERE, 'F4b_sampler_s5_step0.npz'), levels=np.asarray(S5, dtype=np.int32), n_images=np.int64(M),
                        image_leaf=img_leaf.astype(np.int32), pos_from=pos_from.astype(np.int32), pos_to=pos_to.astype(np.int32),
                        K=np.int64(K), neg=neg, stream_after=np.asarray(after, dtype=np.uint32))
    print('F4b done: %.1f s' % (time.time() - t0))


if __name__ == '__main__':
    main()
