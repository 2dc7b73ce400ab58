"""Label hierarchies as integer data + the negative-sampling graph.

`NegativeGraph` is the product's replacement of the reference's dense "negative adjacency" (oe_h.py:554-561 builds a
(N+M)^2 bool matrix A = 1 - TC - I; set_negative_graph oe_h.py:799-809 stores it; sample_negative_edge :849-902 scans a
row/column of it per draw).  Here the transitive closure lives as sorted CSR lists inside liblecone.so's host sampler
and a draw is one binary search; results are bit-identical to the reference's `random.choice` stream.
"""
import ctypes as C
import json
import os
import numpy as np

from . import _lib
from ._lib import lib, check


class SyntheticLabelMap:
    """Duck-typed labelmap (db.py:3461-3478 / embed_toy.py:33-62 contract: levels, level_names, level_start, level_stop,
    n_classes, classes, edges).  Deterministic tree: child c of level l has parent floor(c * n_{l-1} / n_l)."""

    def __init__(self, levels, edges=None, level_names=None):
        self.levels = [int(v) for v in levels]
        self.level_names = list(level_names) if level_names else ['l%d' % i for i in range(len(self.levels))]
        self.n_classes = sum(self.levels)
        self.classes = ['%s_%d' % (self.level_names[l], i) for l in range(len(self.levels)) for i in range(self.levels[l])]
        self.level_start, self.level_stop = [], []
        s = 0
        for n in self.levels:
            self.level_start.append(s); s += n; self.level_stop.append(s)
        if edges is None:
            edges = set()
            for l in range(1, len(self.levels)):
                for c in range(self.levels[l]):
                    p = (c * self.levels[l - 1]) // self.levels[l]
                    edges.add((self.level_start[l - 1] + p, self.level_start[l] + c))
        self.edges = set((int(u), int(v)) for u, v in edges)

    @classmethod
    def ethec(cls, path=None):
        """The real ETHEC label DAG (6/21/135/561 nodes, 717 edges) as integer data exported from data/db.py:1122-3468.  The
        package carries its own copy (learning_embeddings_amd/data/ethec_hierarchy.json); tests/golden/F9 is the fixture the
        tests pin it against."""
        if path is None:
            path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'ethec_hierarchy.json')
        with open(path) as f:
            d = json.load(f)
        return cls(d['levels'], edges=[tuple(e) for e in d['edges']], level_names=d['level_names'])

    def parents(self):
        par = {}
        for u, v in sorted(self.edges):
            par.setdefault(v, []).append(u)
        return par

    def leaf_start(self):
        return self.level_start[-1]


SYNTHETIC = {
    'S1': [2, 8],
    'S3': [8, 64, 384, 1544],
    'S5': [2, 8, 32, 128, 512, 2048, 8192, 39078],
}


def image_parents_by_leaf(labelmap, n_images):
    """Image j hangs under leaf (j mod n_leaf); the reference adds one (label, image) edge per level (oe_h.py:524-531),
    the closure is taken by the sampler, so listing the leaf alone is equivalent."""
    leaf0 = labelmap.level_start[-1]; nleaf = labelmap.levels[-1]
    ptr = np.arange(n_images + 1, dtype=np.int64)
    adj = (leaf0 + (np.arange(n_images, dtype=np.int64) % nleaf)).astype(np.int32)
    return ptr, adj


class NegativeGraph:
    """Host sampler handle (include/lecone.h section 5).  Nodes: labels [0, N) in level order, images [N, N+M)."""

    def __init__(self, levels, label_edges, image_ptr=None, image_adj=None, pick_per_level=False, labels_only=False, seed=0):
        self.levels = [int(v) for v in levels]
        self.n_labels = int(sum(self.levels))
        edges = np.ascontiguousarray(np.array(sorted(label_edges), dtype=np.int32).reshape(-1, 2))
        if image_ptr is None:
            image_ptr = np.zeros(1, dtype=np.int64); image_adj = np.zeros(0, dtype=np.int32)
        image_ptr = np.ascontiguousarray(image_ptr, dtype=np.int64); image_adj = np.ascontiguousarray(image_adj, dtype=np.int32)
        self.n_images = int(len(image_ptr) - 1)
        self.n_nodes = self.n_labels + self.n_images
        lv = np.ascontiguousarray(self.levels, dtype=np.int32)
        h = C.c_void_p()
        check(lib.lec_sampler_create(C.byref(h), lv.ctypes.data, len(self.levels), edges.ctypes.data, len(edges),
                                     image_ptr.ctypes.data, image_adj.ctypes.data, self.n_images,
                                     int(bool(pick_per_level)), 1 if labels_only else 0, int(seed)))
        self._h = h
        self.pick_per_level = bool(pick_per_level)
        self.labels_only = bool(labels_only)
        self._spec = (edges, image_ptr, image_adj)          # what save() writes: the structure, never the dense matrix

    @classmethod
    def from_labelmap(cls, labelmap, n_images=0, image_leaf=None, **kw):
        if image_leaf is not None:
            ptr = np.arange(len(image_leaf) + 1, dtype=np.int64); adj = np.asarray(image_leaf, dtype=np.int32)
        elif n_images:
            ptr, adj = image_parents_by_leaf(labelmap, n_images)
        else:
            ptr = adj = None
        return cls(labelmap.levels, labelmap.edges, ptr, adj, **kw)

    @classmethod
    def from_dense(cls, A, levels, **kw):
        """Drop-in for the reference's dense matrix (1 = negative edge, diag 0; labels first, then images)."""
        A = np.asarray(A).astype(bool)
        n = A.shape[0]; N = int(sum(levels))
        if A.shape != (n, n) or n < N:
            raise ValueError('negative adjacency must be square with at least n_labels rows')
        if N < n and not A[N:, :].sum() == (n - N) * (n - 1):
            raise ValueError('image nodes must have no outgoing transitive-closure edges')
        tc = ~A
        np.fill_diagonal(tc, False)
        lu, lv = np.nonzero(tc[:N, :N])
        edges = list(zip(lu.tolist(), lv.tolist()))
        ptr = np.zeros(n - N + 1, dtype=np.int64); adj = []
        for j in range(n - N):
            p = np.nonzero(tc[:N, N + j])[0]
            adj.append(p); ptr[j + 1] = ptr[j] + len(p)
        adj = np.concatenate(adj).astype(np.int32) if adj else np.zeros(0, np.int32)
        return cls(levels, edges, ptr, adj, **kw)

    def __del__(self):
        h = getattr(self, '_h', None)
        if h:
            lib.lec_sampler_destroy(h); self._h = None

    def seed(self, s=0):
        check(lib.lec_sampler_seed(self._h, int(s)))

    def set_levels_to_hide(self, levels):
        a = np.ascontiguousarray(list(levels), dtype=np.int32)
        check(lib.lec_sampler_set_levels_to_hide(self._h, a.ctypes.data, len(a)))

    def visible_slots(self):
        """The level slots a draw's `level_id` is mapped onto, in the reference's order (oe_h.py:854: CPython set order)."""
        out = np.empty(len(self.levels) + 1, dtype=np.int32); n = C.c_int()
        check(lib.lec_sampler_visible_slots(self._h, out.ctypes.data, C.byref(n)))
        return out[:n.value].tolist()

    def draw(self, side, node, level_id=0):
        """side 0: `u` fixed, corrupt the "to" end (row of A); side 1: `v` fixed (column of A)."""
        out = C.c_int32()
        check(lib.lec_sampler_draw(self._h, int(side), int(node), int(level_id), C.byref(out)))
        return out.value

    def draw_batch(self, pos_from, pos_to, K, out=None):
        f = np.ascontiguousarray(pos_from, dtype=np.int32); t = np.ascontiguousarray(pos_to, dtype=np.int32)
        B = len(f)
        if out is None:
            out = np.empty((B, 2 * K), dtype=np.int32)
        check(lib.lec_sampler_draw_batch(self._h, f.ctypes.data, t.ctypes.data, B, int(K), out.ctypes.data))
        return out

    def next_u32(self):
        out = C.c_uint32()
        check(lib.lec_sampler_next_u32(self._h, C.byref(out)))
        return out.value

    @property
    def tc_edges(self):
        return int(lib.lec_sampler_tc_edges(self._h))

    def closure_csr(self):
        """The transitive closure the sampler holds, as CSR over node indices: descendants of u (ascending) are
        adj[ptr[u]:ptr[u+1]]."""
        ptr = np.empty(self.n_nodes + 1, dtype=np.int64); adj = np.empty(self.tc_edges, dtype=np.int32)
        check(lib.lec_sampler_tc_export(self._h, ptr.ctypes.data, adj.ctypes.data))
        return ptr, adj

    # ---- on-disk form (replaces the reference's dense `neg_adjacency.npy`, oe_h.py:563: O((N+M)^2) bytes) ----------
    def save(self, path):
        """`neg_structure.npz`: levels, label edges [E, 2], image parents CSR -- O(N + M) bytes; the closure is rebuilt on load."""
        edges, ptr, adj = self._spec
        np.savez_compressed(path, levels=np.asarray(self.levels, dtype=np.int32), label_edges=edges, image_ptr=ptr, image_adj=adj,
                            pick_per_level=np.asarray(int(self.pick_per_level)), labels_only=np.asarray(int(self.labels_only)))

    @classmethod
    def load(cls, path, seed=0, pick_per_level=None):
        d = np.load(path)
        ppl = bool(int(d['pick_per_level'])) if pick_per_level is None else bool(pick_per_level)
        return cls(d['levels'].tolist(), [tuple(e) for e in d['label_edges'].tolist()], d['image_ptr'], d['image_adj'],
                   pick_per_level=ppl, labels_only=bool(int(d['labels_only'])), seed=seed)

    def to_dense(self):
        """The reference's dense negative adjacency A = 1 - TC - I (oe_h.py:554-561), for interchange with its
        `neg_adjacency.npy`; O((N+M)^2) bytes -- refuses above 30 000 nodes."""
        if self.n_nodes > 30000:
            raise ValueError('dense negative adjacency of %d nodes would take %.1f GB' % (self.n_nodes, self.n_nodes ** 2 / 1e9))
        ptr, adj = self.closure_csr()
        A = np.ones((self.n_nodes, self.n_nodes), dtype=bool)
        rows = np.repeat(np.arange(self.n_nodes), np.diff(ptr))
        A[rows, adj] = False
        np.fill_diagonal(A, False)
        return A
