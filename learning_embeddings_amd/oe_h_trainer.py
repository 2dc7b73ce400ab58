"""JointEmbeddings trainer + pair dataset + graph construction: host mirror of network/oe_h.py:447-736, 1318-2297.

The trainer keeps the reference's constructor and method surface and runs the step MI355X-style:

    images --CNN fwd (bf16/NHWC)--> raw feats --[ONE fused HIP kernel: projections, 1+2K cone energies per positive,
    hinge, loss, d/d table, d/d feats]--> CNN bwd --(bucketed RCCL SUM all-reduce, overlapped)--> ONE table-step kernel
    (lambda-rescale + Adam + clip) + ONE flat-arena Adam kernel for the CNN.

No host synchronisation inside the step; the negative sampler runs on the host in the reference's RNG order.
"""
import copy
import os
import random
import time

import numpy as np
import torch
import torch.nn as nn

from . import _lib, ops, parallel
from .hierarchy import NegativeGraph
from .image_store import ImageRef, ImageStore, decode_u8


# ------------------------------------------------------------------------------------------------ tiny graph type
class DiGraph:
    """The handful of networkx.DiGraph methods the reference's trainer touches (edges/nodes/size/add_edge/
    predecessors/successors), insertion-ordered like networkx.  A real networkx graph works wherever this does."""

    def __init__(self):
        self._succ, self._pred = {}, {}

    def add_node(self, n):
        if n not in self._succ:
            self._succ[n] = {}; self._pred[n] = {}

    def add_edge(self, u, v):
        self.add_node(u); self.add_node(v)
        self._succ[u][v] = True; self._pred[v][u] = True

    def add_edges_from(self, es):
        for u, v in es:
            self.add_edge(u, v)

    def nodes(self):
        return list(self._succ)

    def __iter__(self):
        return iter(self._succ)

    def __len__(self):
        return len(self._succ)

    def __contains__(self, n):
        return n in self._succ

    def edges(self, n=None):
        if n is not None:
            return [(n, v) for v in self._succ.get(n, {})]
        return [(u, v) for u in self._succ for v in self._succ[u]]

    def size(self):
        return sum(len(s) for s in self._succ.values())

    def successors(self, n):
        return list(self._succ[n])

    def predecessors(self, n):
        return list(self._pred[n])

    def has_edge(self, u, v):
        return u in self._succ and v in self._succ[u]


def transitive_closure(G):
    out = DiGraph()
    for n in G.nodes():
        out.add_node(n)
    memo = {}

    def desc(u):
        if u in memo:
            return memo[u]
        memo[u] = d = {}
        for v in G.successors(u):
            d[v] = True
            for w in desc(v):
                d[w] = True
        return d
    import sys
    sys.setrecursionlimit(max(10000, sys.getrecursionlimit()))
    for u in G.nodes():
        for v in desc(u):
            out.add_edge(u, v)
    return out


def closure_graph(skeleton, neg, mapping_ix_to_node, mapping_node_to_ix):
    """nx.transitive_closure(skeleton) (oe_h.py:539) from the closure liblecone's sampler already holds (CSR, built in C++):
    nodes in the skeleton's order, each node's skeleton edges first (in their order), then its remaining descendants by
    ascending node index.  (networkx leaves the order of the added edges unspecified: 2.2 adds them in DFS preorder, 3.x in
    set order; only the edge SET is pinned by fixture F11.)  Seconds for 50 000 labels + 100 000 images, where a memoised
    python recursion over dict-of-dict graphs takes minutes."""
    ptr, adj = neg.closure_csr()
    out = DiGraph()
    for n in skeleton.nodes():
        out.add_node(n)
    for u in skeleton.nodes():
        succ = out._succ[u]
        for v in skeleton._succ[u] if isinstance(skeleton, DiGraph) else skeleton.successors(u):
            succ[v] = True; out._pred[v][u] = True
        iu = mapping_node_to_ix[u]
        for iv in adj[ptr[iu]:ptr[iu + 1]].tolist():
            v = mapping_ix_to_node[iv]
            if v not in succ:
                succ[v] = True; out._pred[v][u] = True
    return out


GRAPH_FILES = ('G', 'G_tc', 'G_train', 'G_val', 'G_test', 'G_train_skeleton_full', 'G_train_tc')
_GRAPH_KEYS = ('graph', 'graph_tc', 'G_train', 'G_val', 'G_test', 'G_train_skeleton_full', 'G_train_tc')


def save_combined_graphs(graph_dict, path_to_folder, dense=None, reference_compatible=False):
    """The writer half of oe_h.py:563-571.  Same file names as the reference: the seven graphs `G`, `G_tc`, `G_train`, `G_val`,
    `G_test`, `G_train_skeleton_full`, `G_train_tc` -- by default each a pickle of {'nodes': [...], 'edges': [(u, v), ...]} in
    insertion order (version-independent where a pickled networkx object is not; load_combined_graphs reads it back, the reference
    does NOT: its nx.read_gpickle would hand `G.size()` a dict) -- plus the negative structure as `neg_structure.npz`
    (NegativeGraph.save: O(N + M) bytes).  dense=True (default: only up to 4 000 nodes) also writes the reference's own
    `neg_adjacency.npy`, the one file of the folder both sides read as is.
    reference_compatible=True: the seven files hold pickled `networkx.DiGraph` objects instead -- what the reference's
    nx.write_gpickle writes (it IS pickle.dump of the graph object, oe_h.py:565-571) and its nx.read_gpickle (:2257-2263) reads
    back, under the networkx version that wrote them; needs networkx importable and implies dense=True."""
    import pickle
    os.makedirs(path_to_folder, exist_ok=True)
    if reference_compatible:
        import networkx as nx
        dense = True
    for fname, key in zip(GRAPH_FILES, _GRAPH_KEYS):
        g = graph_dict[key]
        if reference_compatible:
            obj = nx.DiGraph()
            obj.add_nodes_from(g.nodes()); obj.add_edges_from(tuple(e) for e in g.edges())
        else:
            obj = {'nodes': list(g.nodes()), 'edges': [tuple(e) for e in g.edges()]}
        with open(os.path.join(path_to_folder, fname), 'wb') as f:
            pickle.dump(obj, f, protocol=4)
    neg = graph_dict['G_train_neg']
    if not isinstance(neg, NegativeGraph):
        neg = NegativeGraph.from_dense(np.asarray(neg), graph_dict['levels'])
    neg.save(os.path.join(path_to_folder, 'neg_structure.npz'))
    if dense or (dense is None and neg.n_nodes <= 4000):
        np.save(os.path.join(path_to_folder, 'neg_adjacency.npy'), neg.to_dense())


def _read_graph(path):
    import pickle
    with open(path, 'rb') as f:
        obj = pickle.load(f)                  # a reference-written file unpickles to a networkx.DiGraph (needs networkx importable)
    if isinstance(obj, dict) and 'edges' in obj:
        g = DiGraph()
        for n in obj['nodes']:
            g.add_node(n)
        g.add_edges_from(obj['edges'])
        return g
    g = DiGraph()                              # anything with nodes() / edges(): copy into the insertion-ordered type
    for n in obj.nodes():
        g.add_node(n)
    g.add_edges_from(obj.edges())
    return g


def load_combined_graphs(debug_or_path, labelmap=None, pick_per_level=False):
    """oe_h.py:2250-2297.  `debug_or_path`: the reference's bool (its two hard-wired relative folders) or a folder path.
    Reads folders written by save_combined_graphs AND by the reference (gpickled networkx graphs + dense neg_adjacency.npy;
    for those pass `labelmap`, whose level sizes the sampler needs).  Returns the reference's graph_dict keys."""
    print('Reading graphs from disk!')
    if isinstance(debug_or_path, (str, os.PathLike)):
        folder = debug_or_path
    else:
        folder = '../database/ETHEC/ETHECSmall_embeddings/graphs' if debug_or_path else '../database/ETHEC/ETHEC_embeddings/graphs'
    gd = {key: _read_graph(os.path.join(folder, fname)) for fname, key in zip(GRAPH_FILES, _GRAPH_KEYS)}
    mapping_ix_to_node = {}
    img_label = len(gd['graph'].nodes()) if labelmap is None else labelmap.n_classes
    for node in gd['G_train_tc'].nodes():                       # oe_h.py:2276-2283
        if isinstance(node, (int, np.integer)):
            mapping_ix_to_node[int(node)] = int(node)
        else:
            mapping_ix_to_node[img_label] = node; img_label += 1
    mapping_node_to_ix = {v: k for k, v in mapping_ix_to_node.items()}
    st = os.path.join(folder, 'neg_structure.npz')
    if os.path.exists(st):
        neg = NegativeGraph.load(st, seed=0, pick_per_level=pick_per_level)
    else:
        if labelmap is None:
            raise ValueError('a folder written by the reference holds only the dense neg_adjacency.npy: pass `labelmap` (its level sizes)')
        neg = NegativeGraph.from_dense(np.load(os.path.join(folder, 'neg_adjacency.npy')), labelmap.levels, pick_per_level=pick_per_level, seed=0)
    print('Graph with labels connected has {} edges, {} nodes'.format(gd['graph'].size(), len(gd['graph'].nodes())))
    print('Transitive closure of graphs with labels & images: train {}'.format(gd['G_train_tc'].size()))
    gd.update({'G_train_neg': neg, 'mapping_ix_to_node': mapping_ix_to_node, 'mapping_node_to_ix': mapping_node_to_ix})
    return gd


def create_combined_graphs(dataloaders, labelmap, pick_per_level=False):
    """oe_h.py:506-580 on the same inputs (dataloaders yielding {'level_labels': [b, L], 'image_filename': [b]}), but the
    negative structure is a CSR NegativeGraph instead of the dense (N+M)^2 matrix, nothing is pickled to disk, and
    label ids are the labelmap's (every label is a node even if no training image reaches it)."""
    G = DiGraph()
    L = len(labelmap.levels)
    for data_item in dataloaders['train']:
        ll = np.asarray(data_item['level_labels'])
        for level_id in range(L - 1):
            for s in range(ll.shape[0]):
                G.add_edge(int(ll[s, level_id]) + labelmap.level_start[level_id],
                           int(ll[s, level_id + 1]) + labelmap.level_start[level_id + 1])
    graphs = {'train': DiGraph(), 'val': DiGraph(), 'test': DiGraph()}
    G_train_tc_base = copy.deepcopy(G)
    image_leaf_parents = {}
    for split in ('train', 'val', 'test'):
        for data_item in dataloaders[split]:
            ll = np.asarray(data_item['level_labels'])
            for level_id in range(L):
                for s in range(ll.shape[0]):
                    lab = int(ll[s, level_id]) + labelmap.level_start[level_id]
                    fname = data_item['image_filename'][s]
                    graphs[split].add_edge(lab, fname)
                    if split == 'train':
                        G_train_tc_base.add_edge(lab, fname)
                        image_leaf_parents.setdefault(fname, []).append(lab)
    G_train_skeleton_full = copy.deepcopy(G_train_tc_base)
    n_labels = labelmap.n_classes
    mapping_ix_to_node = {i: i for i in range(n_labels)}
    img_label = n_labels
    for node in G_train_tc_base.nodes():                        # the closure has the skeleton's nodes in the skeleton's order
        if isinstance(node, str):
            mapping_ix_to_node[img_label] = node; img_label += 1
    mapping_node_to_ix = {v: k for k, v in mapping_ix_to_node.items()}
    names = [mapping_ix_to_node[i] for i in range(n_labels, img_label)]
    ptr = np.zeros(len(names) + 1, dtype=np.int64); adj = []
    for j, nm in enumerate(names):
        adj.extend(image_leaf_parents[nm]); ptr[j + 1] = len(adj)
    label_edges = [(u, v) for u, v in G.edges()]
    neg = NegativeGraph(labelmap.levels, label_edges, ptr, np.asarray(adj, dtype=np.int32), pick_per_level=pick_per_level, seed=0)
    G_train_tc = closure_graph(G_train_tc_base, neg, mapping_ix_to_node, mapping_node_to_ix)    # oe_h.py:539, closure from the C++ sampler
    return {'graph': G, 'graph_tc': transitive_closure(G), 'G_train': graphs['train'], 'G_val': graphs['val'],
            'G_test': graphs['test'], 'G_train_skeleton_full': G_train_skeleton_full, 'G_train_neg': neg,
            'mapping_ix_to_node': mapping_ix_to_node, 'mapping_node_to_ix': mapping_node_to_ix,
            'G_train_tc': G_train_tc}


# ------------------------------------------------------------------------------------------------ pair dataset
class ETHECHierarchyWithImages(torch.utils.data.Dataset):
    """oe_h.py:583-736: one positive edge (u, v) of the graph per item; endpoints that are image names are loaded.
    `imageless_dataloaders` yields {'image_filename': [...], 'path_to_image': [...]}; a "path" may also be an in-memory
    tensor [3,H,W] (synthetic stores) or a callable returning one."""

    def __init__(self, graph, labelmap, has_negative=False, neg_to_pos_ratio=1, imageless_dataloaders=None,
                 transform=None, half_half=False):
        self.G = graph
        self.num_edges = self.G.size()
        self.has_negative = has_negative
        self.neg_to_pos_ratio = neg_to_pos_ratio
        self.half_half = half_half
        if self.half_half:
            self.edge_list_complete_ll = [e for e in self.G.edges() if type(e[0]) != str and type(e[1]) != str]
            self.edge_list_complete_li = [e for e in self.G.edges() if type(e[0]) == str or type(e[1]) == str]
            self.edge_list_ll = self.edge_list_complete_ll
            self.edge_list_li = self.edge_list_complete_li
        else:
            self.edge_list_complete = [e for e in self.G.edges()]
            self.edge_list = self.edge_list_complete
        self.labelmap = labelmap
        self.image_to_loc = {}
        self.transform = transform
        self.load_images = False
        if imageless_dataloaders:
            self.load_images = True
            for batch in imageless_dataloaders:
                for fname, loc in zip(batch['image_filename'], batch['path_to_image']):
                    self.image_to_loc[fname] = loc
        self.input_size = 224
        self.levels_to_hide = []

    def _hidden(self, u, v):
        for lvl in self.levels_to_hide:
            s, e = self.labelmap.level_start[lvl], self.labelmap.level_stop[lvl]
            if (type(u) != str and s <= u < e) or (type(v) != str and s <= v < e):
                return True
        return False

    def set_levels_to_hide(self, list_of_levels):
        self.levels_to_hide = list_of_levels
        if self.half_half:
            self.edge_list_ll = [e for e in self.edge_list_complete_ll if not self._hidden(*e)]
            self.edge_list_li = [e for e in self.edge_list_complete_li if not self._hidden(*e)]
        else:
            self.edge_list = [e for e in self.edge_list_complete if not self._hidden(*e)]

    # An image store (image_store.ImageStore, set by the trainer) turns file-backed items into ImageRef handles: the float tensor is built on
    # the GPU from the store's uint8 copy (bit-identical to this class's own tensor path, tests/test_image_store_gpu.py).
    store_view = None

    def _load(self, filename, train_transform):
        loc = self.image_to_loc[filename]
        sv = self.store_view
        if (sv is not None and isinstance(loc, (str, os.PathLike)) and filename in sv.index_of
                and (not (train_transform and self.transform) or hasattr(self.transform, 'decide'))):
            flip = bool(train_transform and self.transform and self.transform.decide())
            pixels = None
            if not sv.is_resident(filename) and torch.utils.data.get_worker_info() is not None:
                pixels = torch.from_numpy(decode_u8(loc, self.input_size))    # a DataLoader worker decodes what it was going to decode anyway
            return ImageRef(filename, flip, pixels)
        if torch.is_tensor(loc):
            img = loc
        elif callable(loc):
            img = loc()
        else:
            # decode_u8: the reference's decode + ToPILImage + Resize (oe_h.py:668-677, 700-712, 1463-1471; B, G, R channel order); ToTensor:
            img = torch.from_numpy(decode_u8(loc, self.input_size)).permute(2, 0, 1).float().div_(255.0)
        if train_transform and self.transform:
            img = self.transform(img)
        return img

    def get_image(self, filename):
        """oe_h.py:668-677: the val/test transform (no flip) -- used for images drawn as negatives.  Always the float tensor, like the
        reference's (the criterion itself goes through the image store when there is one)."""
        sv, self.store_view = self.store_view, None
        try:
            return self._load(filename, False)
        finally:
            self.store_view = sv

    @staticmethod
    def map_ranges(input, output_range, input_range):
        return round(input * output_range / input_range)

    def edge_of(self, item):
        """The positive edge (u, v) item `item` stands for (oe_h.py:690-703): with half_half, even items walk the label-label edges
        and odd items the label-image edges, each list stretched over half the dataset's length by map_ranges."""
        if self.half_half:
            if item % 2 == 0 and len(self.edge_list_ll) != 0:
                item_ix = self.map_ranges(item // 2, len(self.edge_list_ll) - 1, round(self.__len__() / 2))
                return self.edge_list_ll[item_ix]
            item_ix = self.map_ranges(item // 2, len(self.edge_list_li) - 1, self.__len__() // 2)
            return self.edge_list_li[item_ix]
        return self.edge_list[item]

    def __getitem__(self, item):
        u, v = self.edge_of(item)
        original_from, original_to = u, v
        if self.load_images:
            if type(u) == str:
                u = self._load(u, True)
            if type(v) == str:
                v = self._load(v, True)
        return {'from': u, 'to': v, 'status': 1, 'original_from': original_from, 'original_to': original_to}

    def __len__(self):
        if self.half_half:
            return max(2 * len(self.edge_list_ll), 2 * len(self.edge_list_li))
        return len(self.edge_list)


class RandomHorizontalFlip:
    """transforms.RandomHorizontalFlip of the train transform (oe_h.py:1465) on a [3, H, W] tensor; `decide()` is the same draw without the
    image, for items that travel as ImageRef handles (the mirror then happens in lec_image_gather_u8)."""

    def __init__(self, p=0.5):
        self.p = p

    def decide(self):
        return bool(torch.rand(()) < self.p)

    def __call__(self, img):
        return img.flip(-1) if self.decide() else img


class EmbeddingMetrics:
    """oe_h.py:447-503.  The 'val' threshold sweep (a 512-process pool in the reference, one pass per candidate
    threshold) is a sort + prefix sums here: identical F1-optimal threshold, O(n log n)."""

    def __init__(self, e_for_u_v_positive, e_for_u_v_negative, threshold, phase, n_proc=4):
        self.e_for_u_v_positive = e_for_u_v_positive.reshape(-1)
        self.e_for_u_v_negative = e_for_u_v_negative.reshape(-1)
        self.threshold = threshold
        self.phase = phase

    def calculate_best(self, threshold):
        p, n = self.e_for_u_v_positive, self.e_for_u_v_negative
        cp = int((p <= threshold).sum()); cn = int((n > threshold).sum())
        acc = (cp + cn) / (p.shape[0] + n.shape[0])
        prec = cp / (cp + (n.shape[0] - cn)) if (cp + (n.shape[0] - cn)) else 0.0
        rec = cp / p.shape[0]
        f1 = 0.0 if prec + rec == 0 else 2 * prec * rec / (prec + rec)
        return f1, threshold, acc, prec, rec, cp / p.shape[0], cn / n.shape[0]

    def calculate_metrics(self):
        p = self.e_for_u_v_positive.detach().double().cpu().numpy(); n = self.e_for_u_v_negative.detach().double().cpu().numpy()
        if self.phase != 'val':
            return self.calculate_best(self.threshold)
        # candidate thresholds: the distinct energies, NaN (an energy whose apex row is zero) last as np.unique puts it; a NaN energy is
        # neither "<= t" nor "> t" for any t, and a NaN threshold classifies nothing as anything (torch comparisons, oe_h.py:456-457)
        th = np.unique(np.concatenate((p, n)))
        ps, ns = np.sort(p[~np.isnan(p)]), np.sort(n[~np.isnan(n)])
        live = ~np.isnan(th)
        cp = np.where(live, np.searchsorted(ps, th, side='right'), 0)                       # positives <= t
        cn = np.where(live, len(ns) - np.searchsorted(ns, th, side='right'), 0)             # negatives  > t
        ps, ns = p, n                                                   # totals count every pair, NaN or not
        fp = len(ns) - cn
        with np.errstate(invalid='ignore', divide='ignore'):
            prec = np.where(cp + fp > 0, cp / np.maximum(cp + fp, 1), 0.0)
            rec = cp / len(ps)
            f1 = np.where(prec + rec > 0, 2 * prec * rec / np.maximum(prec + rec, 1e-300), 0.0)
        b = int(np.argmax(f1))                                          # first maximum, like np.argmax(F[:, 0])
        return np.array([f1[b], th[b], (cp[b] + cn[b]) / (len(ps) + len(ns)), prec[b], rec[b], cp[b] / len(ps), cn[b] / len(ns)])


class _NullWriter:
    def add_scalar(self, *a, **k): pass
    def close(self): pass


class GlobalBatchSampler(torch.utils.data.Sampler):
    """Deterministic global batches, sliced per rank: every rank sees the same permutation; rank r loads
    only positions [r*bs, (r+1)*bs) of each global batch.  `global_batches()` replays the index lists so that the
    negative stream can be drawn for the GLOBAL batch in the reference's order on every rank."""

    def __init__(self, n, batch_size, shuffle=True, seed=0, rank=0, world=1):
        self.n, self.bs, self.shuffle, self.seed, self.rank, self.world = n, batch_size, shuffle, seed, rank, world
        self.epoch = 0

    def set_epoch(self, e):
        self.epoch = e

    def global_batches(self):
        if self.shuffle:
            g = torch.Generator(); g.manual_seed(self.seed + self.epoch)
            perm = torch.randperm(self.n, generator=g).tolist()
        else:
            perm = list(range(self.n))
        gb = self.bs * self.world
        return [perm[i:i + gb] for i in range(0, len(perm) - (len(perm) % self.world if self.world > 1 else 0), gb)]

    def __iter__(self):
        for b in self.global_batches():
            per = len(b) // self.world
            yield b[self.rank * per:(self.rank + 1) * per]

    def __len__(self):
        return len(self.global_batches())


# ------------------------------------------------------------------------------------------------ the trainer
class JointEmbeddings:
    """oe_h.py:1318-2247.  `oe.JointEmbeddings` (the Euclidean-cone sibling, oe.py:1224-1991) subclasses this with its
    own model classes and a plain Adam table step."""
    riemannian_table_step = True             # oe_h.py:1768-1771: lambda-rescale before and clip after Adam

    @staticmethod
    def _model_classes():
        from .oe_h import Embedder, FeatCNN18, FeatCNN, FeatNet
        return Embedder, FeatCNN18, FeatCNN, FeatNet

    def __init__(self, graph_dict, imageless_dataloaders, image_dir, use_CNN, labelmap, criterion, lr, n_workers,
                 batch_size, experiment_name, embedding_dim, neg_to_pos_ratio, image_fc7, normalize, alpha,
                 lr_step=[], experiment_dir='../exp/', n_epochs=10, eval_interval=2, feature_extracting=True,
                 use_pretrained=True, load_wt=False, model_name=None, optimizer_method='adam', use_grayscale=False,
                 load_emb_from=None, load_cosine_emb=None, hide_levels=None, half_half=False,
                 compute_dtype=torch.float32, cnn_weights=None, writer=None, fast_path=True, cnn_passes=None,
                 image_store=True, image_store_gb=None, img_feat_net=None, reference_exact_batches=False):
        Embedder, FeatCNN18, FeatCNN, FeatNet = self._model_classes()
        from .resnet import WgradOverlap
        torch.manual_seed(0)                                               # oe_h.py:1338
        self.classes = labelmap.classes; self.n_classes = labelmap.n_classes
        self.levels = labelmap.levels; self.n_levels = len(self.levels); self.level_names = labelmap.level_names
        self.lr = lr; self.lr_step = lr_step; self.batch_size = batch_size
        self.feature_extracting = feature_extracting; self.optimizer_method = optimizer_method
        self.labelmap = labelmap; self.model_name = model_name; self.n_workers = n_workers
        self.imageless_dataloaders = imageless_dataloaders; self.use_CNN = use_CNN; self.image_dir = image_dir
        self.hide_levels = hide_levels; self.half_half = half_half
        self.use_rsgd = False                                              # oe_h.py:1359
        self.lr_labels = self.lr; self.lr_images = 1e-3
        self.best_model_wts = None; self.best_score = 0.0
        self.epoch = 0; self.exp_dir = experiment_dir; self.load_wt = load_wt
        self.criterion = criterion
        # image_store: keep the resized uint8 copy of every image FILE the datasets name in HBM (image_store.ImageStore; image_store_gb
        # caps it, default: all of them) and build the step's float batch on the GPU; False: the reference's host tensors
        self.use_image_store = bool(image_store); self.image_store_gb = image_store_gb; self.image_store = None
        if not torch.cuda.is_available():
            raise RuntimeError('JointEmbeddings runs on the MI355X only (no CPU fallback)')
        self.rank, self.local_rank, self.world = parallel.init_process_group()
        self.device = torch.device('cuda', self.local_rank % torch.cuda.device_count() if self.world > 1 else torch.cuda.current_device())
        torch.cuda.set_device(self.device)
        print('Using device: {}'.format(self.device))
        self.n_epochs = n_epochs; self.eval_interval = eval_interval
        self.log_dir = os.path.join(self.exp_dir, '{}').format(experiment_name)
        self.path_to_save_model = os.path.join(self.log_dir, 'weights')
        self.make_dir_if_non_existent(self.path_to_save_model)
        self.writer = writer if writer is not None else _NullWriter()
        self.graph_dict = graph_dict
        self.optimal_threshold = 0; self.alpha = alpha
        self.embedding_dim = embedding_dim; self.neg_to_pos_ratio = neg_to_pos_ratio; self.normalize = None
        is_hyp = getattr(criterion, 'K', None) is not None              # cone criteria carry K; order embeddings do not
        self.model = Embedder(embedding_dim=self.embedding_dim, labelmap=labelmap, normalize=self.normalize,
                              K=criterion.K if is_hyp else None)
        self.model.to(self.device)
        if img_feat_net is not None:
            self.img_feat_net = img_feat_net.to(self.device)     # the caller's own image network (an oe_h.FeatCNN18 / FeatNet-like module)
        elif self.use_CNN:
            cls = FeatCNN if (model_name or '').lower() == 'resnet50' else FeatCNN18      # reference hard-wires resnet18 (:1404)
            self.img_feat_net = cls(image_dir=self.image_dir, output_dim=self.embedding_dim,
                                    K=criterion.K if is_hyp else None, weights=cnn_weights,
                                    compute_dtype=compute_dtype).to(self.device)
        else:
            self.img_feat_net = FeatNet(output_dim=self.embedding_dim, normalize=self.normalize,
                                        K=criterion.K if is_hyp else None).to(self.device)
        self.criterion.set_negative_graph(self.graph_dict['G_train_neg'], self.graph_dict['mapping_node_to_ix'],
                                          self.graph_dict['mapping_ix_to_node'])       # oe_h.py:1420
        self.create_splits()
        self.prepare_model()
        # MI355X: CNN parameters + grads in one flat arena; gradient reducer with backward overlap
        self.arena = parallel.FlatArena(self.img_feat_net.parameters(), self.device)
        w = self.model.embeddings.weight
        self.table_grad = torch.zeros_like(w.data); w.grad = self.table_grad
        self.table_m = torch.zeros_like(w.data); self.table_v = torch.zeros_like(w.data); self.table_step = 0
        self.reducer = parallel.GradientReducer(self.arena, extra=[self.table_grad])
        if self.world > 1:                                                  # replicas start identical
            torch.distributed.broadcast(self.arena.data, 0); torch.distributed.broadcast(w.data, 0)
        # The measured path of engine.StepEngine, behind this trainer too: liblecone's convolutions (fp32: the f32-MFMA implicit
        # GEMMs; bf16: the MFMA kernels + low-precision shadow weights written by the Adam kernel), BatchNorm statistics / apply /
        # backward fusions, weight gradients accumulated straight into the flat arena on a second HIP stream.
        self.overlap = None
        if self.use_CNN and fast_path and compute_dtype in (torch.float32, torch.bfloat16):
            if compute_dtype == torch.bfloat16:
                self.arena.enable_lowp_transposed()     # bf16 shadow + its transposed twin (the data gradients' operand)
            # fp32: the CNN batch of a step goes through the backbone as two concurrent halves, one HIP stream each -- one half's HBM-bound
            # BatchNorm passes under the other's matrix-bound convolutions (engine.StepEngine) -- with the weight
            # gradients in line; bf16: one pass, weight gradients on a side stream by default -- `cnn_passes=2` runs the bf16 backbone as two concurrent passes too
            # (round 6: 4 % faster on the engine's step with the own bf16 convolution family).  cnn_passes overrides.
            if cnn_passes is None:
                cnn_passes = 2 if compute_dtype == torch.float32 else 1
            self.cnn_passes = int(cnn_passes)
            self.overlap = WgradOverlap(self.reducer, self.arena, side_stream=self.cnn_passes == 1)
            if self.cnn_passes > 1:
                self.img_feat_net.cnn_passes = self.cnn_passes
        # reference_exact_batches: the reference's own CNN batches -- up to four forwards per step, every fixed image end embedded K more times,
        # each its own BatchNorm batch (criterion._forward_reference_batches; oe_h.py:929-967, 980-985, 1003-1009) -- instead of one
        # forward per distinct image.  (1 + K) x the CNN rows; for parity runs against the reference (fixture F13).
        self.reference_exact_batches = bool(reference_exact_batches)
        self.criterion.reference_exact_batches = self.reference_exact_batches
        if self.reference_exact_batches and self.use_CNN:
            self.cnn_passes = 1; self.img_feat_net.cnn_passes = 1     # every forward of the step is ONE BatchNorm batch
        if self.use_CNN and hasattr(self.img_feat_net, 'model'):
            # this trainer's settings travel with ITS backbone (resnet.ResNet.wgrad_overlap / bn_grad_accumulate / conv_schedule -> the
            # FusionContext of each forward): nothing process-wide is switched around a step, a second trainer in the process keeps its own
            bb = self.img_feat_net.model
            bb.wgrad_overlap = self.overlap if self.overlap is not None else False
            if getattr(self, 'cnn_passes', 1) > 1:
                # two backward passes add into the same gradient slots from concurrent streams: BatchNorm's d gamma / d beta accumulate with
                # atomics like the weight gradients; concurrent passes fill each other's tails: tile walk (engine._core_passes)
                bb.bn_grad_accumulate = True
                bb.conv_schedule = _lib.SCHEDULE_TILE_WALK
            if self.reference_exact_batches:
                bb.bn_grad_accumulate = True                        # several forwards / backwards of the backbone per step share the slots
        self.check_graph_embedding_neg_graph = None
        self.check_reconstr_every = 1; self.save_model_every = 1
        self.reconstruction_f1 = self.reconstruction_threshold = self.reconstruction_accuracy = 0.0
        self.reconstruction_prec = self.reconstruction_recall = 0.0
        self.levels_to_hide_for_epoch = {}

    @staticmethod
    def make_dir_if_non_existent(directory):
        if not os.path.exists(directory):
            os.makedirs(directory, exist_ok=True)

    def prepare_model(self):
        self.params_to_update = [{'params': self.model.parameters(), 'lr': self.lr_labels},
                                 {'params': self.img_feat_net.parameters(), 'lr': self.lr_images}]

    def create_splits(self):
        random.seed(0)                                                      # oe_h.py:1472
        self.criterion.seed_sampler(0)

        flip = RandomHorizontalFlip(0.5)                                    # train only (:1465)
        il = self.imageless_dataloaders
        mk = lambda key, tr, hh: ETHECHierarchyWithImages(self.graph_dict[key], labelmap=self.labelmap,
                                                          imageless_dataloaders=il[tr] if (self.use_CNN and il) else None,
                                                          transform=flip if tr == 'train' else None, half_half=hh)
        train_set = mk('G_train_tc', 'train', self.half_half)
        val_set = mk('G_val', 'val', False); test_set = mk('G_test', 'test', False)
        self.train_set = train_set
        self.datasets = {'train': train_set, 'val': val_set, 'test': test_set}
        self.dataset_length = {k: len(v) for k, v in self.datasets.items()}
        locs = {}
        for ds in (train_set, val_set, test_set):
            for nm, loc in ds.image_to_loc.items():
                if isinstance(loc, (str, os.PathLike)):
                    locs.setdefault(nm, loc)
        if self.use_CNN and self.use_image_store and locs:
            per = train_set.input_size * train_set.input_size * 3
            cap = None if self.image_store_gb is None else max(4 * self.batch_size * (1 + 2 * self.neg_to_pos_ratio), int(self.image_store_gb * 1e9) // per)
            self.image_store = ImageStore(locs, self.device, hw=train_set.input_size, capacity=cap, decode_threads=max(4, self.n_workers))
            for ds in (train_set, val_set, test_set):
                ds.store_view = self.image_store.view()
            print('Image store: %d image files, %d slots of %d bytes in HBM' % (len(locs), self.image_store.capacity, per))
        self.criterion.image_store = self.image_store
        self._make_train_loader()
        from .oe_h import my_collate
        self.dataloaders['val'] = torch.utils.data.DataLoader(val_set, batch_size=self.batch_size, collate_fn=my_collate,
                                                              num_workers=self.n_workers, shuffle=False)
        self.dataloaders['test'] = torch.utils.data.DataLoader(test_set, batch_size=self.batch_size, collate_fn=my_collate,
                                                               num_workers=self.n_workers, shuffle=False)

    # Who decodes image files.  With the HBM image store (the default for file-backed datasets) the train loader runs WITHOUT worker processes:
    # its items are ImageRef handles, and every file a step needs -- positives and negatives alike -- is requested from the store's decode
    # THREADS one step ahead by train_epoch's lookahead (`n_workers` sizes that pool; PIL releases the GIL while it decodes and resizes).
    # Forking DataLoader workers from the training process is what the reference does, and on this stack it is a hazard: the process holds a
    # live HIP context and a dozen threads (autograd, RCCL / gloo, decode pool, lookahead).  Measured on the MI355X box
    # (tools/probe_trainer_files.py, tools/dp8_trainer_files.py): every fork of 8 workers stalls the GPU work of the parent ONCE for
    # 2.7 s (a loader that forks per epoch pays it every epoch: 233 ms per step instead of 108 over 16-step epochs), and under
    # torch.distributed.run with gloo threads alive the forked workers never delivered a batch at all (the classic fork-with-threads
    # deadlock).  Without the store (image_store=False, or in-memory tensors) the loader keeps the reference's `n_workers` processes,
    # started once and kept (persistent_workers) so that the stall is paid once.
    persistent_workers = True
    worker_context = None                    # multiprocessing context of those workers (None: the platform default, fork)

    def _make_train_loader(self):
        from .oe_h import my_collate
        self.train_sampler = GlobalBatchSampler(len(self.train_set), self.batch_size, shuffle=True, seed=0,
                                                rank=getattr(self, 'rank', 0), world=getattr(self, 'world', 1))
        self.dataloaders = getattr(self, 'dataloaders', {})
        nw = 0 if getattr(self, 'image_store', None) is not None else self.n_workers
        self.dataloaders['train'] = torch.utils.data.DataLoader(self.train_set, batch_sampler=self.train_sampler,
                                                                num_workers=nw, collate_fn=my_collate,
                                                                persistent_workers=bool(self.persistent_workers) and nw > 0,
                                                                multiprocessing_context=(self.worker_context if nw > 0 else None))
        self.datasets['train'] = self.train_set
        self.dataset_length['train'] = len(self.train_set)

    def _set_hidden_levels(self, levels):
        print('Set levels to hide to: {}'.format(levels))
        self.train_set.set_levels_to_hide(levels)
        self.criterion.set_levels_to_hide(levels)
        self._make_train_loader()

    # ---- Riemannian helpers (oe_h.py:1604-1644), kept as methods; the step itself uses the fused kernels ---------
    def lambda_x(self, x):
        return 2. / (1 - torch.norm(x, p=2, dim=1, keepdim=True).repeat(1, self.embedding_dim))

    # ---- training loop ---------------------------------------------------------------------------------------------
    def run_model(self, optimizer=None):
        self.levels_to_hide_for_epoch = {}
        if self.hide_levels:
            self.levels_to_hide_for_epoch = {0: [1, 2, 3], 20: [2, 3], 50: [3], 100: []}       # oe_h.py:1536
        current = None
        for key in self.levels_to_hide_for_epoch:
            if self.epoch >= key:
                current = key
        if current is not None:
            self._set_hidden_levels(self.levels_to_hide_for_epoch[current])
        if self.load_wt:
            self.find_existing_weights()
        self.best_model_wts = copy.deepcopy(self.model.state_dict()); self.best_score = 0.0
        since = time.time()
        for self.epoch in range(self.epoch, self.n_epochs):
            print('=' * 10); print('Epoch {}/{}'.format(self.epoch, self.n_epochs - 1)); print('=' * 10)
            if self.epoch in self.levels_to_hide_for_epoch:
                self._set_hidden_levels(self.levels_to_hide_for_epoch[self.epoch])
            t0 = time.time()
            self.pass_samples(phase='train')
            self.writer.add_scalar('epoch_time_train', time.time() - t0, self.epoch)
            if self.epoch % self.eval_interval == 0:
                t1 = time.time(); self.pass_samples(phase='val')
                self.writer.add_scalar('epoch_time_val', time.time() - t1, self.epoch)
                t2 = time.time(); self.pass_samples(phase='test')
                self.writer.add_scalar('epoch_time_test', time.time() - t2, self.epoch)
            self._lr_scale = 0.1 ** sum(1 for m in self.lr_step if self.epoch + 1 >= m)       # MultiStepLR(gamma=0.1)
            self.writer.add_scalar('epoch_time', time.time() - t0, self.epoch)
        print('Training complete in {:.0f}s'.format(time.time() - since))
        print('Best val score: {:4f}'.format(self.best_score))
        self.model.load_state_dict(self.best_model_wts)
        self.writer.close()
        return self.model

    def train(self):
        self.run_model(None)
        self.load_best_model()

    def train_step(self, data_item):
        """oe_h.py:1734-1774 for one batch.  Returns the (device) loss; nothing here synchronises with the host.
        Exactly one forward / backward of the image network per step: the weight-gradient kernels ADD into the arena's gradient
        slots (zeroed here), BatchNorm gradients are written in place."""
        ov = self.overlap
        # The host may run at most two steps ahead of the GPU.  Nothing below synchronises, and activations the side stream has
        # touched go back to the allocator only when its events have passed: a caller that never reads the loss (a timing loop)
        # would otherwise pile up one step's activations per step of run-ahead (measured: 234 GB live after 11 fp32 steps).
        pend = self.__dict__.setdefault('_steps_in_flight', [])
        if len(pend) >= 2:
            t_w = time.perf_counter()
            pend.pop(0).synchronize()
            self.host_wait_s = getattr(self, 'host_wait_s', 0.0) + time.perf_counter() - t_w      # (bench / tools: host time of a step = wall - waits)
        multi = getattr(self, 'cnn_passes', 1) > 1 or self.reference_exact_batches     # several backward passes over the parameters per step
        live = self.reducer.live
        if multi:
            # every parameter reports once per pass: the reducer's per-parameter hooks stay muted, the buckets are reduced once, below
            self.reducer.live = False
        try:
            self.arena.zero_grad(); self.table_grad.zero_()
            loss, e_pos, e_neg = self.criterion(self.model, self.img_feat_net, data_item['from'], data_item['to'],
                                                data_item['original_from'], data_item['original_to'], data_item['status'], 'train')
            loss.backward()                                                 # oe_h.py:1766
            if ov is not None:
                ov.join()                                                   # weight gradients from the side stream
            if hasattr(self.img_feat_net, 'join_passes'):
                self.img_feat_net.join_passes()                             # ... and the concurrent passes' backward from their streams
        finally:
            if multi:
                self.reducer.live = live; self.reducer.reset()
        t_w = time.perf_counter()
        self.reducer.finish()
        self.host_wait_s = getattr(self, 'host_wait_s', 0.0) + time.perf_counter() - t_w
        self.apply_updates()
        if loss.is_cuda:
            done = torch.cuda.Event(); done.record(); pend.append(done)
        return loss.detach(), e_pos, e_neg

    def apply_updates(self):
        lr = self.lr_labels * getattr(self, '_lr_scale', 1.0)
        w = self.model.embeddings.weight
        Kc = getattr(self.criterion, 'K', None)
        self.table_step += 1
        if self.use_rsgd:                                                   # oe_h.py:1757-1764
            ops.table_step_rsgd(w.data, self.table_grad, lr, Kc)
            self.arena.adam_step(self.lr_images * getattr(self, '_lr_scale', 1.0))
        else:                                                               # :1766-1771, one Adam over table + CNN at lr
            ops.table_step_adam(w.data, self.table_grad, self.table_m, self.table_v, self.table_step, lr,
                                Kc or 0.0, riemannian=bool(Kc) and self.riemannian_table_step,
                                clip=bool(Kc) and self.riemannian_table_step)
            self.arena.adam_step(lr)

    negative_lookahead = True                # draw step t+1's negatives (and start decoding their images) while step t runs

    def train_epoch(self, max_steps=None, on_step=None):
        """The train branch's loop over the DataLoader (oe_h.py:1734-1774), one `train_step` per batch.  Returns (summed loss on the device,
        steps run).  The batches of the epoch are known up front (GlobalBatchSampler's permutation), so a host thread walks them one step
        AHEAD of the GPU: it draws each batch's negatives -- in batch order, on the GLOBAL batch when data parallel: the sampler's MT19937
        stream is consumed exactly as the reference's training thread consumes it (oe_h.py:940-957) whatever the world size -- and asks
        the image store to decode the negative images it does not hold yet, off the training thread (the reference decodes them
        synchronously in it: oe_h.py:980-983, 1003-1007)."""
        self.criterion.set_dataloader(self.datasets['train'])
        self.model.train(); self.img_feat_net.train()
        running = torch.zeros((), device=self.device)
        self.train_sampler.set_epoch(self.epoch)
        global_batches = self.train_sampler.global_batches()
        if max_steps is not None:
            global_batches = global_batches[:max_steps]
        n2i = self.graph_dict['mapping_node_to_ix']; i2n = self.graph_dict['mapping_ix_to_node']
        N = self.n_classes
        store = self.image_store

        def shard_of(s):
            if s >= len(global_batches):
                return None
            gb = global_batches[s]
            edges = [self.train_set.edge_of(i_) for i_ in gb]         # (half_half: the same item -> edge map __getitem__ uses)
            g_from = np.fromiter((n2i[u] for u, _ in edges), dtype=np.int32, count=len(edges))
            g_to = np.fromiter((n2i[v] for _, v in edges), dtype=np.int32, count=len(edges))
            return g_from, g_to

        def on_item(item):
            if store is not None:                                       # every image file of this rank's shard of the step, positives and negatives
                ixs = np.unique(np.concatenate([np.asarray(x).reshape(-1) for x in item]))
                names = [i2n[ix] for ix in ixs[ixs >= N].tolist()]
                store.request([nm for nm in names if store.holds(nm)])

        look = None
        if self.negative_lookahead:
            look = parallel.NegativePrefetcher(self.criterion.negative_G, shard_of, self.neg_to_pos_ratio, mode='replicated', depth=2,
                                               on_item=on_item, rank=self.rank, world=self.world)
        steps = 0
        try:
            for index, data_item in enumerate(self.dataloaders['train']):
                if max_steps is not None and index >= max_steps:
                    break
                if look is not None:
                    self.criterion.predrawn = look.next()
                elif self.world > 1:                                    # this rank's slice of the global batch
                    g_from, g_to = shard_of(index)
                    per = len(g_from) // self.world
                    self.criterion.dp_global = (g_from, g_to, self.rank * per, (self.rank + 1) * per)
                loss, _, _ = self.train_step(data_item)
                running += loss
                steps += 1
                if on_step is not None:
                    on_step(steps)
        finally:
            self.criterion.dp_global = None; self.criterion.predrawn = None
            if look is not None:
                look.close()
        if self.world > 1:
            torch.distributed.all_reduce(running)
        return running, steps

    def pass_samples(self, phase, save_to_tensorboard=True):
        self.criterion.set_dataloader(self.datasets[phase])
        if phase == 'train':
            self.model.train(); self.img_feat_net.train()
            running, index = self.train_epoch()
            index -= 1
            classification_metrics = self.calculate_classification_metrics(phase)
            epoch_loss = running.item() / max(1, (index + 1) * self.batch_size * self.world * self.neg_to_pos_ratio * 2)   # :1780
            if save_to_tensorboard:
                self.writer.add_scalar('{}_loss'.format(phase), epoch_loss, self.epoch)
            print('train loss: {}'.format(epoch_loss))
            self.last_epoch_loss = epoch_loss
        else:
            self.model.eval(); self.img_feat_net.eval()
            classification_metrics = self.calculate_classification_metrics(phase)
            if phase == 'test' and self.epoch % self.save_model_every == 0:
                self.save_model(-9999.0)
            if phase == 'val' and classification_metrics['m-f1'] >= self.best_score:
                self.best_score = classification_metrics['m-f1']
                self.best_model_wts = copy.deepcopy(self.model.state_dict())
                self.save_model(-9999.0, filename='best_model')
            if phase == 'test' and (self.epoch % self.check_reconstr_every == 0 or not save_to_tensorboard):
                (self.reconstruction_f1, self.reconstruction_threshold, self.reconstruction_accuracy,
                 self.reconstruction_prec, self.reconstruction_recall, c_pos, c_neg) = self.check_graph_embedding()
        self.last_metrics = classification_metrics
        return classification_metrics

    # ---- checkpoints (oe_h.py:1876-1957; same file names and dict keys, `module.` prefix kept for interchange) ----
    def _state(self, module):
        return {'module.' + k: v for k, v in module.state_dict().items()}

    def save_model(self, loss, filename=None):
        if self.rank != 0:
            return
        rec = {'f1': self.reconstruction_f1, 'precision': self.reconstruction_prec, 'recall': self.reconstruction_recall,
               'accuracy': self.reconstruction_accuracy, 'threshold': self.reconstruction_threshold}
        tag = filename if filename else self.epoch
        opt = self._optimizer_state_dict()
        torch.save({'epoch': self.epoch, 'model_state_dict': self._state(self.model), 'optimizer_state_dict': opt,
                    'loss': loss, 'optimal_threshold': self.optimal_threshold, 'reconstruction_scores': rec},
                   os.path.join(self.path_to_save_model, '{}_model.pth'.format(tag)))
        torch.save({'epoch': self.epoch, 'model_state_dict': self._img_state(), 'optimizer_state_dict': self._optimizer_images_state_dict(),
                    'loss': loss, 'optimal_threshold': self.optimal_threshold, 'reconstruction_scores': rec},
                   os.path.join(self.path_to_save_model, '{}_img_feat_net.pth'.format(tag)))

    def _img_state(self):
        """The image network's state dict with the reference's key names: FeatCNN18 keeps its ResNet inside nn.DataParallel
        (oe_h.py:301: `model.module.conv1.weight`, ...), and the feature-input FeatNet is itself wrapped (oe_h.py:1439:
        `module.fc1.weight`)."""
        sd = self.img_feat_net.state_dict()
        if self.use_CNN:
            return {('model.module.' + k[len('model.'):] if k.startswith('model.') else k): v for k, v in sd.items()}
        return {'module.' + k: v for k, v in sd.items()}

    def _optimizer_images_state_dict(self):
        """torch.optim.Adam state-dict layout of the reference's `optimizer_images` (oe_h.py:1521: Adam over the image network's
        parameters at lr_images; it is stepped only under use_rsgd, so its state is empty otherwise).  The reference's load_model
        calls optimizer_images.load_state_dict on this entry (oe_h.py:1956): it must be a valid Adam state dict."""
        n = len(self.arena.params)
        state = self.arena.export_adam_state(first_index=0) if (self.use_rsgd and self.arena.exp_avg is not None) else {}
        group = {'lr': self.lr_images, 'betas': (0.9, 0.999), 'eps': 1e-8, 'weight_decay': 0, 'amsgrad': False,
                 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
                 'params': list(range(n))}
        return {'state': state, 'param_groups': [group]}

    @staticmethod
    def _plain_key(k):
        """Key without the nn.DataParallel segments the reference's wrappers insert (leading `module.` and inner `.module.`)."""
        if k.startswith('module.'):
            k = k[len('module.'):]
        return k.replace('.module.', '.')

    def _load_sd(self, module, sd):
        sd = {self._plain_key(k): v for k, v in sd.items()}
        with torch.no_grad():
            own = module.state_dict()
            for k, v in sd.items():
                own[k].copy_(v)                                            # in place: parameters stay inside the arena
        if getattr(self, 'arena', None) is not None:
            self.arena.refresh_lowp()                                      # the bf16 shadow (if any) follows the loaded weights

    def load_model(self, epoch_to_load):
        ck = torch.load(os.path.join(self.path_to_save_model, '{}_model.pth'.format(epoch_to_load)), map_location=self.device)
        self._load_sd(self.model, ck['model_state_dict'])
        self.epoch = ck['epoch']; self.optimal_threshold = ck['optimal_threshold']
        self._load_optimizer_state_dict(ck.get('optimizer_state_dict') or {})
        ck = torch.load(os.path.join(self.path_to_save_model, '{}_img_feat_net.pth'.format(epoch_to_load)), map_location=self.device)
        self._load_sd(self.img_feat_net, ck['model_state_dict'])

    def _optimizer_state_dict(self):
        """torch.optim.Adam state-dict layout of the reference's `optimizer_labels` (oe_h.py:1523: ONE group holding the
        label table followed by the CNN's parameters), so that checkpoints move between the two code bases."""
        state = {0: {'step': torch.tensor(float(self.table_step)), 'exp_avg': self.table_m.clone(), 'exp_avg_sq': self.table_v.clone()}}
        state.update(self.arena.export_adam_state(first_index=1))
        n = 1 + len(self.arena.params)
        group = {'lr': self.lr_labels, 'betas': (0.9, 0.999), 'eps': 1e-8, 'weight_decay': 0, 'amsgrad': False,
                 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
                 'params': list(range(n))}
        return {'state': state, 'param_groups': [group]}

    def _load_optimizer_state_dict(self, sd):
        state = sd.get('state') or {}
        if 0 in state:
            self.table_m.copy_(state[0]['exp_avg']); self.table_v.copy_(state[0]['exp_avg_sq'])
            self.table_step = int(float(state[0]['step']))
        self.arena.import_adam_state(state, first_index=1)

    def load_emb_model(self, path_to_weights):
        """oe_h.py:1904-1916: label embeddings (+ threshold / reconstruction scores) from another run."""
        ck = torch.load(path_to_weights, map_location=self.device)
        self._load_sd(self.model, ck['model_state_dict'])
        self.optimal_threshold = ck['optimal_threshold']
        if 'reconstruction_scores' in ck:
            r = ck['reconstruction_scores']
            (self.reconstruction_f1, self.reconstruction_threshold, self.reconstruction_accuracy, self.reconstruction_prec,
             self.reconstruction_recall) = r['f1'], r['threshold'], r['accuracy'], r['precision'], r['recall']

    def find_existing_weights(self):
        weights = sorted([f.split('_')[0] for f in os.listdir(self.path_to_save_model)])
        weights = [w for w in weights if w.isdigit()]
        weights.sort(key=int)
        if len(weights) < 1:
            print('Could not find weights to load from, will train from scratch.')
        else:
            self.load_model(epoch_to_load=weights[-1])

    def load_best_model(self):
        self.load_model(epoch_to_load='best_model')
        return self.pass_samples(phase='test', save_to_tensorboard=False)

    # ---- metrics -----------------------------------------------------------------------
    # reference_exact_eval = True (default): the evaluation phase gives the reference's numbers on the same inputs (fixture F12), which
    # includes what its chunk loops do -- `rows[ix:min(ix + bs, len - 1)]` never reaches the LAST image and the LAST label, whose rows
    # stay zero (oe_h.py:1997-2011, 2230-2234); a zero label row has NaN energies, which rank last / fail every threshold test.
    # False: every row is embedded (the corrected variant), in large chunks, with the image network in eval mode.
    reference_exact_eval = True

    @torch.no_grad()
    def embed_images(self, names, bs=256, skip_last=False):
        """Rows of the image network's output for `names`, `bs` images per forward.  skip_last: the reference's chunk rule
        (oe_h.py:1997-2003): every chunk ends at min(ix + bs, len - 1), so the last image is never embedded and its row stays zero
        (an EMPTY last chunk -- len = 1 mod bs -- makes the reference's torch.stack raise; here it is skipped)."""
        ds = self.criterion.dataloader
        store = self.image_store
        out = torch.zeros((len(names), self.embedding_dim), device=self.device)
        n = len(names)
        for i in range(0, n, bs):
            hi = min(i + bs, n - 1) if skip_last else min(i + bs, n)
            if hi <= i:
                continue
            chunk = names[i:hi]
            if store is not None and all(store.holds(nm) for nm in chunk):
                store.request([nm for nm in names[hi:hi + 2 * bs] if store.holds(nm)])    # the next chunks decode while this one runs
                stack = store.batch(chunk)
            else:
                stack = torch.stack([ds.get_image(nm) for nm in chunk]).to(self.device)
            out[i:hi] = self._embed_forward(stack, full=(hi - i == bs))
        return out

    # The embedding forward of a full chunk is replayed as a hipGraph from its third occurrence on (eval_graphs = False: always eager).  Why: the
    # reference's own chunk size in the 'train' phase is 10 images (oe_h.py:1972), a forward of ~270 launches whose kernels take ~1 ms and whose
    # Python / ctypes enqueue takes 4.7 -- launch-bound; replayed, the same kernels run back to back.  Same kernels, same order, same results: in
    # train mode a replay updates the BatchNorm running statistics exactly as the eager forward (and the reference's) does.  The graph holds raw
    # pointers to the parameters: load_model / optimizer steps write in place and are seen (an eval-mode graph recomputes every BatchNorm's
    # scale / shift vectors from them on each replay: lec_bn_eval_coeffs_f32 is part of the graph); `drop_eval_graphs()` after anything that
    # re-allocates parameters.
    eval_graphs = True

    def drop_eval_graphs(self):
        self._eval_graphs = {}

    def _embed_forward(self, stack, full):
        net = self.img_feat_net
        if not (self.eval_graphs and full and stack.is_cuda and stack.dim() == 4 and getattr(net, 'compute_dtype', None) == torch.float32
                and not torch.is_grad_enabled() and not torch.cuda.is_current_stream_capturing()):
            return net(stack).float()
        cache = self.__dict__.setdefault('_eval_graphs', {})
        key = (tuple(stack.shape), bool(net.training))
        ent = cache.get(key)
        if ent is None:
            ent = cache[key] = {'seen': 0, 'graph': None, 'stream': torch.cuda.Stream(device=stack.device)}
        if ent['graph'] is None:
            ent['seen'] += 1
            cur = torch.cuda.current_stream()
            if ent['seen'] <= 2:
                # eager, on the stream the graph will be captured on (its convolution scratch gets registered: ops._conv_scratch)
                ent['stream'].wait_stream(cur)
                with torch.cuda.stream(ent['stream']):
                    y = net(stack).float()
                stack.record_stream(ent['stream']); y.record_stream(cur)
                cur.wait_stream(ent['stream'])
                return y
            try:
                ent['in'] = torch.empty_like(stack, memory_format=torch.channels_last)
                ent['in'].copy_(stack)
                torch.cuda.synchronize(stack.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=ent['stream']):
                    ent['out'] = net(ent['in']).float()
                ent['graph'] = g
            except Exception as e:                              # a capture that fails costs nothing but the replay: say so and stay eager
                print('embed_images: graph capture failed (%s: %s); eager launches from here on' % (type(e).__name__, e))
                self.eval_graphs = False
                return net(stack).float()
        ent['in'].copy_(stack)
        ent['graph'].replay()
        return ent['out'].clone()

    @torch.no_grad()
    def calculate_classification_metrics(self, phase, k=[1, 3, 5], reference_exact=None):
        """oe_h.py:1971-2178 with the per-image python loop replaced by ONE fused scoring + per-level top-k launch on the GPU
        (lec_level_topk).  Same return dict as the reference (micro / macro metrics, hit@k, `level_metrics`, the two median norms) and
        the same side effects (`self.img_rep`, `self.image_is_a_member_of` for phase 'train').  reference_exact (default: the
        attribute `reference_exact_eval`): see above; pinned by fixture F12, the reference's own output on the same inputs.
        One case is NOT reproduced: a split whose graph misses some label -- the reference indexes its label rows by POSITION in the
        sorted list of the split's labels while slicing levels by label id (oe_h.py:2011-2036), i.e. it assumes every label occurs;
        here rows are always addressed by label id."""
        exact = self.reference_exact_eval if reference_exact is None else bool(reference_exact)
        G = self.graph_dict['G_{}'.format(phase)]
        nodes = list(G)
        images = [n for n in nodes if type(n) == str]
        labels = sorted(n for n in nodes if type(n) != str)
        metrics = {}
        if not images:
            return {'m-f1': 0.0, 'accuracy': 0.0}
        if exact:
            # the reference's chunks of 10 images, in whatever mode the networks are in (oe_h.py:1972, 1778, 1792-1795).  In eval mode
            # (the val / test phases) a row's embedding does not depend on what shares its forward (BatchNorm uses its running
            # statistics), so larger chunks give the same rows; in training mode (the train phase's own call) the chunk IS the
            # BatchNorm batch and stays 10.
            img_rep = self.embed_images(images, bs=10 if self.img_feat_net.training else 250, skip_last=True)
        else:
            was_training = self.img_feat_net.training
            self.img_feat_net.eval()
            img_rep = self.embed_images(images)
            self.img_feat_net.train(was_training)
        label_rep = self.model(torch.arange(self.n_classes, device=self.device)).detach().clone()
        if exact:
            label_rep[labels[-1]] = 0.0                             # the label row the reference's loop never fills
        metrics['median_img_norm'] = torch.median(torch.norm(img_rep, dim=1)).item()
        metrics['median_label_norm'] = torch.median(torch.norm(label_rep[labels], dim=1)).item()
        starts = list(self.labelmap.level_start[:self.n_levels]) + [self.labelmap.level_stop[self.n_levels - 1]]
        kk_all = min(max(k), 8)
        top_idx, _ = ops.level_topk(label_rep, img_rep, starts, kk_all, getattr(self.criterion, 'K', None),
                                    self.criterion.energy)           # [n_img, n_levels, k]: scoring + top-k in one launch
        member = np.zeros((len(images), self.n_levels), dtype=np.int64)
        member_lists = {}
        for i, name in enumerate(images):
            m = sorted(G.predecessors(name))
            member[i, :len(m)] = m[:self.n_levels]
            member_lists[i] = m
        member_t = torch.from_numpy(member).to(self.device)
        N = self.n_classes
        cnt = {n: torch.zeros(N, dtype=torch.int64, device=self.device) for n in ('tp', 'fp', 'fn', 'tn')}
        hit = {kv: torch.zeros(N, dtype=torch.int64, device=self.device) for kv in k}
        for lvl in range(self.n_levels):
            s, e = self.labelmap.level_start[lvl], self.labelmap.level_stop[lvl]
            kk = min(kk_all, e - s)
            idx = top_idx[:, lvl, :kk].long()
            # a slot the kernel left empty (-1): every remaining energy of the level is NaN.  torch.topk ranks NaN last, so the
            # reference's list continues with the NaN label (there is at most one: the zero row)
            idx = torch.where(idx < 0, torch.full_like(idx, labels[-1] if s <= labels[-1] < e else s), idx)
            truth = member_t[:, lvl]
            for kv in k:
                h = (idx[:, :min(kv, kk)] == truth[:, None]).any(dim=1).long()
                hit[kv].index_add_(0, truth, h)
            correct = idx[:, 0] == truth
            cnt['tp'].index_add_(0, truth, correct.long())
            cnt['fp'].index_add_(0, idx[:, 0], (~correct).long())
            cnt['fn'].index_add_(0, truth, (~correct).long())
            cnt['tn'][s:e] += correct.long().sum(); cnt['tn'].index_add_(0, truth, -correct.long())   # tn for every OTHER label of the level
        # counts -> metrics on the host in python floats, in the reference's order of operations (oe_h.py:2064-2160)
        c = {n: t.cpu().numpy() for n, t in cnt.items()}
        hk = {kv: t.cpu().numpy() for kv, t in hit.items()}
        tot = {n: int(c[n][labels].sum()) for n in c}
        f1_of = {}; prec_of = {}; rec_of = {}
        for l in labels:
            tp_, fp_, fn_ = int(c['tp'][l]), int(c['fp'][l]), int(c['fn'][l])
            prec_of[l] = 0.0 if tp_ + fp_ == 0 else tp_ / (tp_ + fp_)
            rec_of[l] = 0.0 if tp_ + fn_ == 0 else tp_ / (tp_ + fn_)
            f1_of[l] = 0.0 if prec_of[l] + rec_of[l] == 0 else (2 * prec_of[l] * rec_of[l]) / (prec_of[l] + rec_of[l])
        prec = tot['tp'] / (tot['tp'] + tot['fp']); rec = tot['tp'] / (tot['tp'] + tot['fn'])
        metrics['accuracy'] = (tot['tp'] + tot['tn']) / (tot['tp'] + tot['tn'] + tot['fp'] + tot['fn'])
        metrics['m-precision'], metrics['m-recall'] = prec, rec
        metrics['m-f1'] = 0.0 if prec + rec == 0 else (2 * prec * rec) / (prec + rec)
        for kv in k:
            metrics['hit@{}'.format(kv)] = int(hk[kv][labels].sum()) / (self.n_levels * len(images))
        metrics['M-precision'] = sum(prec_of[l] for l in labels) / len(labels)
        metrics['M-recall'] = sum(rec_of[l] for l in labels) / len(labels)
        metrics['M-f1'] = sum(f1_of[l] for l in labels) / len(labels)
        metrics['level_metrics'] = {}
        for lvl in range(self.n_levels):                            # oe_h.py:2123-2160 (M-f1 is divided by stop - start + 1 there)
            s, e = self.labelmap.level_start[lvl], self.labelmap.level_stop[lvl]
            tp_, tn_, fp_, fn_ = (int(c[n][s:e].sum()) for n in ('tp', 'tn', 'fp', 'fn'))
            lm_ = {}
            for kv in k:
                lm_['hit@{}'.format(kv)] = int(hk[kv][s:e].sum()) / len(images)
            lp = tp_ / (tp_ + fp_); lr_ = tp_ / (tp_ + fn_)
            lm_['m-precision'], lm_['m-recall'] = lp, lr_
            lm_['m-f1'] = 0.0 if lp + lr_ == 0 else (2 * lp * lr_) / (lp + lr_)
            lm_['M-f1'] = sum(f1_of.get(l, 0.0) for l in range(s, e)) / (e - s + 1)
            lm_['accuracy'] = (tp_ + tn_) / (tp_ + tn_ + fp_ + fn_)
            metrics['level_metrics'][lvl] = lm_
        print('=' * 30, '{} - Classification metrics'.format(phase), '=' * 30)
        print('m-F1: {:.4f} Accuracy: {:.4f}'.format(metrics['m-f1'], metrics['accuracy']))
        if phase == 'train':
            self.img_rep = img_rep.unsqueeze(0).cpu()
            self.image_is_a_member_of = member_lists
        return metrics

    def _score(self, label_rep, img_rep):
        return ops.energy_matrix(label_rep, img_rep, getattr(self.criterion, 'K', None), self.criterion.energy)

    @torch.no_grad()
    def check_graph_embedding(self, reference_exact=None):
        """oe_h.py:2180-2247: label-graph reconstruction F1 over ALL label pairs.  One all-pairs energy launch (N x N)
        + a sort-based threshold sweep instead of N^2 python-indexed pairs and a process pool.  reference_exact (default: the
        attribute `reference_exact_eval`): the last label row stays zero as in the reference's chunk loop (:2230-2234) -- its NaN
        energies as an apex count as "not above any threshold" in EmbeddingMetrics, exactly as torch's comparisons count them."""
        exact = self.reference_exact_eval if reference_exact is None else bool(reference_exact)
        tc = self.graph_dict['graph_tc']
        N = self.n_classes
        hidden = self.levels_to_hide_for_epoch.get(self.epoch, []) if self.hide_levels else []
        pos = torch.zeros((N, N), dtype=torch.bool)
        keep = torch.zeros(N, dtype=torch.bool)
        def hid(u):
            return any(self.labelmap.level_start[l] <= u < self.labelmap.level_stop[l] for l in hidden)
        for u, v in tc.edges():
            if type(u) == str or type(v) == str or hid(u) or hid(v):
                continue
            pos[u, v] = True; keep[u] = True; keep[v] = True
        nodes = torch.nonzero(keep).flatten()
        rep = self.model(nodes.to(self.device)).detach().clone()
        if exact and len(nodes):
            rep[-1] = 0.0
        E = self._score(rep, rep).t().cpu()                                 # E[u, v] = E(apex u, point v)
        sub_pos = pos[nodes][:, nodes]
        off = ~torch.eye(len(nodes), dtype=torch.bool)
        m = EmbeddingMetrics(E[sub_pos], E[(~sub_pos) & off], 0.0, 'val')
        best = m.calculate_metrics()
        print('Checking graph reconstruction: +ve edges {}, -ve edges {}'.format(int(sub_pos.sum()), int(((~sub_pos) & off).sum())))
        return tuple(float(x) for x in best)


# the Riemannian helper methods of the reference (oe_h.py:1604-1644) as thin kernel calls, for API completeness
def _soft_clip(self, x):
    """oe_h.py:1604-1617 (in place): rows clipped into [r_in, 1 - 1e-5] -- the clip-only pass of the table-step kernel."""
    z = torch.zeros_like(x)
    ops.table_step_adam(x, z, z.clone(), z.clone(), 1, 0.0, self.criterion.K, riemannian=False, clip=True)
    return x


def _exp_map_x(self, x, v):
    """oe_h.py:1638-1644 exp_map_x(x, v) with Mobius addition; evaluated by the RSGD kernel with lr = -1 on the
    pre-rescaled tangent vector (the kernel multiplies by (1/lambda_x)^2 itself, so divide it out first)."""
    lam = self.lambda_x(x)
    g = v * lam ** 2
    out = x.clone()
    ops.table_step_rsgd(out, g.contiguous(), -1.0, self.criterion.K)
    return out


JointEmbeddings.soft_clip = _soft_clip
JointEmbeddings.exp_map_x = _exp_map_x
