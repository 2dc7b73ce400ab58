"""Host-side mirror of the reference's joint image+label hyperbolic-cone trainer, `network/oe_h.py`.

Same class names, constructor arguments, method names and error behaviour as the reference for the hot path
(SURVEY.md 8b), so its call sites read unchanged; underneath, every tensor operation of the path is one of the HIP
kernels in liblecone.so (ops.py) and negative sampling is the bit-exact C++ sampler (hierarchy.NegativeGraph).

Naming trap kept on purpose: `EuclideanConesWithImagesHypernymLoss` IS the hyperbolic entailment-cone loss
(oe_h.py:739; its E_operator :811-833 is the Poincare-ball cone angle).

Deliberate deviations (all documented in DESIGN.md):
  * one CNN forward per DISTINCT image per step (the reference re-embeds every positive image K more times in separate
    BatchNorm batches, oe_h.py:980-985,1003-1009); parity is asserted at the embedding boundary;
  * no host synchronisation inside the step (the reference syncs twice per step, oe_h.py:1752-1753,1774);
  * the dense (N+M)^2 negative adjacency is accepted for drop-in use but converted to CSR lists once.
"""
import copy
import os
import random
import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops
from .hierarchy import NegativeGraph
from .image_store import ImageRef
from .resnet import resnet18, resnet50

random.seed(0)                                                   # oe_h.py:42


def inner_radius_of(K):
    return 2 * K / (1 + np.sqrt(1 + 4 * K * K))                  # oe_h.py:59


def _unwrap(model):
    return model.module if hasattr(model, 'module') else model   # reference wraps in nn.DataParallel (oe_h.py:1434)


# ------------------------------------------------------------------------------------------------------------------
class Embedder(nn.Module):
    """oe_h.py:51-110.  Label-embedding table + exp-map style projection into the ball, as one HIP kernel."""

    def __init__(self, embedding_dim, labelmap, normalize, K=None):
        super().__init__()
        self.labelmap = labelmap
        self.embedding_dim = embedding_dim
        self.normalize = normalize
        self.K = K
        self.epsilon = 1e-5
        if self.normalize == 'max_norm':
            self.embeddings = nn.Embedding(self.labelmap.n_classes, self.embedding_dim, max_norm=1.0)
        else:
            self.embeddings = nn.Embedding(self.labelmap.n_classes, self.embedding_dim)
        print('Embeds {} objects'.format(self.labelmap.n_classes))
        if K:
            self.inner_radius = inner_radius_of(self.K)
            with torch.no_grad():                                # oe_h.py:68-73: rows renormed to r_in + U[0, 0.05)
                w = self.embeddings.weight.data
                norm = torch.norm(w, dim=1, keepdim=True).repeat(1, self.embedding_dim)
                new_norm = self.inner_radius + torch.rand((w.shape[0])) * 0.05
                new_norm = torch.unsqueeze(new_norm, 1).repeat(1, self.embedding_dim)
                self.embeddings.weight.data = new_norm * w / norm
            self.inner_radius_h = self.arctanh(torch.tensor(self.inner_radius))
        else:
            self.inner_radius = None

    @property
    def device(self):
        return self.embeddings.weight.device

    def forward(self, inputs):
        w = self.embeddings.weight
        if not self.K:
            return F.embedding(inputs, w)
        if self.normalize == 'unit_norm':
            raise NotImplementedError("normalize='unit_norm' is outside the hot path (JointEmbeddings passes None, oe_h.py:1392)")
        shp = inputs.shape
        out = ops.LabelProjectFn.apply(w, inputs.reshape(-1), self.K)
        return out.view(*shp, self.embedding_dim)

    @staticmethod
    def arctanh(x):
        x = x.clamp(-1 + 1e-5, 1 - 1e-5)
        return (torch.log(1 + x) - torch.log(1 - x)) * 0.5


class _ImageNetBase(nn.Module):
    """Shared by FeatCNN18 / FeatCNN: ResNet -> Linear(., D) -> soft_clip x/|x|*(|x| + r_in) (oe_h.py:313-328)."""

    def _setup(self, K):
        self.K = K
        self.inner_radius = inner_radius_of(K) if K else None

    @property
    def device(self):
        return next(self.parameters()).device

    cnn_passes = 1           # > 1 (fp32 / bf16, training): forward_raw pushes the batch through the backbone as that many concurrent parts

    def __deepcopy__(self, memo):
        """copy.deepcopy of a network that has trained: the HIP streams of its concurrent passes are per-object launch resources, not state
        (a Stream cannot be copied); the copy creates its own on first use."""
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k != '_pass_streams':
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def _forward_raw_passes(self, x, split=None):
        """forward_raw with the rows of x as concurrent parts, one HIP stream each: the backbone up to the pooled features per part
        (BatchNorm statistics per part, running statistics updated in part order), then the fully connected layer once over all rows
        on the caller's stream.  `split` (the criterion passes it): rows [0, split) are the positives' images, the rest images drawn as
        negatives -- the parts are then the reference's own separate forwards (oe_h.py:980-985 | 1003-1009); without it, `cnn_passes` equal
        parts.  Autograd runs every part's backward on the stream its forward ran on; the caller calls join_passes() after backward, zeroes the gradient
        slots once per step and has set the backbone's bn_grad_accumulate (trainers' __init__)."""
        cur = torch.cuda.current_stream()
        n = x.shape[0]
        if split is not None and 0 < split < n:
            bounds = [0, int(split), n]
        else:
            h = -(-n // self.cnn_passes)
            bounds = list(range(0, n, h)) + [n]
        streams = self.__dict__.setdefault('_pass_streams', [])
        while len(streams) < len(bounds) - 1:
            streams.append(torch.cuda.Stream())
        parts, order, used = [], {}, []
        for p in range(len(bounds) - 1):
            xs = x[bounds[p]:bounds[p + 1]]
            st = streams[p]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                with torch.autocast('cuda', dtype=self.compute_dtype, enabled=self.compute_dtype != torch.float32):
                    f = self.model(xs, pooled_only=True, pass_order=(order, p))
            f.record_stream(cur); parts.append(f); used.append(st)
        for st in used:
            x.record_stream(st); cur.wait_stream(st)
        self.__dict__['_passes_in_flight'] = used                   # their backward runs on these streams: join_passes() after loss.backward()
        with torch.autocast('cuda', dtype=self.compute_dtype, enabled=self.compute_dtype != torch.float32):
            return self.model.fc(torch.cat(parts)).float()

    def join_passes(self):
        """Make the current stream wait for the streams the concurrent passes' backward ran on.  Autograd only joins the streams on which an
        AccumulateGrad node ran; this backbone's parameter gradients are written in place by its kernels (weight gradients, BatchNorm
        d gamma / d beta into the flat arena), so nothing tells the engine about the pass streams: whoever reads the gradients next
        (all-reduce, optimizer) must wait for them explicitly."""
        cur = torch.cuda.current_stream()
        for st in self.__dict__.pop('_passes_in_flight', []):
            cur.wait_stream(st)

    def forward_raw(self, x, split=None):
        """CNN output BEFORE soft_clip, fp32 [n, D] -- the fused loss applies soft_clip itself.  split: see _forward_raw_passes."""
        if self.channels_last and x.dim() == 4:
            x = x.contiguous(memory_format=torch.channels_last)
        if (self.cnn_passes > 1 and x.is_cuda and self.compute_dtype in (torch.float32, torch.bfloat16) and self.training and torch.is_grad_enabled()
                and x.dim() == 4 and x.shape[0] >= 4 * self.cnn_passes and (split is None or 2 <= split <= x.shape[0] - 2)):
            return self._forward_raw_passes(x, split)
        if self.compute_dtype != torch.float32:
            with torch.autocast('cuda', dtype=self.compute_dtype):
                y = self.model(x)
        else:
            y = self.model(x)
        return y.float()

    def forward_pooled(self, x, pass_order=None):
        """The backbone up to its global average pooling, [n, fc.in_features]; `head` finishes forward_raw.  (the
        engine's concurrent passes, which pool per pass and apply the fully connected layer once; 16-bit backbones run it under autocast.)"""
        if self.channels_last and x.dim() == 4:
            x = x.contiguous(memory_format=torch.channels_last)
        if self.compute_dtype != torch.float32:
            with torch.autocast('cuda', dtype=self.compute_dtype):
                return self.model(x, pooled_only=True, pass_order=pass_order)
        return self.model(x, pooled_only=True, pass_order=pass_order)

    def head(self, pooled):
        if self.compute_dtype != torch.float32:
            with torch.autocast('cuda', dtype=self.compute_dtype):
                return self.model.fc(pooled).float()
        return self.model.fc(pooled).float()

    def forward(self, x):
        y = self.forward_raw(x)
        return self.soft_clip(y) if self.K else y

    def soft_clip(self, x):
        return ops.ImageSoftClipFn.apply(x, self.K)


class FeatCNN18(_ImageNetBase):
    """oe_h.py:281-328: ResNet-18 with fc -> Linear(512, output_dim).  `pretrained` weights cannot be downloaded here
    (no network): pass `weights=<state_dict or path>` to load a torchvision-format checkpoint, else default init."""

    def __init__(self, image_dir, path_to_exp='../exp', input_dim=2048, output_dim=10,
                 exp_name='ethec_resnet50_lr_1e-5_1_1_1_1/', K=None, weights=None, compute_dtype=torch.float32,
                 channels_last=True, arch='resnet18'):
        super().__init__()
        self.path_to_exp = os.path.join(path_to_exp, exp_name)
        self.image_dir = image_dir
        self._setup(K)
        self.compute_dtype = compute_dtype
        self.channels_last = channels_last
        self.arch = arch
        self.model = None
        self.load_model(weights)
        self.model.fc = nn.Linear(self.model.fc.in_features, output_dim)       # oe_h.py:302
        if channels_last:
            self.model = self.model.to(memory_format=torch.channels_last)

    def load_model(self, weights=None):
        # arch: 'resnet18' | 'resnet50' | a callable returning a resnet.ResNet (narrow stand-in backbones of the parity fixtures)
        self.model = self.arch() if callable(self.arch) else (resnet18() if self.arch == 'resnet18' else resnet50())
        if weights is not None:
            sd = torch.load(weights, map_location='cpu') if isinstance(weights, str) else weights
            sd = {k[len('module.'):] if k.startswith('module.') else k: v for k, v in sd.items()}
            self.model.load_state_dict(sd)


class FeatCNN(FeatCNN18):
    """oe_h.py:331-378: the ResNet-50 wrapper (fc -> Linear(2048, output_dim)).  The reference builds it from a trained
    classifier checkpoint through `Inference`; here the checkpoint is passed as `weights`."""

    def __init__(self, image_dir, path_to_exp='../exp', input_dim=2048, output_dim=10,
                 exp_name='ethec_resnet50_lr_1e-5_1_1_1_1/', K=None, weights=None, compute_dtype=torch.float32,
                 channels_last=True):
        super().__init__(image_dir, path_to_exp, input_dim, output_dim, exp_name, K, weights, compute_dtype,
                         channels_last, arch='resnet50')


class FeatNet(nn.Module):
    """oe_h.py:113-224: Linear(input_dim, D) on precomputed image features followed by the same ball projection as the
    label Embedder (used when --use_CNN is off).  The reference's lower-clip branch adds 1e-6 guards (:222); rows reach
    it only at |x| = 0, where both forms give r_in * direction to 1e-5 relative."""

    def __init__(self, normalize, input_dim=2048, output_dim=10, K=None):
        super().__init__()
        self.output_dim = output_dim
        self.normalize = normalize
        self.K = K
        self.inner_radius = inner_radius_of(K) if K else None
        self.fc1 = nn.Linear(input_dim, output_dim)
        self.epsilon = 1e-5

    def forward(self, x):
        shp = x.shape
        y = self.fc1(x).reshape(-1, self.output_dim).float()
        if self.K and self.normalize is None:
            idx = torch.arange(y.shape[0], device=y.device)
            y = ops.LabelProjectFn.apply(y, idx, self.K)
        elif self.normalize is not None:
            raise NotImplementedError('FeatNet normalize modes are outside the hot path')
        return y.view(*shp[:-1], self.output_dim)


def my_collate(data):
    """oe_h.py:435-444: keeps `from`/`to` as python lists (ints and image tensors mixed)."""
    from_data, to_data, status_data, original_from, original_to = [], [], [], [], []
    for data_item in data:
        from_data.append(data_item['from']); to_data.append(data_item['to'])
        status_data.append(data_item['status'])
        original_from.append(data_item['original_from']); original_to.append(data_item['original_to'])
    return {'from': from_data, 'to': to_data, 'status': torch.tensor(status_data), 'original_from': original_from,
            'original_to': original_to}


# ------------------------------------------------------------------------------------------------------------------
class _JointCriterionBase(torch.nn.Module):
    """Everything EuclideanConesWithImagesHypernymLoss and OrderEmbeddingWithImagesHypernymLoss share
    (oe_h.py:739-1058 and :1061-1315 differ only in E_operator and the Embedder they are paired with)."""

    energy = 'hyp_cone'
    default_image_proj = _lib.IMAGE_SOFTCLIP         # which soft_clip the fused kernel applies to raw CNN outputs

    def _init_common(self, labelmap, neg_to_pos_ratio, feature_dict, alpha, pick_per_level, use_CNN):
        torch.nn.Module.__init__(self)
        self.labelmap = labelmap
        self.neg_to_pos_ratio = neg_to_pos_ratio
        self.alpha = alpha
        self.device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
        self.mapping_from_node_to_ix = None
        self.mapping_from_ix_to_node = None
        self.negative_G = None
        self.feature_dict = feature_dict
        self.pick_per_level = pick_per_level
        self.use_CNN = use_CNN
        self.dataloader = None
        self.levels_to_hide = []
        self.n_labels = labelmap.n_classes

    # ---- reference API ------------------------------------------------------------------------------------------
    def set_levels_to_hide(self, list_of_levels):
        self.levels_to_hide = list(list_of_levels)
        if self.negative_G is not None:
            self.negative_G.set_levels_to_hide(self.levels_to_hide)

    def set_dataloader(self, dataloader):
        self.dataloader = dataloader

    def set_negative_graph(self, n_G, mapping_from_node_to_ix, mapping_from_ix_to_node):
        """oe_h.py:799-809.  `n_G` is either the reference's dense bool matrix (1 = negative edge) or a
        hierarchy.NegativeGraph.  The global `random` stream position is not touched; the sampler owns its own MT19937
        (seed with `seed_sampler`, default 0 = the reference's `random.seed(0)` at oe_h.py:42,1472)."""
        if isinstance(n_G, NegativeGraph):
            self.negative_G = n_G
        else:
            self.negative_G = NegativeGraph.from_dense(np.asarray(n_G), self.labelmap.levels,
                                                       pick_per_level=self.pick_per_level, seed=0)
        if self.negative_G.pick_per_level != bool(self.pick_per_level):
            raise ValueError('negative graph was built with pick_per_level=%s' % self.negative_G.pick_per_level)
        self.mapping_from_node_to_ix = mapping_from_node_to_ix
        self.mapping_from_ix_to_node = mapping_from_ix_to_node
        if self.levels_to_hide:
            self.negative_G.set_levels_to_hide(self.levels_to_hide)

    def seed_sampler(self, seed=0):
        self.negative_G.seed(seed)

    def get_img_features(self, x):
        """oe_h.py:770-797 (precomputed-feature path): list of names -> [1, n, F]; list of lists -> [m, n, F]."""
        if len(x) and isinstance(x[0], str):
            return torch.tensor(np.stack([np.asarray(self.feature_dict[n]) for n in x]), dtype=torch.float32).unsqueeze(0)
        return torch.tensor(np.stack([np.stack([np.asarray(self.feature_dict[n]) for n in sub]) for sub in x]), dtype=torch.float32)

    def positive_pair(self, x, y):
        return self.E_operator(x, y)

    def negative_pair(self, x, y):
        e = self.E_operator(x, y)                                # the reference evaluates it twice (oe_h.py:839)
        return torch.clamp(self.alpha - e, min=0.0), e

    def get_image_label_loss(self, e_for_u_v_positive, e_for_u_v_negative, weights=None):
        if weights is None:
            return torch.sum(e_for_u_v_positive) + torch.sum(torch.clamp(self.alpha - e_for_u_v_negative, min=0.0))
        weights = torch.as_tensor(weights, dtype=torch.float32, device=e_for_u_v_positive.device)
        return torch.sum(weights * e_for_u_v_positive) + torch.sum(
            weights * torch.sum(torch.clamp(self.alpha - e_for_u_v_negative, min=0.0), dim=1))

    def sample_negative_edge(self, u=None, v=None, level_id=None):
        """oe_h.py:849-902: returns the corrupted node's INDEX.  Raises IndexError on an empty candidate list like
        `random.choice`; raises ValueError where the reference only prints (both/neither of u, v given)."""
        if (u is None) == (v is None):
            raise ValueError('Error! Both (u, v) given or neither (u, v) given!')
        node = u if u is not None else v
        return self.negative_G.draw(0 if u is not None else 1, self.mapping_from_node_to_ix[node],
                                    0 if level_id is None else level_id)

    # ---- the hot path -------------------------------------------------------------------------------------------
    def _proj_flags(self, model):
        m = _unwrap(model)
        label_proj = _lib.LABEL_HYP if getattr(m, 'K', None) else _lib.LABEL_RAW
        return label_proj

    def forward(self, model, img_feat_net, inputs_from, inputs_to, original_from, original_to, status, phase):
        if phase != 'train':
            return self._forward_eval(model, img_feat_net, inputs_from, inputs_to)
        n2i = self.mapping_from_node_to_ix
        N = self.n_labels
        B = len(original_from)
        Kn = self.neg_to_pos_ratio
        ix_from = np.fromiter((n2i[o] for o in original_from), dtype=np.int32, count=B)
        ix_to = np.fromiter((n2i[o] for o in original_to), dtype=np.int32, count=B)
        dp = getattr(self, 'dp_global', None)
        pre = getattr(self, 'predrawn', None)
        if pre is not None:
            # the trainer's lookahead drew this batch's negatives one step ahead, in the reference's stream order, and asked the image
            # store for the images among them (oe_h_trainer.JointEmbeddings.train_epoch); same indices as the draw below would give
            self.predrawn = None
            p_from, p_to, neg = pre
            if not (np.array_equal(p_from, ix_from) and np.array_equal(p_to, ix_to)):
                raise RuntimeError('negatives were drawn ahead for a different batch than the one passed in')
        elif dp is None:
            neg = self.negative_G.draw_batch(ix_from, ix_to, Kn)         # oe_h.py:940-957, bit-exact stream
        else:
            # data parallel: every rank walks the GLOBAL batch's stream (same order as a single process would) and keeps
            # its shard, so the negatives do not depend on the world size and no collective is needed
            g_from, g_to, lo, hi = dp
            if not (np.array_equal(g_from[lo:hi], ix_from) and np.array_equal(g_to[lo:hi], ix_to)):
                raise RuntimeError('data-parallel shard does not match the global batch order')
            neg = self.negative_G.draw_batch(g_from, g_to, Kn)[lo:hi]
        self.last_negatives = neg
        # one CNN forward over the DISTINCT images of the step: the batch's own images + images drawn as negatives.  A row is either a
        # float tensor (the reference's host path) or (name, mirrored) for the image store to build on the GPU.
        if getattr(self, 'reference_exact_batches', False):
            return self._forward_reference_batches(model, img_feat_net, inputs_from, inputs_to, original_from, original_to, neg)
        store = getattr(self, 'image_store', None)
        i2n = self.mapping_from_ix_to_node
        slot, rows = {}, []
        for elem, ix in zip(list(inputs_from) + list(inputs_to), np.concatenate([ix_from, ix_to]).tolist()):
            if ix >= N and ix not in slot:
                slot[ix] = len(rows); rows.append(self._image_row(elem, i2n[ix]))
        n_pos_rows = len(rows)                                          # rows [0, n_pos_rows): the batch's own images; behind them: image negatives
        for ix in np.unique(neg[neg >= N]).tolist():
            if ix not in slot:
                slot[ix] = len(rows); rows.append(self._image_row(i2n[ix], i2n[ix]))
        dev = _unwrap(model).embeddings.weight.device
        feats, image_proj = None, _lib.IMAGE_RAW
        self.last_cnn_rows = len(rows)                                  # distinct images of the step = rows of the one CNN batch
        if rows:
            batch = self._image_batch(rows, dev, store)
            if self.use_CNN and hasattr(img_feat_net, 'forward_raw') and getattr(img_feat_net, 'K', None):
                if isinstance(_unwrap(img_feat_net), _ImageNetBase):
                    feats = img_feat_net.forward_raw(batch, split=n_pos_rows)       # concurrent passes cut at the positives | negatives boundary
                else:
                    feats = img_feat_net.forward_raw(batch)
                image_proj = self.default_image_proj
            else:
                feats = _unwrap(img_feat_net)(batch).reshape(len(rows), -1).float()

        ring = self.__dict__.get('_h2d')
        if ring is None:
            from .parallel import PinnedRing
            ring = self.__dict__['_h2d'] = PinnedRing()
        ring.begin_step()

        def codes(a):
            a = np.asarray(a, dtype=np.int64)
            if slot:
                lut = np.full(int(max(slot)) - N + 1, -1, dtype=np.int64)
                for ix, s in slot.items():
                    lut[ix - N] = s
                img = a >= N
                out = a.copy()
                out[img] = -1 - lut[a[img] - N]
                a = out
            return ring.upload(a.astype(np.int32), dev)             # pinned: see parallel.PinnedRing

        c_from, c_to, c_neg = codes(ix_from), codes(ix_to), codes(neg)
        ring.end_step()
        return self.forward_indices(model, feats, c_from, c_to, c_neg, image_proj=image_proj)

    def _image_row(self, elem, name):
        """One row of a CNN batch: a float tensor (the reference's host path: an item's tensor, or get_image(name)) or (name, mirrored) for the
        image store to build on the GPU.  `elem`: what the batch carries for the image (tensor / ImageRef) or its name (images drawn as
        negatives: get_image's val/test transform, no flip -- oe_h.py:668-677)."""
        store = getattr(self, 'image_store', None)
        if isinstance(elem, ImageRef):
            if store is None:
                raise RuntimeError('the batch carries ImageRef handles but the criterion has no image_store')
            if elem.pixels is not None:
                store.offer(elem.name, elem.pixels)
            return (elem.name, elem.flip)
        if torch.is_tensor(elem):
            return elem
        if store is not None and store.holds(name):
            return (name, False)
        return self.dataloader.get_image(name)

    def _forward_reference_batches(self, model, img_feat_net, inputs_from, inputs_to, original_from, original_to, neg):
        """reference_exact_batches = True: the train branch composed exactly like the reference's (oe_h.py:929-967), CNN batch by CNN batch.
        The default path embeds every DISTINCT image of a step once; the reference runs up to four separate forwards -- the image ends of
        the positives' from side, of their to side (oe_h.py:980-985, 1003-1009 through calculate_from_and_to_emb: the batch's own tensors,
        train transform), then of the 2K B negative pairs' from side and to side, where every FIXED image end is embedded again, K times,
        through get_image (no flip) -- each forward its own BatchNorm batch (duplicates included) and its own update of the running
        statistics.  Same kernels as everywhere (Embedder / soft_clip / E_operator / the backbone), unfused: the loss is assembled by
        autograd over them as the reference assembles it.  Opt-in: it costs (1 + K) times the CNN rows."""
        B, Kn = len(original_from), self.neg_to_pos_ratio
        i2n = self.mapping_from_ix_to_node
        pf, pt = self.calculate_from_and_to_emb(model, img_feat_net, inputs_from, inputs_to)
        e_pos = self.positive_pair(pf, pt)
        negative_from, negative_to = [None] * (2 * Kn * B), [None] * (2 * Kn * B)
        for b in range(B):                                              # oe_h.py:940-957 (the indices are the ones already drawn)
            for p in range(Kn):
                negative_from[2 * Kn * b + p] = original_from[b]
                negative_to[2 * Kn * b + p] = i2n[int(neg[b, p])]
                negative_from[2 * Kn * b + p + Kn] = i2n[int(neg[b, p + Kn])]
                negative_to[2 * Kn * b + p + Kn] = original_to[b]
        nf, nt = self.calculate_from_and_to_emb(model, img_feat_net, negative_from, negative_to)
        _, e_neg = self.negative_pair(nf, nt)
        e_neg = e_neg.view(B, 2 * Kn, -1)
        loss = torch.sum(self.get_image_label_loss(e_pos.reshape(-1), e_neg.reshape(B, 2 * Kn), [1.0] * B))
        self.last_cnn_rows = sum(1 for lst in (inputs_from, inputs_to, negative_from, negative_to) for e in lst if not isinstance(e, (int, np.integer)))
        return loss, e_pos, e_neg

    def _image_batch(self, rows, dev, store):
        """The step's CNN batch on the device.  Store-backed rows: ONE gather kernel (image_store.ImageStore.batch).  Tensor rows (the
        reference's host path, in-memory synthetic stores): one stack -- host tensors through a pinned buffer and ONE asynchronous copy
        instead of a pageable copy per image."""
        st = [j for j, r in enumerate(rows) if isinstance(r, tuple)]
        if len(st) == len(rows):
            return store.batch([r[0] for r in rows], [r[1] for r in rows])
        tens = [r for r in rows if not isinstance(r, tuple)]
        if all(t.device == dev for t in tens):
            t = torch.stack(tens)
        elif all(not t.is_cuda for t in tens) and all(t.shape == tens[0].shape and t.dtype == tens[0].dtype for t in tens):
            pin = self.__dict__.get('_pin')
            if pin is None or pin[0].shape[1:] != tens[0].shape or pin[0].dtype != tens[0].dtype or pin[0].shape[0] < len(tens):
                pin = self.__dict__['_pin'] = [torch.empty((max(len(tens), 64),) + tuple(tens[0].shape), dtype=tens[0].dtype).pin_memory(), None]
            if pin[1] is not None:
                pin[1].synchronize()                                     # the previous step's copy out of this buffer
            torch.stack(tens, out=pin[0][:len(tens)])
            t = pin[0][:len(tens)].to(dev, non_blocking=True)
            pin[1] = torch.cuda.Event(); pin[1].record()
        else:
            t = torch.stack([x.to(dev, non_blocking=True) for x in tens])
        if not st:
            return t
        out = torch.empty((len(rows),) + tuple(t.shape[1:]), dtype=torch.float32, device=dev)
        tt = [j for j, r in enumerate(rows) if not isinstance(r, tuple)]
        out[torch.tensor(st, device=dev)] = store.batch([rows[j][0] for j in st], [rows[j][1] for j in st])
        out[torch.tensor(tt, device=dev)] = t.float()
        return out

    def forward_indices(self, model, feats, pos_from, pos_to, neg, weights=None, image_proj=None):
        """The device part of the train step: node codes in, (loss, e_pos [B], e_neg [B,2K,1]) out, one kernel."""
        if image_proj is None:
            image_proj = self.default_image_proj
        w = _unwrap(model).embeddings.weight
        Kc = getattr(self, 'K', None) or 0.0
        loss, e_pos, e_neg = ops.JointLossFn.apply(w, feats, pos_from, pos_to, neg, weights, Kc, self.alpha,
                                                   ops.ENERGY[self.energy], self._proj_flags(model),
                                                   image_proj if feats is not None else _lib.IMAGE_RAW)
        return loss, e_pos, e_neg.view(e_neg.shape[0], e_neg.shape[1], 1)

    def _forward_eval(self, model, img_feat_net, inputs_from, inputs_to):
        """oe_h.py:908-925: pre-built [B, 1+2K, D] embeddings (slot 0 positive, the rest negatives)."""
        f, t = self.calculate_from_and_to_emb(model, img_feat_net, inputs_from, inputs_to)
        e_pos = self.positive_pair(f[:, 0, :], t[:, 0, :])
        _, e_neg = self.negative_pair(f[:, 1:, :], t[:, 1:, :])
        loss = torch.sum(self.get_image_label_loss(e_pos, e_neg))
        return loss, e_pos, e_neg

    def calculate_from_and_to_emb(self, model, img_feat_net, from_elem, to_elem):
        """oe_h.py:969-1058: embeds two parallel lists whose elements are label ints or images (tensors / names)."""
        dev = _unwrap(model).embeddings.weight.device

        def embed(elems):
            if torch.is_tensor(elems):
                return elems
            lab_ix = [i for i, e in enumerate(elems) if isinstance(e, (int, np.integer))]
            img_ix = [i for i, e in enumerate(elems) if not isinstance(e, (int, np.integer))]
            out = None
            if img_ix:
                if self.use_CNN:
                    rows = [self._image_row(elems[i], elems[i]) for i in img_ix]
                    img_emb = img_feat_net(self._image_batch(rows, dev, getattr(self, 'image_store', None)))
                else:
                    img_emb = img_feat_net(self.get_img_features([elems[i] for i in img_ix]).to(dev)).reshape(len(img_ix), -1)
                out = torch.zeros((len(elems), img_emb.shape[-1]), device=dev)
                out[img_ix, :] = img_emb.float()
            if lab_ix:
                lab_emb = model(torch.tensor([int(elems[i]) for i in lab_ix], dtype=torch.long, device=dev))
                if out is None:
                    out = torch.zeros((len(elems), lab_emb.shape[-1]), device=dev)
                out[lab_ix, :] = lab_emb
            return out

        return embed(from_elem), embed(to_elem)


class EuclideanConesWithImagesHypernymLoss(_JointCriterionBase):
    """oe_h.py:739-1058 -- the HYPERBOLIC entailment-cone criterion (see the naming trap above)."""
    energy = 'hyp_cone'

    def __init__(self, labelmap, neg_to_pos_ratio, feature_dict, alpha, pick_per_level=False, K=0.1, use_CNN=False):
        print('Using Euclidean cones loss!')
        self._init_common(labelmap, neg_to_pos_ratio, feature_dict, alpha, pick_per_level, use_CNN)
        self.K = K
        self.inner_radius = inner_radius_of(K)
        self.epsilon = 1e-5

    def E_operator(self, x, y):
        return ops.pair_energy(x, y, self.K, 'hyp_cone')


class OrderEmbeddingWithImagesHypernymLoss(_JointCriterionBase):
    """oe_h.py:1061-1315 -- the joint criterion with the Euclidean order-violation energy."""
    energy = 'order'

    def __init__(self, labelmap, neg_to_pos_ratio, feature_dict, alpha, pick_per_level=False, use_CNN=False):
        print('Using order-embedding loss!')
        self._init_common(labelmap, neg_to_pos_ratio, feature_dict, alpha, pick_per_level, use_CNN)

    @staticmethod
    def E_operator(x, y):
        return ops.pair_energy(x, y, None, 'order')


from .oe_h_trainer import (DiGraph, transitive_closure, create_combined_graphs, ETHECHierarchyWithImages,  # noqa: E402,F401
                           EmbeddingMetrics, GlobalBatchSampler, JointEmbeddings)
