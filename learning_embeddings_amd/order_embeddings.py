"""Host mirror of network/order_embeddings.py for config 1: the labels-only Euclidean
order-embedding criterion and its train step.  Reference: Embedder :179-199, OrderEmbeddingLoss :760-923,
OrderEmbedding.pass_samples :606-645."""
import numpy as np
import torch
import torch.nn as nn

from . import _lib, ops
from .hierarchy import NegativeGraph


class Embedder(nn.Module):
    """order_embeddings.py:179-199 (K=None: plain lookup; K given: direction * (norm + K))."""

    def __init__(self, embedding_dim, labelmap, K=None):
        super().__init__()
        self.labelmap = labelmap
        self.embedding_dim = embedding_dim
        self.K = K
        self.embeddings = nn.Embedding(self.labelmap.n_classes, self.embedding_dim)
        print('Embeds {} objects'.format(self.labelmap.n_classes))

    def forward(self, inputs):
        embeds = self.embeddings(inputs)
        if self.K:
            return self.soft_clip(embeds)
        return embeds

    def soft_clip(self, x):
        # order_embeddings.py:194-199: direction * (norm + K) -- the oe.py form of the soft clip, lec_image_softclip_fwd / _bwd
        return ops.ImageSoftClipFn.apply(x, float(self.K), _lib.IMAGE_SOFTCLIP_K)


class OrderEmbeddingLoss(torch.nn.Module):
    """order_embeddings.py:760-923."""

    def __init__(self, labelmap, neg_to_pos_ratio, alpha=1.0, pick_per_level=True, weigh_neg_term=False,
                 level_weights=None, weigh_pos_term=False):
        print('Using order-embedding loss!')
        torch.nn.Module.__init__(self)
        self.labelmap = labelmap
        self.neg_to_pos_ratio = neg_to_pos_ratio
        self.alpha = alpha
        self.pick_per_level = pick_per_level
        if weigh_neg_term:
            raise NotImplementedError('weigh_neg_term (order_embeddings.py:881-915) is off in every shipped configuration')
        self.weigh_neg_term = False
        self.weigh_pos_term = weigh_pos_term
        self.level_weights = level_weights if level_weights is not None else torch.ones((len(self.labelmap.levels)))
        self.negative_G = None
        self.G_tc = None

    def set_graph_tc(self, graph_tc):
        self.G_tc = graph_tc

    def set_negative_graph(self, n_G, mapping_from_node_to_ix, mapping_from_ix_to_node):
        """order_embeddings.py:785-795; dense matrix or a labels-only NegativeGraph."""
        if isinstance(n_G, NegativeGraph):
            self.negative_G = n_G
        else:
            self.negative_G = NegativeGraph.from_dense(np.asarray(n_G), self.labelmap.levels,
                                                       pick_per_level=self.pick_per_level, labels_only=True, seed=0)
        self.mapping_from_node_to_ix = mapping_from_node_to_ix
        self.mapping_from_ix_to_node = mapping_from_ix_to_node

    def seed_sampler(self, seed=0):
        self.negative_G.seed(seed)

    def sample_negative_edge(self, u=None, v=None, level_id=None):
        if (u is None) == (v is None):
            raise ValueError('Error! Both (u, v) given or neither (u, v) given!')
        node = u if u is not None else v
        return self.negative_G.draw(0 if u is not None else 1, self.mapping_from_node_to_ix[node], level_id or 0)

    @staticmethod
    def E_operator(x, y):
        return ops.pair_energy(x.contiguous(), y.contiguous(), None, 'order')

    def positive_pair(self, x, y):
        return self.E_operator(x, y)

    def negative_pair(self, x, y):
        e = self.E_operator(x, y)
        return torch.clamp(self.alpha - e, min=0.0), e

    def get_level_weight_for_edge(self, to):
        retval = torch.ones((len(to)))
        for level_ix, (s, e) in enumerate(zip(self.labelmap.level_start, self.labelmap.level_stop)):
            for i in range(len(to)):
                if s <= to[i] < e:
                    retval[i] = self.level_weights[level_ix]
        return retval

    def forward(self, model, inputs_from, inputs_to, status, phase, neg_to_pos_ratio):
        m = model.module if hasattr(model, 'module') else model
        dev = m.embeddings.weight.device
        f_ix = torch.as_tensor(inputs_from, dtype=torch.long, device=dev); t_ix = torch.as_tensor(inputs_to, dtype=torch.long, device=dev)
        if phase != 'train':                                                   # order_embeddings.py:851-864
            pf, pt = model(f_ix), model(t_ix)
            status = torch.as_tensor(status).to(dev)
            pi = (status == 1).nonzero().squeeze(dim=1); ni = (status == 0).nonzero().squeeze(dim=1)
            e_pos = self.positive_pair(pf[pi], pt[pi])
            neg_term, e_neg = self.negative_pair(pf[ni], pt[ni])
            return pf, pt, torch.sum(e_pos) + torch.sum(neg_term), e_pos, e_neg
        if getattr(m, 'K', None):
            raise NotImplementedError('train step with the Euclidean-cone Embedder (K given) is outside config 1')
        B = len(inputs_from)
        fa = np.asarray([self.mapping_from_node_to_ix[int(x)] for x in inputs_from], dtype=np.int32)
        ta = np.asarray([self.mapping_from_node_to_ix[int(x)] for x in inputs_to], dtype=np.int32)
        neg = self.negative_G.draw_batch(fa, ta, self.neg_to_pos_ratio)        # :886-915 host loop, bit-exact stream
        self.last_negatives = neg
        lw = self.get_level_weight_for_edge(to=list(ta)).to(dev)               # :868 (weights both terms unless weigh_pos_term)
        to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)
        loss, e_pos, e_neg = ops.JointLossFn.apply(m.embeddings.weight, None, to_dev(fa), to_dev(ta), to_dev(neg),
                                                   lw.float().contiguous(), 0.0, self.alpha, _lib.ENERGY_ORDER,
                                                   _lib.LABEL_RAW, _lib.IMAGE_RAW)
        with torch.no_grad():
            pf, pt = m.embeddings.weight[f_ix], m.embeddings.weight[t_ix]
        return pf, pt, loss, e_pos, e_neg.reshape(-1)
