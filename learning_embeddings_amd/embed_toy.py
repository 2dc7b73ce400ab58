"""Host mirror of network/embed_toy.py: the synthetic b-ary tree labelmap (embed_toy.py:29-62) and a labels-only
trainer step for config 1 (ToyOrderEmbedding / OrderEmbedding.pass_samples, order_embeddings.py:606-645)."""
import numpy as np
import torch

from . import ops
from .hierarchy import NegativeGraph
from .order_embeddings import Embedder, OrderEmbeddingLoss


class ToyGraph:
    """The synthetic b-ary tree labelmap of config 1 (the duck type network/embed_toy.py:29-62 builds: `levels, level_names,
    level_start, level_stop, n_classes, classes, edges`).  Level l (1-based name) holds b**l nodes; node i of a level is the
    parent of nodes b*i .. b*i + b - 1 of the next one; global ids are level offsets + the index inside the level.  Built from
    that contract by integer arithmetic; fixture F7 pins `levels` and `edges` against the reference."""

    def __init__(self, levels=4, branching_factor=3):
        b = int(branching_factor)
        self.n_levels, self.branching_factor = levels, b
        depth = levels - 1                                        # the root is implicit: `levels` counts it, the labelmap does not
        self.levels = [b ** (l + 1) for l in range(depth)]
        self.level_names = [str(l + 1) for l in range(depth)]
        offs = np.concatenate(([0], np.cumsum(self.levels))).astype(int)
        self.level_start, self.level_stop = offs[:-1].tolist(), offs[1:].tolist()
        self.n_classes = int(offs[-1])
        self.classes = ['%s_%d' % (nm, i) for nm, n in zip(self.level_names, self.levels) for i in range(n)]
        self.edges = set()
        for l in range(depth - 1):
            child = np.arange(self.levels[l + 1])
            self.edges.update(zip((self.level_start[l] + child // b).tolist(), (self.level_start[l + 1] + child).tolist()))


class ToyOrderEmbedding:
    """The parts of ToyOrderEmbedding / OrderEmbedding (embed_toy.py:65-135, order_embeddings.py:308-693) on the train
    path: model + criterion wiring and one optimisation step (Adam over the label table, no Riemannian rescale)."""

    def __init__(self, labelmap, criterion, lr, batch_size, embedding_dim, neg_to_pos_ratio, alpha=1.0,
                 pick_per_level=True, random_seed=0, device='cuda'):
        torch.manual_seed(random_seed)
        self.labelmap, self.criterion, self.lr, self.batch_size = labelmap, criterion, lr, batch_size
        self.embedding_dim, self.neg_to_pos_ratio = embedding_dim, neg_to_pos_ratio
        self.device = torch.device(device)
        self.model = Embedder(embedding_dim=embedding_dim, labelmap=labelmap).to(self.device)
        ident = {i: i for i in range(labelmap.n_classes)}
        self.negative_graph = NegativeGraph.from_labelmap(labelmap, pick_per_level=pick_per_level, labels_only=True, seed=0)
        self.criterion.set_negative_graph(self.negative_graph, ident, ident)
        w = self.model.embeddings.weight
        self.m = torch.zeros_like(w.data); self.v = torch.zeros_like(w.data); self.step = 0

    def train_step(self, inputs_from, inputs_to):
        w = self.model.embeddings.weight
        w.grad = None
        _, _, loss, e_pos, e_neg = self.criterion(self.model, inputs_from, inputs_to,
                                                  torch.ones(len(inputs_from)), 'train', self.neg_to_pos_ratio)
        loss.backward()
        self.step += 1
        ops.table_step_adam(w.data, w.grad.contiguous(), self.m, self.v, self.step, self.lr, 0.0, riemannian=False, clip=False)
        return loss.detach(), e_pos, e_neg
