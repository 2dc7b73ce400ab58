"""Host mirror of network/embed_toy.py: the synthetic b-ary tree labelmap (embed_toy.py:29-62) and a labels-only
trainer step for config 1 (ToyOrderEmbedding / OrderEmbedding.pass_samples, order_embeddings.py:606-645)."""
import numpy as np
import torch

from . import ops
from .hierarchy import NegativeGraph
from .order_embeddings import Embedder, OrderEmbeddingLoss


class ToyGraph:
    """embed_toy.py:29-62: levels = [b**i for i in 1..levels-1]; duck-typed labelmap with `edges`."""

    def __init__(self, levels=4, branching_factor=3):
        self.n_levels = levels
        self.branching_factor = branching_factor
        self.levels = [self.branching_factor ** i for i in range(1, self.n_levels)]
        self.level_names = [str(i) for i in range(1, self.n_levels)]
        for level_id, level_name in enumerate(self.level_names):
            setattr(self, level_name, {'{}_{}'.format(level_name, str(i)): i for i in range(self.levels[level_id])})
        for level_id, level_name in enumerate(self.level_names[:-1]):
            setattr(self, 'child_of_' + level_name,
                    {'{}_{}'.format(level_name, str(i)): ['{}_{}'.format(self.level_names[level_id + 1], str(j + (self.branching_factor * i)))
                                                          for j in range(self.branching_factor)] for i in range(self.levels[level_id])})
        self.n_classes = sum(self.levels)
        self.classes = [key for class_list in [getattr(self, n) for n in self.level_names] for key in class_list]
        self.level_stop, self.level_start = [], []
        for level_id, level_len in enumerate(self.levels):
            self.level_start.append(0 if level_id == 0 else self.level_stop[level_id - 1])
            self.level_stop.append(self.level_start[level_id] + level_len)
        self.edges = set()
        for level_id, level_name in enumerate(self.level_names[:-1]):
            child_of = getattr(self, 'child_of_' + level_name)
            for parent in child_of:
                for child in child_of[parent]:
                    u = getattr(self, level_name)[parent] + self.level_start[level_id]
                    v = getattr(self, self.level_names[level_id + 1])[child] + self.level_start[level_id + 1]
                    self.edges.add((u, v))


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
