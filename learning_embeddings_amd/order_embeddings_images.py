"""Host mirror of the API surface of network/order_embeddings_images.py -- the older joint trainer: Euclidean
order-violation energy on PRECOMPUTED 2048-d image features through a 2-layer FeatNet.

Mirrored: FeatNet (:143-178), my_collate (:181-188), OrderEmbeddingWithImagesLoss (:371-470) and the train step of
EmbeddingLabelsWithImages (:712-960).  Energies, hinge, loss and gradients run in the same fused HIP kernel as the
hyperbolic path (ENERGY_ORDER, raw label rows, raw image points).  NOT mirrored bit-for-bit: this trainer's negative
sampler (`random.choice(list(set_a - set_b))` over python sets of str/int nodes, :433-442) -- set iteration order of str
nodes depends on PYTHONHASHSEED, so the reference's own stream is not reproducible ("parity unpinned" for the negative
indices of this legacy trainer).  The candidate SETS are the reference's: for `u` fixed, images that are not
transitive-closure neighbours of u; for `v` fixed, labels that are not ancestors of v -- exactly the slot-L rule of the
oe_h.py sampler, drawn here from the bit-exact MT19937 stream in the same call order."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops
from .hierarchy import NegativeGraph
from .order_embeddings import Embedder, OrderEmbeddingLoss


class FeatNet(nn.Module):
    """order_embeddings_images.py:143-178: Linear(input_dim, 512) -> ReLU -> |Linear(512, output_dim)| (+ optional norm)."""

    def __init__(self, normalize, input_dim=2048, output_dim=10):
        super().__init__()
        self.fc1 = nn.Linear(input_dim, 512)
        self.fc2 = nn.Linear(512, output_dim)
        self.normalize = normalize

    def forward(self, x):
        x = torch.abs(self.fc2(F.relu(self.fc1(x))))
        if self.normalize == 'unit_norm':
            shp = x.shape
            x = F.normalize(x.view(-1, shp[-1]), p=2, dim=1).view(shp)
        elif self.normalize == 'max_norm':
            shp = x.shape
            x = x.view(-1, shp[-1])
            n = torch.norm(x, p=2, dim=1, keepdim=True)
            x = torch.where(n > 1.0, x / n.clamp_min(1e-12), x).view(shp)
        return x


def my_collate(data):
    from_data, to_data, status_data = [], [], []
    for d in data:
        from_data.append(d['from']); to_data.append(d['to']); status_data.append(d['status'])
    return {'from': torch.tensor(from_data), 'to': to_data, 'status': torch.tensor(status_data)}


class OrderEmbeddingWithImagesLoss(OrderEmbeddingLoss):
    """order_embeddings_images.py:371-470.  `img_feat_net(list_of_image_names)` must return the image points [n, D]
    (the reference's FeatNet wrapper looks the names up in its precomputed feature table)."""

    def __init__(self, labelmap, neg_to_pos_ratio, alpha=1.0):
        OrderEmbeddingLoss.__init__(self, labelmap, neg_to_pos_ratio, alpha, pick_per_level=True)
        self.image_nodes_in_graph = set(); self.non_image_nodes_in_graph = set()

    def set_graph_tc(self, graph_tc):
        """:378-391: the training transitive closure; here it also builds the CSR negative graph (labels 0..N-1, images
        N.. in node order)."""
        self.G_tc = graph_tc
        nodes = list(graph_tc)
        self.image_nodes_in_graph = set(n for n in nodes if type(n) == str)
        self.non_image_nodes_in_graph = set(n for n in nodes if type(n) != str)
        N = self.labelmap.n_classes
        names = [n for n in nodes if type(n) == str]
        self.mapping_from_node_to_ix = {i: i for i in range(N)}
        self.mapping_from_node_to_ix.update({n: N + j for j, n in enumerate(names)})
        self.mapping_from_ix_to_node = {v: k for k, v in self.mapping_from_node_to_ix.items()}
        label_edges = [(u, v) for u, v in graph_tc.edges() if type(u) != str and type(v) != str]
        ptr = np.zeros(len(names) + 1, dtype=np.int64); adj = []
        for j, n in enumerate(names):
            adj.extend(int(p) for p in graph_tc.predecessors(n)); ptr[j + 1] = len(adj)
        self.negative_G = NegativeGraph(self.labelmap.levels, label_edges, ptr, np.asarray(adj, dtype=np.int32),
                                        pick_per_level=True, seed=0)
        self._slot_L = len(self.labelmap.levels)

    def forward(self, model, img_feat_net, inputs_from, inputs_to, status, phase):
        m = model.module if hasattr(model, 'module') else model
        dev = m.embeddings.weight.device
        if phase != 'train':                                                    # :400-416
            pf = model(torch.as_tensor(inputs_from, dtype=torch.long, device=dev)); pt = img_feat_net(inputs_to)
            st = torch.as_tensor(status).to(dev)
            pi = (st == 1).nonzero().squeeze(1); ni = (st == 0).nonzero().squeeze(1)
            e_pos = self.positive_pair(pf[pi], pt[pi]); neg_term, e_neg = self.negative_pair(pf[ni], pt[ni])
            return pf, pt, torch.sum(neg_term), e_pos, e_neg
        # train (:418-467): per batch entry a list of positives; negatives: K x (corrupt image, corrupt label)
        frm = [int(x) for b in inputs_from for x in (b.tolist() if torch.is_tensor(b) else [b])]
        to = [x for b in inputs_to for x in (b if isinstance(b, (list, tuple)) else [b])]
        n2i = self.mapping_from_node_to_ix
        N, K, B = self.labelmap.n_classes, self.neg_to_pos_ratio, len(frm)
        fa = np.asarray([n2i[u] for u in frm], dtype=np.int32); ta = np.asarray([n2i[v] for v in to], dtype=np.int32)
        neg = np.empty((B, 2 * K), dtype=np.int32)
        for b in range(B):
            for p in range(K):
                neg[b, p] = self.negative_G.draw(0, int(fa[b]), self._slot_L)        # an image that is not below u
                neg[b, K + p] = self.negative_G.draw(1, int(ta[b]), self._slot_L)    # a label that is not above v
        self.last_negatives = neg
        names = sorted(set(int(i) for i in ta.tolist()) | set(int(i) for i in neg[neg >= N].tolist()))
        slot = {ix: s for s, ix in enumerate(names)}
        feats = img_feat_net([self.mapping_from_ix_to_node[ix] for ix in names]).reshape(len(names), -1).float()

        def codes(a):
            a = np.asarray(a, dtype=np.int64)
            out = a.copy(); img = a >= N
            out[img] = [-1 - slot[int(i)] for i in a[img]]
            return torch.from_numpy(out.astype(np.int32)).to(dev)
        loss, e_pos, e_neg = ops.JointLossFn.apply(m.embeddings.weight, feats, codes(fa), codes(ta), codes(neg), None, 0.0,
                                                   self.alpha, _lib.ENERGY_ORDER, _lib.LABEL_RAW, _lib.IMAGE_RAW)
        with torch.no_grad():
            pf = m.embeddings.weight[torch.as_tensor(fa, dtype=torch.long, device=dev)]
            pt = feats[torch.as_tensor([slot[int(i)] for i in ta], dtype=torch.long, device=dev)]
        return pf, pt, loss, e_pos, e_neg.reshape(-1)


class EmbeddingLabelsWithImages:
    """order_embeddings_images.py:712-960, train path: Embedder + FeatNet over a feature table, Adam over both, one
    fused loss launch per step."""

    def __init__(self, graph_dict, labelmap, criterion, lr, batch_size, experiment_name, embedding_dim, neg_to_pos_ratio,
                 image_fc7, normalize, alpha, has_fixed_alpha, lr_step=[], experiment_dir='../exp/', n_epochs=10,
                 eval_interval=2, feature_extracting=True, use_pretrained=True, load_wt=False, model_name=None,
                 optimizer_method='adam', use_grayscale=False):
        torch.manual_seed(0)
        if not torch.cuda.is_available():
            raise RuntimeError('EmbeddingLabelsWithImages runs on the MI355X only (no CPU fallback)')
        self.device = torch.device('cuda', torch.cuda.current_device())
        self.labelmap, self.criterion, self.lr, self.batch_size = labelmap, criterion, lr, batch_size
        self.embedding_dim, self.neg_to_pos_ratio, self.normalize = embedding_dim, neg_to_pos_ratio, normalize
        self.optimal_threshold, self.has_fixed_alpha = alpha, has_fixed_alpha
        self.graph_dict = graph_dict
        self.image_fc7 = image_fc7                                              # {image name: 2048-d feature}
        self.model = Embedder(embedding_dim=embedding_dim, labelmap=labelmap).to(self.device)
        self.feat_net = FeatNet(output_dim=embedding_dim, normalize=normalize).to(self.device)
        self.criterion.set_graph_tc(graph_dict['G_train_tc'])
        self.optimizer = torch.optim.Adam(list(self.model.parameters()) + list(self.feat_net.parameters()), lr=lr)

    def img_feat_net(self, names):
        x = torch.tensor(np.stack([np.asarray(self.image_fc7[n], dtype=np.float32) for n in names]), device=self.device)
        return self.feat_net(x)

    def train_step(self, inputs_from, inputs_to, status=None):
        self.optimizer.zero_grad()
        _, _, loss, e_pos, e_neg = self.criterion(self.model, self.img_feat_net, inputs_from, inputs_to, status, 'train')
        loss.backward()
        self.optimizer.step()
        return loss.detach(), e_pos, e_neg
