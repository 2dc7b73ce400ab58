"""Host-side mirror of `network/oe.py`: the EUCLIDEAN entailment-cone sibling of the joint image+label trainer
(SURVEY.md 8f rank 4).  Same pipeline as oe_h.py -- one CNN pass over the step's images, the bit-exact negative sampler,
ONE fused loss kernel, Adam -- with three differences, all taken from the reference:

  * energy: oe.py:721-739 (cone half-aperture and angle compared in cosine space, K = 3.0 by default);
  * both label rows and CNN outputs become points through soft_clip  x/|x| * (|x| + K)  (oe.py:75-80, :235-240):
    every point lies outside the radius-K ball around the origin, where the cones are defined;
  * the optimizer step is plain Adam over table + CNN (oe.py:1519-1520): no Riemannian rescale, no table clip.

Everything else (dataset, graphs, metrics, checkpoints, data parallelism) is inherited from the oe_h mirror.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops
from .oe_h import (FeatCNN18 as _HypFeatCNN18, FeatCNN as _HypFeatCNN, _JointCriterionBase, _unwrap, my_collate,  # noqa: F401
                   OrderEmbeddingWithImagesHypernymLoss)
from .oe_h_trainer import (DiGraph, transitive_closure, create_combined_graphs, ETHECHierarchyWithImages,  # noqa: F401
                           EmbeddingMetrics, GlobalBatchSampler, JointEmbeddings as _HypJointEmbeddings)


class Embedder(nn.Module):
    """oe.py:51-80: N(0,1)-initialised table; forward = gather, then soft_clip when K is set."""

    def __init__(self, embedding_dim, labelmap, normalize, K=None):
        super().__init__()
        self.labelmap = labelmap
        self.embedding_dim = embedding_dim
        self.normalize = normalize
        self.K = K
        if self.normalize == 'max_norm':
            self.embeddings = nn.Embedding(self.labelmap.n_classes, self.embedding_dim, max_norm=1.0)
        else:
            self.embeddings = nn.Embedding(self.labelmap.n_classes, self.embedding_dim)
        print('Embeds {} objects'.format(self.labelmap.n_classes))

    @property
    def device(self):
        return self.embeddings.weight.device

    def forward(self, inputs):
        w = self.embeddings.weight
        if self.normalize == 'unit_norm':
            return F.normalize(F.embedding(inputs, w), p=2, dim=1)
        if not self.K:
            return F.embedding(inputs, w)
        shp = inputs.shape
        out = ops.LabelProjectFn.apply(w, inputs.reshape(-1), self.K, _lib.LABEL_SOFTCLIP_K)
        return out.view(*shp, self.embedding_dim)

    def soft_clip(self, x):
        return ops.ImageSoftClipFn.apply(x, self.K, _lib.IMAGE_SOFTCLIP_K)


class FeatCNN18(_HypFeatCNN18):
    """oe.py:194-240: ResNet-18 -> Linear(512, D) -> soft_clip with additive constant K."""
    def _setup(self, K):
        self.K = K
        self.inner_radius = None

    def soft_clip(self, x):
        return ops.ImageSoftClipFn.apply(x, self.K, _lib.IMAGE_SOFTCLIP_K)


class FeatCNN(FeatCNN18):
    """oe.py:243-289: the ResNet-50 wrapper."""

    def __init__(self, image_dir, path_to_exp='../exp', input_dim=2048, output_dim=10,
                 exp_name='ethec_resnet50_lr_1e-5_1_1_1_1/', K=None, weights=None, compute_dtype=torch.float32,
                 channels_last=True):
        super().__init__(image_dir, path_to_exp, input_dim, output_dim, exp_name, K, weights, compute_dtype,
                         channels_last, arch='resnet50')


class FeatNet(nn.Module):
    """oe.py:83-138: Linear(input_dim, D) on precomputed image features, then the same soft_clip."""

    def __init__(self, normalize, input_dim=2048, output_dim=10, K=None):
        super().__init__()
        self.output_dim = output_dim
        self.normalize = normalize
        self.K = K
        self.fc1 = nn.Linear(input_dim, output_dim)

    def forward(self, x):
        shp = x.shape
        y = self.fc1(x).reshape(-1, self.output_dim).float()
        if self.normalize is not None:
            raise NotImplementedError('FeatNet normalize modes are outside the hot path')
        if self.K:
            y = ops.ImageSoftClipFn.apply(y, self.K, _lib.IMAGE_SOFTCLIP_K)
        return y.view(*shp[:-1], self.output_dim)


class EuclideanConesWithImagesHypernymLoss(_JointCriterionBase):
    """oe.py:650-965 -- the Euclidean entailment-cone criterion (K = 3.0)."""
    energy = 'euc_cone'
    default_image_proj = _lib.IMAGE_SOFTCLIP_K

    def __init__(self, labelmap, neg_to_pos_ratio, feature_dict, alpha, pick_per_level=False, K=3.0, use_CNN=False):
        print('Using Euclidean cones loss!')
        self._init_common(labelmap, neg_to_pos_ratio, feature_dict, alpha, pick_per_level, use_CNN)
        self.K = K
        self.epsilon = 1e-5

    def _proj_flags(self, model):
        return _lib.LABEL_SOFTCLIP_K if getattr(_unwrap(model), 'K', None) else _lib.LABEL_RAW

    def E_operator(self, x, y):
        return ops.pair_energy(x, y, self.K, 'euc_cone')


class JointEmbeddings(_HypJointEmbeddings):
    """oe.py:1224-1991: same trainer surface; plain optimizer step (oe.py:1519-1520)."""
    riemannian_table_step = False

    @staticmethod
    def _model_classes():
        return Embedder, FeatCNN18, FeatCNN, FeatNet
