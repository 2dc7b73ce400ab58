#!/usr/bin/env python3
"""2-rank JointEmbeddings run sharing one GPU over gloo (torchrun), for debugging the trainer's DP path."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('LEC_DIST_BACKEND', 'gloo')
import numpy as np, torch as t
from test_host_cpu import _fake_loaders
from learning_embeddings_amd import oe_h as m
from learning_embeddings_amd.hierarchy import SyntheticLabelMap as LM
lm = LM([2, 4, 8])
dl = _fake_loaders(lm, 32, 8)
for split in dl.values():
    for b in split:
        b['path_to_image'] = [t.rand(3, 32, 32, generator=t.Generator().manual_seed(int(n[4:]))) for n in b['image_filename']]
gd = m.create_combined_graphs(dl, lm, pick_per_level=True)
crit = m.EuclideanConesWithImagesHypernymLoss(lm, 5, {}, 0.05, True, K=0.1, use_CNN=True)
rank = int(os.environ.get('RANK', 0))
tr = m.JointEmbeddings(gd, dl, image_dir='', use_CNN=True, labelmap=lm, criterion=crit, lr=1e-3, n_workers=0,
                       batch_size=8, experiment_name='dp', embedding_dim=10, neg_to_pos_ratio=5, image_fc7=None,
                       normalize=None, alpha=0.05, experiment_dir='/tmp/lec_dp_r%d' % rank, n_epochs=1, eval_interval=5)
tr.pass_samples('train')
print('rank', rank, 'loss', tr.last_epoch_loss, 'table sum', float(tr.model.embeddings.weight.sum()), flush=True)
if t.distributed.is_initialized():
    t.distributed.barrier(); t.distributed.destroy_process_group()
