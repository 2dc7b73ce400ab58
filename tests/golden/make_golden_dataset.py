#!/usr/bin/env python3
"""Generate fixture F11 (the pair dataset and the graph build) by IMPORTING the reference -- build container only.

    python tests/golden/make_golden_dataset.py        # rewrites tests/golden/F11_pair_dataset.json

Reference entry points exercised (file:line under /root/reference):
  network/oe_h.py:506-580   create_combined_graphs: label graph, per-split (label, image) graphs, transitive closure of the
                            train graph, node <-> index mapping (labels keep their id, images numbered in node-iteration order),
                            dense negative adjacency A = 1 - TC - I
  network/oe_h.py:583-736   ETHECHierarchyWithImages: edge order, __len__/__getitem__ (plain and half_half with map_ranges),
                            set_levels_to_hide filtering
The inputs are synthetic (a [2, 4, 8] tree, 12 / 4 / 4 images); the outputs are the reference's own return values.
nx.write_gpickle no longer exists in networkx 3 (oe_h.py:565-571 calls it only to cache the graphs on disk): it is replaced by a
no-op for the call, and the function runs in a scratch directory because it also np.save()s the adjacency.
"""
import json, os, sys, tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, SynthLabelMap  # noqa: E402


def loaders(lm, counts, bs=5):
    """Imageless dataloaders: lists of batches {'image','labels','level_labels' [b, L] tensor, 'image_filename' [b]}."""
    import torch
    L = len(lm.levels)
    out = {}
    k = 0
    for split, n in counts.items():
        recs = []
        for j in range(n):
            leaf = (k * 5 + 3) % lm.levels[-1]; k += 1          # scattered leaves; several images share a leaf, some leaves have none
            chain = [leaf]
            for l in range(L - 1, 0, -1):
                chain.append((chain[-1] * lm.levels[l - 1]) // lm.levels[l])
            recs.append(('%s_img_%02d.jpg' % (split, j), chain[::-1]))
        batches = []
        for i in range(0, n, bs):
            part = recs[i:i + bs]
            batches.append({'image': None, 'labels': None, 'level_labels': torch.tensor([c for _, c in part], dtype=torch.long),
                            'image_filename': [f for f, _ in part]})
        out[split] = batches
    return out


def main():
    import networkx as nx
    mods = import_reference()
    oe_h = mods['oe_h']
    lm = SynthLabelMap([2, 4, 8])
    dl = loaders(lm, {'train': 12, 'val': 4, 'test': 4})
    nx.write_gpickle = lambda *a, **k: None
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            gd = oe_h.create_combined_graphs(dl, lm)
        finally:
            os.chdir(cwd)
    n_nodes = len(gd['mapping_ix_to_node'])
    enc = lambda e: [e[0], e[1]]
    fx = {'levels': lm.levels,
          'loaders': {s: [{'level_labels': b['level_labels'].tolist(), 'image_filename': b['image_filename']} for b in bl] for s, bl in dl.items()},
          'mapping_ix_to_node': [gd['mapping_ix_to_node'][i] for i in range(n_nodes)],
          'graph_edges': [enc(e) for e in gd['graph'].edges()],
          'graph_tc_edges': sorted(enc(e) for e in gd['graph_tc'].edges()),
          'G_train_edges': [enc(e) for e in gd['G_train'].edges()],
          'G_val_edges': [enc(e) for e in gd['G_val'].edges()],
          'G_test_edges': [enc(e) for e in gd['G_test'].edges()],
          'G_train_skeleton_full_edges': [enc(e) for e in gd['G_train_skeleton_full'].edges()],
          'G_train_tc_nodes': list(gd['G_train_tc'].nodes()), 'graph_nodes': list(gd['graph'].nodes()),
          'G_train_tc_edges': [enc(e) for e in gd['G_train_tc'].edges()],
          'neg_adjacency': np.asarray(gd['G_train_neg']).astype(int).tolist()}
    ds = {}
    for hh in (False, True):
        d = oe_h.ETHECHierarchyWithImages(gd['G_train_tc'], lm, imageless_dataloaders=None, half_half=hh)
        for hide in ([], [1], [0, 2], [0, 1, 2]):
            if hide or hh:
                d.set_levels_to_hide(hide)
            rec = {'len': len(d)}
            if hh:
                rec['edge_list_ll'] = [enc(e) for e in d.edge_list_ll]; rec['edge_list_li'] = [enc(e) for e in d.edge_list_li]
            else:
                rec['edge_list'] = [enc(e) for e in d.edge_list]
            items = []
            for i in range(len(d)):
                try:
                    it = d[i]
                    items.append([it['original_from'], it['original_to'], it['status']])
                except IndexError:
                    items.append('IndexError')
            rec['items'] = items
            ds['half_half=%s hide=%s' % (hh, hide)] = rec
    fx['dataset'] = ds
    with open(os.path.join(HERE, 'F11_pair_dataset.json'), 'w') as f:
        json.dump(fx, f, separators=(',', ':'))
    print('F11 written: %d nodes, %d TC edges, %d dataset variants' % (n_nodes, len(fx['G_train_tc_edges']), len(ds)))


if __name__ == '__main__':
    main()
