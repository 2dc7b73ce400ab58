#!/usr/bin/env python3
"""Generate fixture F13: ONE whole train step of the reference's joint trainer with a BatchNorm-bearing CNN, by IMPORTING the reference.

Runs only in the build container (where /root/reference exists); the GPU box never runs it.  What is exercised, in the reference's own
code (file:line under /root/reference/network):
  oe_h.py:904-967    EuclideanConesWithImagesHypernymLoss.forward, train branch: calculate_from_and_to_emb for the positives
                     (:969-1016: the batch's own image tensors, ONE CNN forward per side), 2K negatives per positive
                     (sample_negative_edge :849-902, python `random` seeded 0), calculate_from_and_to_emb for the 2K B negative pairs
                     (image ends by name through dataloader.get_image: every FIXED image end is embedded K more times, unflipped),
                     positive_pair / negative_pair / get_image_label_loss (:835-847)
  oe_h.py:1766-1771  loss.backward(); W.grad *= (1/lambda_x(W))^2; ONE Adam over list(model.parameters()) + list(img_feat_net.parameters())
                     at lr (:1523); W = soft_clip(W)
The CNN is a stand-in the reference never defines (torchvision is not installed): a ResNet-10 of width 8 written here with stock torch.nn
modules and torchvision's parameter names, followed by the reference's own FeatCNN18.soft_clip (:323-328) -- small enough for a fixture,
with everything that makes "identical inputs" subtle: BatchNorm in train mode, four forwards per step, duplicated rows inside a batch.

    python tests/golden/make_golden_step.py        # rewrites tests/golden/F13_train_step.npz
"""
import os, random, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, SynthLabelMap, build_joint_graph, label_parents   # noqa: E402


def main():
    mods = import_reference()
    oe_h = mods['oe_h']
    import torch
    import torch.nn as nn
    torch.set_num_threads(1)                                    # one summation order
    t2n = lambda t: t.detach().cpu().numpy().copy()
    Kc, D, Kneg, alpha, lr, B, n_img, hw, width = 0.1, 10, 4, 0.05, 1e-3, 10, 20, 32, 8   # K = levels + 1: pass 3 draws IMAGES as negatives
    r_in = 2 * Kc / (1 + np.sqrt(1 + 4 * Kc * Kc))

    class Block(nn.Module):                                     # torchvision BasicBlock
        def __init__(self, cin, planes, stride):
            super().__init__()
            self.conv1 = nn.Conv2d(cin, planes, 3, stride, 1, bias=False); self.bn1 = nn.BatchNorm2d(planes)
            self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False); self.bn2 = nn.BatchNorm2d(planes)
            self.downsample = None
            if stride != 1 or cin != planes:
                self.downsample = nn.Sequential(nn.Conv2d(cin, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        def forward(self, x):
            idt = x if self.downsample is None else self.downsample(x)
            out = torch.relu(self.bn1(self.conv1(x)))
            return torch.relu(self.bn2(self.conv2(out)) + idt)

    class ResNet10(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = nn.Conv2d(3, width, 7, 2, 3, bias=False); self.bn1 = nn.BatchNorm2d(width)
            self.maxpool = nn.MaxPool2d(3, 2, 1)
            self.layer1 = nn.Sequential(Block(width, width, 1)); self.layer2 = nn.Sequential(Block(width, 2 * width, 2))
            self.layer3 = nn.Sequential(Block(2 * width, 4 * width, 2)); self.layer4 = nn.Sequential(Block(4 * width, 8 * width, 2))
            self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
            self.fc = nn.Linear(8 * width, D)
            for m in self.modules():                            # torchvision's init
                if isinstance(m, nn.Conv2d):
                    nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        def forward(self, x):
            x = self.maxpool(torch.relu(self.bn1(self.conv1(x))))
            x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
            return self.fc(torch.flatten(self.avgpool(x), 1))

    class Feat(nn.Module):                                      # FeatCNN18's forward (oe_h.py:317-321) around the stand-in backbone
        inner_radius = r_in
        def __init__(self):
            super().__init__(); self.model = ResNet10()
        def forward(self, x):
            return oe_h.FeatCNN18.soft_clip(self, self.model(x))

    lmap = SynthLabelMap([2, 4, 8])
    N, names, A, n2i, i2n = build_joint_graph(lmap.levels, lmap.edges, n_img)
    torch.manual_seed(0)
    model = oe_h.Embedder(D, lmap, None, K=Kc)
    net = Feat(); net.train(); model.train()
    sd0 = {k: t2n(v) for k, v in net.state_dict().items()}
    W0 = t2n(model.embeddings.weight)
    rs = np.random.RandomState(13)
    u8 = rs.randint(0, 256, size=(n_img, hw, hw, 3)).astype(np.uint8)            # resized uint8 images, HWC
    imgs = torch.from_numpy(u8).permute(0, 3, 1, 2).contiguous().float().div(255)   # ToTensor

    class DL:
        def get_image(self, fname):                             # oe_h.py:668-677: the val/test transform (no flip)
            return imgs[n2i[fname] - N]

    crit = oe_h.EuclideanConesWithImagesHypernymLoss(lmap, Kneg, {}, alpha, True, K=Kc, use_CNN=True)
    crit.set_negative_graph(A, n2i, i2n); crit.set_dataloader(DL())
    par = label_parents(lmap.levels, lmap.edges)
    leaf_start = N - lmap.levels[-1]
    of, ot, flips = [], [], []
    for b in range(B):
        if b % 5 == 4:                                          # a (label, label) edge
            v = int(rs.randint(lmap.level_start[1], N)); of.append(int(par[v][0])); ot.append(v); flips.append(0)
        else:                                                   # (label, image): the image under its leaf or an ancestor; two positives share img 3
            j = 3 if b in (1, 6) else int(rs.randint(n_img))
            lab = leaf_start + (j % lmap.levels[-1])
            for _ in range(b % len(lmap.levels)):
                if lab in par:
                    lab = par[lab][0]
            of.append(int(lab)); ot.append(names[j]); flips.append(int(rs.randint(2)))
    inputs_from = list(of)
    inputs_to = [(imgs[n2i[t] - N].flip(-1) if f else imgs[n2i[t] - N]) if isinstance(t, str) else t for t, f in zip(ot, flips)]
    opt = torch.optim.Adam([{'params': list(model.parameters()) + list(net.parameters())}], lr=lr)      # oe_h.py:1523
    opt.zero_grad()
    random.seed(0)
    loss, e_pos, e_neg = crit(model, net, inputs_from, inputs_to, of, ot, torch.ones(B), 'train')
    loss.backward()
    gW_raw = t2n(model.embeddings.weight.grad)
    grads = {k: t2n(p.grad) for k, p in net.named_parameters()}

    class T:                                                    # the trainer methods of oe_h.py:1604-1636 only touch these attributes
        embedding_dim = D
        class criterion: inner_radius = r_in
    T.soft_clip = oe_h.JointEmbeddings.soft_clip; T.lambda_x = oe_h.JointEmbeddings.lambda_x
    tr = T()
    W = model.embeddings.weight
    W.grad.data *= (1.0 / tr.lambda_x(W.data)) ** 2             # oe_h.py:1768
    opt.step()                                                  # :1769
    W.data = tr.soft_clip(W.data)                               # :1771
    sd1 = {k: t2n(v) for k, v in net.state_dict().items()}
    random.seed(0)                                              # the negative indices the call consumed (same stream, same order; :940-957)
    neg = np.zeros((B, 2 * Kneg), dtype=np.int64)
    for b in range(B):
        for p in range(Kneg):
            neg[b, p] = crit.sample_negative_edge(u=of[b], v=None, level_id=p)
            neg[b, p + Kneg] = crit.sample_negative_edge(u=None, v=ot[b], level_id=p)
    out = {'levels': np.array(lmap.levels), 'edges': np.array(sorted(lmap.edges)), 'n_images': np.int64(n_img), 'images_u8': u8,
           'from': np.array([n2i[x] for x in of]), 'to': np.array([n2i[x] for x in ot]), 'flips': np.array(flips), 'neg': neg,
           'W0': W0, 'W1': t2n(W), 'gW_raw': gW_raw, 'loss': t2n(loss), 'e_pos': t2n(e_pos), 'e_neg': t2n(e_neg),
           'K': np.float64(Kc), 'alpha': np.float64(alpha), 'lr': np.float64(lr), 'Kneg': np.int64(Kneg), 'D': np.int64(D), 'width': np.int64(width)}
    for k, v in sd0.items():
        out['sd0/' + k] = v
    for k, v in sd1.items():
        out['sd1/' + k] = v
    for k, v in grads.items():
        out['grad/' + k] = v
    np.savez_compressed(os.path.join(HERE, 'F13_train_step.npz'), **out)
    n_rows = [sum(1 for t in inputs_to if not isinstance(t, int))]
    print('F13 loss', float(loss), 'e_pos', tuple(e_pos.shape), 'e_neg', tuple(e_neg.shape), 'positives with an image:', n_rows[0],
          'image negatives:', int((neg >= N).sum()), 'bn1 running_mean[:3]', sd1['model.bn1.running_mean'][:3])


if __name__ == '__main__':
    main()
