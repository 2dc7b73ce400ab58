#!/usr/bin/env python3
"""The bench workload (cfg3: ResNet-50, B = 256, K = 5, D = 10) trained for N steps from the same seeds in three convolution modes --
fp32 on the f32-input MFMA, fp32 with split products on the bf16 matrix cores, bf16 -- loss per window of 10 steps.
usage: python tools/loss_curve_modes.py [--steps 100]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning_embeddings_amd import resnet as R
from learning_embeddings_amd.engine import StepEngine
ap = argparse.ArgumentParser(); ap.add_argument('--steps', type=int, default=100); a = ap.parse_args()
out = {}
for tag, dtype, mode in (('f32-mfma', 'fp32', 'native'), ('f32-split', 'fp32', 'x3'), ('bf16', 'bf16', 'native')):
    R.F32_MODE = mode
    torch.manual_seed(0)
    eng = StepEngine('cfg3', n_images=4096, dtype=dtype, use_graph=False)
    losses = []
    for s in range(a.steps):
        l = eng.step()
        losses.append(float(l))
    torch.cuda.synchronize()
    eng.close(); del eng; torch.cuda.empty_cache()
    out[tag] = [round(sum(losses[i:i + 10]) / 10, 4) for i in range(0, a.steps, 10)]
    print(tag, out[tag], flush=True)
ref = out['f32-mfma']
for tag in ('f32-split', 'bf16'):
    print(tag, 'max relative deviation of a 10-step window from f32-mfma: %.2e' % max(abs(x - y) / abs(y) for x, y in zip(out[tag], ref)))
