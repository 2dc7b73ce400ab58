#!/usr/bin/env python3
"""Time the fused scoring + per-level top-k kernel (lec_level_topk) against the unfused route it replaces
(lec_pair_energy_matrix + torch.topk per level) at the classification-metric sizes.

    python tools/bench_topk.py            # prints one JSON object per size
"""
import json, sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learning_embeddings_amd import ops  # noqa: E402


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    dev = 'cuda'
    for name, levels, M, D in (('ethec_train', [6, 21, 135, 561], 38000, 10), ('ethec_val', [6, 21, 135, 561], 4800, 10),
                               ('s5_50k', [2, 8, 32, 128, 512, 2048, 8192, 39078], 16384, 10),
                               ('ethec_D128', [6, 21, 135, 561], 38000, 128)):
        N = sum(levels)
        g = torch.Generator(device='cpu').manual_seed(0)
        lab = (torch.randn(N, D, generator=g) * 0.2).to(dev); img = (torch.randn(M, D, generator=g) * 0.3).to(dev)
        starts = np.concatenate([[0], np.cumsum(levels)])
        k = 5

        def fused():
            return ops.level_topk(lab, img, starts, k, 0.1)

        def unfused():
            E = ops.energy_matrix(lab, img, 0.1)
            return [torch.topk(E[:, starts[l]:starts[l + 1]], k=min(k, levels[l]), largest=False, dim=1) for l in range(len(levels))]
        t_f = timed(fused); t_u = timed(unfused)
        pairs = float(M) * N
        # algorithmic bytes of the fused kernel: both embedding tables once + the k indices / energies per image and level (the M x N matrix is
        # never written); what the UNFUSED route moves on top: the matrix written and read back
        alg = (N + M) * D * 4 + M * len(levels) * k * 8
        print(json.dumps({'size': name, 'M': M, 'N': N, 'D': D, 'fused_ms': round(t_f, 3), 'matrix_plus_torch_topk_ms': round(t_u, 3),
                          'pairs_per_s_fused': pairs / (t_f * 1e-3), 'speedup': round(t_u / t_f, 2),
                          'matrix_bytes_avoided_MB': round(pairs * 4 / 1e6, 1), 'alg_bytes': alg, 'achieved_GBps_on_alg_bytes': round(alg / (t_f * 1e-3) / 1e9, 2),
                          'GBps_if_the_matrix_were_written': round((alg + pairs * 4) / (t_f * 1e-3) / 1e9, 1),
                          'cone_energies_per_s': pairs / (t_f * 1e-3)}))


if __name__ == '__main__':
    main()
