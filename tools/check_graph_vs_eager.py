#!/usr/bin/env python3
"""The step's captured hipGraph against eager launches of the same region, parameter by parameter, after engines of other precisions
and launch modes have lived in the process (how the racing `w.grad.add_` of the stem's weight gradient was found in round 3: two
concurrent passes, only under replay, only after other engines had shifted the timing).  usage: python tools/check_graph_vs_eager.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learning_embeddings_amd.engine import StepEngine
for dtype, mode in (('bf16', False), ('bf16', True), ('fp32', False), ('fp32', True)):
    torch.manual_seed(0)
    eng = StepEngine('tiny', n_images=64, dtype=dtype, use_graph=mode, graph_after=2)
    for _ in range(6):
        eng.step()
    torch.cuda.synchronize()
    print(dtype, mode, 'graph', eng.hip_graph is not None, eng.graph_error, 'streams', [s.cuda_stream for s in eng.pass_streams], 'cur', torch.cuda.current_stream().cuda_stream, flush=True)
    if mode and dtype == 'fp32':
        eng.hip_graph.replay(); torch.cuda.synchronize()
        g1 = eng.arena.grad.clone()
        eng.hip_graph.replay(); torch.cuda.synchronize()
        g1b = eng.arena.grad.clone()
        eng._core(None); torch.cuda.synchronize()
        g2 = eng.arena.grad.clone()
        eng._core(None); torch.cuda.synchronize()
        g2b = eng.arena.grad.clone()
        print('replay vs replay', ((g1 - g1b).norm() / g1.norm()).item(), 'eager vs eager', ((g2 - g2b).norm() / g2.norm()).item(), 'replay vs eager', ((g1 - g2).norm() / g2.norm()).item())
        names = [n for n, p in eng.img_feat_net.named_parameters() if p.requires_grad]
        for k, (n, o) in enumerate(zip(names, eng.arena.offsets)):
            num = eng.arena.params[k].numel()
            a, b = g1[o:o + num], g2[o:o + num]
            r = ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
            if r > 1e-3:
                print('%-40s rel %.3f  |replay| %.4g |eager| %.4g ratio %.3f' % (n, r, a.norm().item(), b.norm().item(), (a.norm() / b.norm().clamp_min(1e-30)).item()))
    eng.close()
