#!/usr/bin/env python3
"""Sweep (T, EPL, iters) geometries of the fused loss kernel through the LEC_JOINT_GEOM tuning hook."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_cone

SIZES = [(256, 5, 10, 2000), (256, 256, 10, 50000), (4096, 256, 10, 50000), (256, 256, 128, 50000), (4096, 64, 128, 50000)]
for B, K, D, N in SIZES:
    cands = [(t, e) for t, e in [(1, 12), (1, 16), (2, 8), (4, 4), (4, 8), (8, 8), (16, 8), (32, 8), (64, 4), (64, 8)] if t * e >= D]
    rows = []
    for t, e in cands:
        for it in (0, 2, 8):
            for st in ((0, 1) if t == 1 else (0,)):
                os.environ['LEC_JOINT_GEOM'] = '%d,%d,%d' % (t, e, it); os.environ['LEC_JOINT_STAGE'] = str(st)
                r = bench_cone.time_joint(B, K, D, N, B, iters=20)
                rows.append((r['us'], t, e, it, st, r['GBps']))
    rows.sort()
    print('B=%d K=%d D=%d:' % (B, K, D), ' | '.join('T%d,E%d,it%d,lds%d %.1fus %.0fGB/s' % (t, e, it, st, us, gb) for us, t, e, it, st, gb in rows[:7]), ' ... worst %.1fus' % rows[-1][0], flush=True)
