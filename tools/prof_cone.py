#!/usr/bin/env python3
"""Run the fused loss kernel a few times at one size (for rocprofv3 --pmc / --kernel-trace passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_cone
B, K, D, N = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (4096, 256, 10, 50000)))
r = bench_cone.time_joint(B, K, D, N, B, iters=5)
print(r)
