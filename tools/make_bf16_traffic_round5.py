#!/usr/bin/env python3
"""profiles/r05_bf16_step_traffic.md from tools/prof_round5.sh's outputs (gpurun_out/r5prof): HBM bytes per kernel family of the bf16 step (the disclosed
secondary of the bench line = the inner loop of config 5's chunked step) next to the families' kernel time.  usage: python tools/make_bf16_traffic_round5.py"""
import collections, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(ROOT, 'gpurun_out', 'r5prof')


def fam(k):
    if 'lec::bn_' in k: return 'BatchNorm family (bn.hip, own)'
    if 'lec::wgrad' in k: return 'weight gradients, own (conv_mfma.hip)'
    if 'lec::conv' in k: return 'forward / data gradient, own (conv_mfma.hip)'
    if 'lec::' in k: return 'other own kernels (loss, pooling, Adam, ...)'
    if 'igemm_wrw' in k or 'wrw' in k.lower(): return 'weight gradients, library (MIOpen igemm)'
    if 'igemm' in k or 'ck' in k or 'Cijk' in k: return 'forward / data gradient, library (MIOpen igemm / CK / hipBLASLt)'
    return 'framework elementwise / copies'


F = json.load(open(os.path.join(D, 'bf16_FETCH_SIZE.json')))['bytes_per_step']; W = json.load(open(os.path.join(D, 'bf16_WRITE_SIZE.json')))['bytes_per_step']
f = collections.defaultdict(float); w = collections.defaultdict(float)
for k, v in F.items(): f[fam(k)] += v
for k, v in W.items(): w[fam(k)] += v
# kernel time per family from the steady-state table of the same script
t = collections.defaultdict(float)
for line in open(os.path.join(D, 'r05_bench_cfg3_bf16_steady_state.md')):
    m = re.match(r'\| `(.+?)` \| ([0-9.]+) \|', line)
    if m:
        t[fam(m.group(1))] += float(m.group(2))
b = json.load(open(os.path.join(D, 'cfg3_bf16.json')))
ms = b['ms_per_step']
md = ['# HBM traffic of the bf16 step by kernel family (rocprofv3 PMC, round 5, one MI355X)', '',
      '`bash tools/prof_round5.sh` then `python tools/make_bf16_traffic_round5.py`: `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) over',
      '`python3 bench.py --dtype bf16 --steps 2 --warmup 1 --launch eager ...` (cfg3: 512 CNN rows of ResNet-50 per step, one pass, weight gradients on a side stream), every kernel',
      'summed per step; read = FETCH_SIZE x 2 (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md), written = WRITE_SIZE.  Kernel time: the kernel trace of the same script',
      '(`profiles/r05_bench_cfg3_bf16_steady_state.md`; sums of launch durations, two streams).  This step is the disclosed `secondary_bf16` of the bench line and the inner loop of',
      'config 5 (15 chunks of it per step).  Last measured in round 2 (VERDICT r04 weak #10).', '',
      '| kernel family | read GB | written GB | kernel ms per step (sum of durations) | GB / ms = TB/s while running |', '|---|---|---|---|---|']
tot_r = tot_w = tot_t = 0.0
for k in sorted(set(f) | set(w) | set(t), key=lambda k: -(2 * f[k] + w[k])):
    r, wv = 2 * f[k] / 1e9, w[k] / 1e9
    tot_r += r; tot_w += wv; tot_t += t[k]
    md.append('| %s | %.2f | %.2f | %.2f | %s |' % (k, r, wv, t[k], '%.2f' % ((r + wv) / t[k]) if t[k] > 0 else '-'))
md += ['| **total** | %.2f | %.2f | %.2f | |' % (tot_r, tot_w, tot_t), '',
       'Step: %.2f ms under the profiler (eager launches; the un-profiled graph-replayed step is in the bench line) = %.1f GB in %.1f ms = **%.2f TB/s** of the ~6.3 TB/s a streaming kernel reaches;'
       % (ms, tot_r + tot_w, ms, (tot_r + tot_w) / ms),
       'the analytic ResNet-50 flops of the step (12.5 TFLOP) over the step are %.1f %% of the dense bf16 matrix peak.  Reading: the bf16 step is neither matrix- nor purely bandwidth-bound as a whole --' % (12.515 / (ms * 1e-3) / 2500 * 100),
       'the BatchNorm family moves a third of the bytes (63 GB) at 3.8 TB/s while it runs (its passes share the chip with the side stream\'s weight gradients), the own 1x1 / 3x3-c64 convolutions',
       'stream at 5.5 TB/s, the library\'s kernels for the remaining shapes at 3.0; 583 launches of 10 - 700 us per step, a third of them BatchNorm finalize / statistics launches of 10 - 50 us whose',
       'boundaries nothing fills (one pass).  Where the bytes could still go: the 3x3 and strided layers the library serves (34 GB: an own bf16 implicit GEMM for them, as the fp32 path has), and the',
       'BatchNorm backward passes of the layers whose producer is a library kernel (no epilogue to fold into).', '']
open(os.path.join(ROOT, 'profiles', 'r05_bf16_step_traffic.md'), 'w').write('\n'.join(md) + '\n')
print('\n'.join(md[8:]))
