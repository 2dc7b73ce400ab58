#!/usr/bin/env python3
"""profiles/r05_step_traffic.{json,md} from the two summaries tools/step_traffic_round5.sh leaves in gpurun_out/r5prof/ (st_<counter>.json).
usage: python tools/make_step_traffic_round5.py <ms per step>"""
import collections, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(ROOT, 'gpurun_out', 'r5prof')


def fam(k):
    if 'lec::bn_' in k: return 'BatchNorm family (bn.hip)'
    if 'conv_f32_wgrad' in k: return 'convolution weight gradients'
    if 'conv_f32_act' in k: return 'convolution forward / data gradient'
    if 'lec::' in k: return 'other liblecone (loss, pooling, Adam, ...)'
    return 'library (index gathers, fills, fc GEMM)'


ms = float(sys.argv[1])
F = json.load(open(os.path.join(D, 'st_FETCH_SIZE.json'))); W = json.load(open(os.path.join(D, 'st_WRITE_SIZE.json')))
f = collections.defaultdict(float); w = collections.defaultdict(float)
for k, v in F['bytes_per_step'].items(): f[fam(k)] += v
for k, v in W['bytes_per_step'].items(): w[fam(k)] += v
native = {k: (2 * f[k] / 1e9, w[k] / 1e9) for k in set(f) | set(w)}
CS = os.path.join(ROOT, 'learning_embeddings_amd', 'csrc')
out = {'native': native, 'ms_per_step': ms,
       'kernel_sources_sha256': {x: hashlib.sha256(open(os.path.join(CS, x), 'rb').read()).hexdigest() for x in ('conv_f32.hip', 'conv_f32_act_body.inc', 'conv_geo.h', 'bn.hip')}}
json.dump(out, open(os.path.join(ROOT, 'profiles', 'r05_step_traffic.json'), 'w'), indent=1)
md = ['# HBM traffic of the whole fp32 bench step, all kernels (rocprofv3 PMC, round 5, MI355X)', '',
      '`bash tools/step_traffic_round5.sh` then `python tools/make_step_traffic_round5.py <ms per step>`: `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes) over',
      '`python3 bench.py --steps 2 --warmup 1 --launch eager --secondary none --no-cpu-baseline --no-stress --through-trainer 0` (two concurrent half-batch passes, BatchNorm backward',
      'partly inside the convolutions), every kernel of the run summed per step (`tools/summarize_pmc.py --prefix ""`).  FETCH_SIZE counts half of the bytes of wide (16 B / lane) streaming',
      'reads on gfx950 (MI355X_MICROARCH.md, HBM section): the read column is FETCH_SIZE x 2.', '',
      '| kernel family | read GB (FETCH_SIZE x 2) | written GB (WRITE_SIZE) | round 4 read / written |', '|---|---|---|---|']
try:
    r2 = json.load(open(os.path.join(ROOT, 'profiles', 'r04_step_traffic.json')))['native']
except Exception:
    r2 = {}
for k, (r, wv) in sorted(native.items(), key=lambda kv: -sum(kv[1])):
    o = r2.get(k)
    md.append('| %s | %.1f | %.1f | %s |' % (k, r, wv, '%.1f / %.1f' % tuple(o) if o else '-'))
tr = sum(v[0] for v in native.values()); tw = sum(v[1] for v in native.values())
md += ['| **total** | %.1f | %.1f | %s |' % (tr, tw, '%.1f / %.1f' % (sum(v[0] for v in r2.values()), sum(v[1] for v in r2.values())) if r2 else '-'), '']
bn = sum(native.get('BatchNorm family (bn.hip)', (0, 0)))
md += ['Reading.  The BatchNorm family moves %.0f GB per step (round 4: see the last column): pass 1 of the foldable BatchNorm backwards now runs in the epilogue of the data gradient that produces' % bn,
       'its gradient and pass 2 of the large bn3 layers on the operand load of conv3\'s two gradient kernels, whose reads show up in the convolution rows instead.  Whole step: %.0f GB in %.0f ms' % (tr + tw, ms),
       '= %.1f TB/s: the fp32 step is bound by the matrix pipe (DESIGN.md section 6), not by HBM bandwidth.' % ((tr + tw) / ms)]
open(os.path.join(ROOT, 'profiles', 'r05_step_traffic.md'), 'w').write('\n'.join(md) + '\n')
print('\n'.join(md[-8:]))
