#!/usr/bin/env python3
"""profiles/<name>.{json,md} from the two summaries tools/step_traffic.sh leaves in gpurun_out/traffic/ (<tag>_<counter>.json).
usage: python tools/make_step_traffic.py <tag> <ms per step> <name> ["title"] [f32|bf16]
The JSON is keyed to the sha256 of the kernel sources the profiled step launches from (f32: conv_f32*.hip; bf16: conv_bf16.hip + conv_mfma.hip; both: conv_geo.h, bn.hip):
bench.py reports roofline.traffic = null once one of them has changed."""
import collections, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(ROOT, 'gpurun_out', 'traffic')


def fam(k):
    if 'lec::bn_' in k: return 'BatchNorm family (bn.hip)'
    if 'wgrad' in k: return 'convolution weight gradients'
    if 'conv' in k and 'lec::' in k: return 'convolution forward / data gradient'
    if 'lec::' in k: return 'other liblecone (loss, pooling, Adam, ...)'
    return 'library / framework (index gathers, fills, fc GEMM)'


tag, ms, name = sys.argv[1], float(sys.argv[2]), sys.argv[3]
title = sys.argv[4] if len(sys.argv) > 4 else tag
kind = sys.argv[5] if len(sys.argv) > 5 else ('bf16' if 'bf16' in name or 'cfg5' in name else 'f32')
SRCS = ('conv_f32.hip', 'conv_f32_act_body.inc', 'conv_f32x3.hip', 'conv_stem_f32.hip', 'conv_geo.h', 'bn.hip') if kind == 'f32' else ('conv_bf16.hip', 'conv_mfma.hip', 'conv_geo.h', 'bn.hip')
F = json.load(open(os.path.join(D, tag + '_FETCH_SIZE.json'))); W = json.load(open(os.path.join(D, tag + '_WRITE_SIZE.json')))
f = collections.defaultdict(float); w = collections.defaultdict(float)
for k, v in F['bytes_per_step'].items(): f[fam(k)] += v
for k, v in W['bytes_per_step'].items(): w[fam(k)] += v
native = {k: (2 * f[k] / 1e9, w[k] / 1e9) for k in set(f) | set(w)}
tr = sum(v[0] for v in native.values()); tw = sum(v[1] for v in native.values())
CS = os.path.join(ROOT, 'learning_embeddings_amd', 'csrc')
out = {'native': native, 'ms_per_step': ms, 'read_gb': tr, 'written_gb': tw, 'bytes_per_step': (tr + tw) * 1e9,
       'kernel_sources_sha256': {x: hashlib.sha256(open(os.path.join(CS, x), 'rb').read()).hexdigest() for x in SRCS}}
json.dump(out, open(os.path.join(ROOT, 'profiles', name + '.json'), 'w'), indent=1)
md = ['# HBM traffic of one step by kernel family: %s (rocprofv3 PMC, one MI355X)' % title, '',
      '`bash tools/step_traffic.sh %s ...` then `python tools/make_step_traffic.py %s %s %s`: `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes) over eager launches of' % (tag, tag, sys.argv[2], name),
      '`bench.py`, every kernel of the run summed per step (`tools/summarize_pmc.py --prefix ""`).  FETCH_SIZE counts half of the bytes of wide (16 B / lane) streaming reads on gfx950',
      '(MI355X_MICROARCH.md, HBM section): the read column is FETCH_SIZE x 2; Infinity-Cache hits are counted, so a family that re-reads through L2 misses shows more than it takes from HBM.', '',
      '| kernel family | read GB (FETCH_SIZE x 2) | written GB (WRITE_SIZE) |', '|---|---|---|']
for k, (r, wv) in sorted(native.items(), key=lambda kv: -sum(kv[1])):
    md.append('| %s | %.1f | %.1f |' % (k, r, wv))
md += ['| **total** | %.1f | %.1f |' % (tr, tw), '', 'Whole step: %.0f GB in %.1f ms = %.2f TB/s.' % (tr + tw, ms, (tr + tw) / ms)]
open(os.path.join(ROOT, 'profiles', name + '.md'), 'w').write('\n'.join(md) + '\n')
print('\n'.join(md[-10:]))
