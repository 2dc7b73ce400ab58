#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace run of `bench.py` (rocpd sqlite output, ROCm 7.2's default format) into per-step
steady-state numbers.  Steps are delimited by the fused cone-loss kernel (one launch per step at the bench grid).
usage: summarize_rocpd.py <results.db> [--steps 3] [--skip-last 3] [--grid 16384] > profiles/rNN_..._steady_state.md"""
import argparse, collections, re, sqlite3, sys

ap = argparse.ArgumentParser(); ap.add_argument('db'); ap.add_argument('--steps', type=int, default=3)
ap.add_argument('--marker', default='joint_loss_kernel'); ap.add_argument('--skip-last', type=int, default=3)
ap.add_argument('--grid', type=int, default=0, help='grid_x of the bench-step launch of the marker kernel (0: any)')
ap.add_argument('--top', type=int, default=45)
a = ap.parse_args()
c = sqlite3.connect(a.db)
rows = c.execute('select name, start, end, grid_x, vgpr_count, lds_size from kernels order by start').fetchall()
marks = [r[1] for r in rows if a.marker in r[0] and (a.grid == 0 or r[3] == a.grid)]
if a.skip_last:
    marks = marks[:-a.skip_last]
n = a.steps
lo, hi = marks[-n - 1], marks[-1]
sel = [r for r in rows if lo <= r[1] < hi]


def short(nm):
    nm = re.sub(r'\[clone .*', '', nm)
    if nm.startswith('void '):
        nm = nm[5:]
    return re.sub(r'\(.*', '', nm)[:110]


def cat(k):
    if 'lec::' in k: return 'liblecone (this repo)'
    if 'BatchNorm' in k or 'batch_norm' in k: return 'batchnorm (library)'
    if ('igemm' in k or 'conv' in k.lower() or 'gemm' in k.lower() or k.startswith('Cijk') or '2ck' in k or 'ck::' in k or 'Im2' in k or 'im2' in k
            or 'Col2' in k or 'wrw' in k.lower()): return 'conv/gemm (library)'
    if 'elementwise' in k or 'SubTensor' in k or 'fillBuffer' in k or 'copyBuffer' in k or 'Fill' in k or 'transpose' in k.lower(): return 'elementwise/copies (library)'
    return 'other'


agg = collections.defaultdict(lambda: [0, 0])
for nm, s, e, g, v, l in sel:
    k = short(nm); agg[k][0] += e - s; agg[k][1] += 1
tot = sum(v[0] for v in agg.values())
cats = collections.defaultdict(float)
for k, (d, cnt) in agg.items():
    cats[cat(k)] += d
print('# steady-state kernel time per step (%d steps, rocprofv3 --kernel-trace)\n' % n)
print('wall per step: %.3f ms; kernel-busy per step: %.3f ms; launches per step: %.0f\n' % ((hi - lo) / n / 1e6, tot / n / 1e6, len(sel) / n))
print('| category | ms/step | share |\n|---|---|---|')
for k, d in sorted(cats.items(), key=lambda kv: -kv[1]):
    print('| %s | %.3f | %.1f %% |' % (k, d / n / 1e6, 100 * d / tot))
print('\n| kernel | ms/step | share | launches/step | avg us |\n|---|---|---|---|---|')
for k, (d, cnt) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:a.top]:
    print('| `%s` | %.3f | %.1f %% | %.0f | %.1f |' % (k, d / n / 1e6, 100 * d / tot, cnt / n, d / cnt / 1e3))
