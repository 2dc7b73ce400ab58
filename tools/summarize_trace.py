#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV of `bench.py` into per-step steady-state numbers.
Steps are delimited by the fused cone-loss kernel (one launch per step); the MIOpen find phase of the first steps
(naive_conv_* reference kernels, hundreds of ms each) is excluded by taking the last `--steps` steps before the
stress sweep.   usage: summarize_trace.py <kernel_trace.csv> [--steps 4] > profiles/rNN_steady_state.md"""
import argparse, collections, csv, re, sys

ap = argparse.ArgumentParser(); ap.add_argument('trace'); ap.add_argument('--steps', type=int, default=4)
ap.add_argument('--marker', default='joint_loss_kernel<'); ap.add_argument('--skip-last', type=int, default=3, help='trailing steps to ignore (bench.py appends 3 single-stream steps for the isolated BN measurement)')
ap.add_argument('--grid', type=int, default=16384, help='Grid_Size_X of the bench step launch of the marker kernel')
a = ap.parse_args()
rows = list(csv.DictReader(open(a.trace)))
# the stress sweep launches the same kernel with other grids: keep the bench-step launches only
step_marks = [int(r['Start_Timestamp']) for r in rows if a.marker in r['Kernel_Name'] and int(r['Grid_Size_X']) == a.grid]
if a.skip_last:
    step_marks = step_marks[:-a.skip_last]
lo, hi = step_marks[-a.steps - 1], step_marks[-1]
n = a.steps
sel = [r for r in rows if lo <= int(r['Start_Timestamp']) < hi]
agg = collections.defaultdict(lambda: [0, 0])
def short(nm):
    if nm.startswith('igemm') or nm.startswith('_ZN2ck') or nm.startswith('Cijk'):
        return nm[:64]
    return re.sub(r'\(.*', '', re.sub(r'<.*', '', nm))[:64]
for r in sel:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    k = short(r['Kernel_Name']); agg[k][0] += d; agg[k][1] += 1
tot = sum(v[0] for v in agg.values())
def cat(k):
    if 'lec::' in k: return 'liblecone (this repo)'
    if 'BatchNorm' in k: return 'batchnorm'
    if 'igemm' in k or 'conv' in k or 'gemm' in k.lower() or k.startswith('Cijk') or '2ck' in k or 'ck::' in k: return 'conv/gemm'
    if 'elementwise' in k or 'SubTensor' in k or 'fillBuffer' in k or 'copyBuffer' in k: return 'elementwise/copies'
    return 'other'
cats = collections.defaultdict(float)
for k, (d, c) in agg.items(): cats[cat(k)] += d
print('# steady-state kernel time per step (%d steps, rocprofv3 --kernel-trace)\n' % n)
print('wall per step: %.3f ms; kernel-busy per step: %.3f ms; launches per step: %.0f\n' % ((hi - lo) / n / 1e6, tot / n / 1e6, len(sel) / n))
print('| category | ms/step | share |\n|---|---|---|')
for k, d in sorted(cats.items(), key=lambda kv: -kv[1]):
    print('| %s | %.3f | %.1f%% |' % (k, d / n / 1e6, 100 * d / tot))
print('\n| kernel | ms/step | share | launches/step | avg us |\n|---|---|---|---|---|')
for k, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    print('| `%s` | %.3f | %.1f%% | %.1f | %.1f |' % (k, d / n / 1e6, 100 * d / tot, c / n, d / c / 1e3))
