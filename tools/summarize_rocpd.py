#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace run of `bench.py` (rocpd sqlite output, ROCm 7.2's default format) into per-step
steady-state numbers.  Steps are delimited by the fused cone-loss kernel (one launch per step at the bench grid).
usage: summarize_rocpd.py <results.db> [--steps 3] [--skip-last 3] [--grid 16384] > profiles/rNN_..._steady_state.md"""
import argparse, collections, re, sqlite3, sys

ap = argparse.ArgumentParser(); ap.add_argument('db'); ap.add_argument('--steps', type=int, default=3)
ap.add_argument('--marker', default='joint_loss_kernel'); ap.add_argument('--skip-last', type=int, default=3)
ap.add_argument('--grid', type=int, default=0, help='grid_x of the bench-step launch of the marker kernel (0: any)')
ap.add_argument('--top', type=int, default=45)
ap.add_argument('--conv-tflop-per-step', type=float, default=0.0, help='algorithmic convolution TFLOP of one step (bench.py: roofline.alg_flops_per_step / 1e12): prints the family\'s roofline fraction')
ap.add_argument('--peak-tflops', type=float, default=157.3)
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



# ---- kernel FAMILIES: sum of launch durations AND the union of their launch intervals (launches of a family overlap when the step runs as
# concurrent passes / streams: the sum then counts shared time once per stream, the union is the time the family had the GPU at all)
def family(k):
    if 'conv_f32' in k or 'conv_bf16' in k or 'conv1x1' in k or 'conv3x3' in k or 'wgrad' in k or 'wt_transpose' in k: return 'convolutions (conv_f32.hip / conv_bf16.hip / conv_mfma.hip / conv_f32x3.hip)'
    if 'lec::bn_' in k or 'lec::bn' in k: return 'BatchNorm family (bn.hip)'
    if 'joint_loss' in k: return 'fused cone loss (joint_loss.hip)'
    if 'lec::' in k: return 'other liblecone kernels'
    return 'library / framework kernels'


def union_ns(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s_, e_ in iv[1:]:
        if s_ > ce:
            tot += ce - cs; cs, ce = s_, e_
        else:
            ce = max(ce, e_)
    return tot + ce - cs


fam_iv = collections.defaultdict(list)
for nm, s_, e_, g, v, l in sel:
    fam_iv[family(short(nm))].append((s_, e_))
print('\n## kernel families: sum of launch durations vs union of launch intervals (ms per step)\n')
print('| family | launches/step | sum of durations | union of intervals | union / step wall |\n|---|---|---|---|---|')
wall = (hi - lo) / n / 1e6
for k, iv in sorted(fam_iv.items(), key=lambda kv: -sum(e_ - s_ for s_, e_ in kv[1])):
    u = union_ns(iv) / n / 1e6
    print('| %s | %.0f | %.3f | %.3f | %.3f |' % (k, len(iv) / n, sum(e_ - s_ for s_, e_ in iv) / n / 1e6, u, u / wall))
print('| ALL kernels | %.0f | %.3f | %.3f | %.3f |' % (len(sel) / n, tot / n / 1e6, union_ns([(r[1], r[2]) for r in sel]) / n / 1e6,
                                                     union_ns([(r[1], r[2]) for r in sel]) / n / 1e6 / wall))
if a.conv_tflop_per_step:
    ck = [k for k in fam_iv if k.startswith('convolutions')]
    if ck:
        u = union_ns(fam_iv[ck[0]]) / n / 1e9
        sm = sum(e_ - s_ for s_, e_ in fam_iv[ck[0]]) / n / 1e9
        print('\nconvolution family roofline (%.3f algorithmic TFLOP per step, peak %.1f TFLOP/s):\n' % (a.conv_tflop_per_step, a.peak_tflops))
        print('| basis | seconds/step | TFLOP/s | fraction of peak |\n|---|---|---|---|')
        for nm_, t_ in (('union of the family\'s launch intervals (bench.py roofline.frac)', u), ('step wall time (lower bound)', wall / 1e3), ('sum of launch durations (as if serial)', sm)):
            print('| %s | %.6f | %.2f | %.4f |' % (nm_, t_, a.conv_tflop_per_step / t_, a.conv_tflop_per_step / t_ / a.peak_tflops))
