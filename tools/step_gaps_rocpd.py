#!/usr/bin/env python3
"""Where a bench step is NOT running a convolution: from a rocprofv3 --kernel-trace database (rocpd sqlite) of `bench.py`, the intervals of the steady-state
steps during which no kernel of the convolution family is in flight, longest first, with the kernels that run inside each -- the part of the step
the matrix pipe cannot be busy in, i.e. what roofline.frac over the step wall time loses against the family's own interval union.
usage: step_gaps_rocpd.py <results.db> [--steps 3] [--skip-last 3] [--grid 16384] [--top 12]"""
import argparse, collections, re, sqlite3

ap = argparse.ArgumentParser(); ap.add_argument('db'); ap.add_argument('--steps', type=int, default=3)
ap.add_argument('--marker', default='joint_loss_kernel'); ap.add_argument('--skip-last', type=int, default=3)
ap.add_argument('--grid', type=int, default=0); ap.add_argument('--top', type=int, default=12)
a = ap.parse_args()
c = sqlite3.connect(a.db)
rows = c.execute('select name, start, end, grid_x from kernels order by start').fetchall()
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
    return re.sub(r'\(.*', '', nm)[:70]


def is_conv(k):
    return 'conv_f32' in k or 'conv1x1' in k or 'conv3x3' in k or 'wgrad' in k


conv = sorted((s, e) for nm, s, e, g in sel if is_conv(nm))
merged = []
for s, e in conv:
    if merged and s <= merged[-1][1]:
        merged[-1][1] = max(merged[-1][1], e)
    else:
        merged.append([s, e])
gaps = []
prev = lo
for s, e in merged:
    if s > prev:
        gaps.append((prev, s))
    prev = max(prev, e)
if hi > prev:
    gaps.append((prev, hi))
tot = sum(e - s for s, e in gaps)
print('# intervals of the step with no convolution in flight (%d steady-state steps)\n' % n)
print('wall per step %.3f ms; without a convolution in flight: %.3f ms per step in %.0f intervals per step\n' % ((hi - lo) / n / 1e6, tot / n / 1e6, len(gaps) / n))
hist = collections.Counter()
for s, e in gaps:
    d = (e - s) / 1e3
    hist['< 20 us' if d < 20 else '20 - 100 us' if d < 100 else '100 - 300 us' if d < 300 else '300 us - 1 ms' if d < 1000 else '>= 1 ms'] += e - s
print('| interval length | ms per step |\n|---|---|')
for k in ('< 20 us', '20 - 100 us', '100 - 300 us', '300 us - 1 ms', '>= 1 ms'):
    print('| %s | %.3f |' % (k, hist[k] / n / 1e6))
# who fills the intervals: every interval's time split over the kernels in flight (equal shares while several run), summed per kernel name
share = collections.Counter(); idle_tot = 0
for s, e in gaps:
    ev = []
    for nm, s_, e_, g in sel:
        if s_ < e and e_ > s and not is_conv(nm):
            ev.append((max(s, s_), 1, short(nm))); ev.append((min(e, e_), -1, short(nm)))
    ev.sort(key=lambda x: (x[0], x[1]))
    live = collections.Counter(); t = s
    for tt, d, nm in ev:
        if tt > t:
            k = sum(live.values())
            if k == 0: idle_tot += tt - t
            else:
                for nm2, c2 in live.items():
                    if c2 > 0: share[nm2] += (tt - t) * c2 / k
            t = tt
        live[nm] += d
    if e > t: idle_tot += e - t
print('\n| kernels in flight while no convolution is (time shared equally between concurrent ones) | ms per step |\n|---|---|')
for nm, v in share.most_common(14):
    print('| `%s` | %.3f |' % (nm.replace('lec::', '').replace('at::native::', ''), v / n / 1e6))
print('| (nothing in flight) | %.3f |' % (idle_tot / n / 1e6))
print('\n| interval (us) | at (ms into the window) | idle inside (us) | kernels inside (us each, in start order) |\n|---|---|---|---|')
for s, e in sorted(gaps, key=lambda g: g[0] - g[1])[:a.top * n]:
    inside = [(nm, max(s, s_), min(e, e_)) for nm, s_, e_, g in sel if s_ < e and e_ > s and not is_conv(nm)]
    iv = sorted((x[1], x[2]) for x in inside)
    busy = 0; cs = ce = None
    for s_, e_ in iv:
        if cs is None: cs, ce = s_, e_
        elif s_ > ce: busy += ce - cs; cs, ce = s_, e_
        else: ce = max(ce, e_)
    if cs is not None: busy += ce - cs
    names = ', '.join('%s %.0f' % (short(nm).replace('lec::', '').replace('at::native::', ''), (e_ - s_) / 1e3) for nm, s_, e_ in sorted(inside, key=lambda x: x[1]))
    print('| %.0f | %.2f | %.0f | %s |' % ((e - s) / 1e3, (s - lo) / 1e6, (e - s - busy) / 1e3, names[:900]))
