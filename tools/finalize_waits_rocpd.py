#!/usr/bin/env python3
"""What the small BatchNorm finalize launches of the bench step wait for: from a rocprofv3 --kernel-trace database, every finalize / coeffs launch of the steady-state
steps with its duration, grouped by the kernel of the OTHER stream that covers most of its interval.  usage: finalize_waits_rocpd.py <results.db> [--steps 3] [--grid 16384]"""
import argparse, collections, re, sqlite3
ap = argparse.ArgumentParser(); ap.add_argument('db'); ap.add_argument('--steps', type=int, default=3); ap.add_argument('--marker', default='joint_loss_kernel')
ap.add_argument('--skip-last', type=int, default=3); ap.add_argument('--grid', type=int, default=0)
a = ap.parse_args()
c = sqlite3.connect(a.db)
rows = c.execute('select name, start, end, grid_x from kernels order by start').fetchall()
marks = [r[1] for r in rows if a.marker in r[0] and (a.grid == 0 or r[3] == a.grid)]
marks = marks[:-a.skip_last] if a.skip_last else marks
lo, hi = marks[-a.steps - 1], marks[-1]
sel = [r for r in rows if lo <= r[1] < hi]
short = lambda nm: re.sub(r'\(.*', '', re.sub(r'^void ', '', nm)).replace('lec::', '')[:60]
fin = [r for r in sel if 'finalize' in r[0] or 'coeffs' in r[0]]
by = collections.defaultdict(list)
for nm, s, e, g in fin:
    best, bt = '(nothing else in flight)', 0
    for nm2, s2, e2, g2 in sel:
        if s2 < e and e2 > s and not (s2 == s and e2 == e and nm2 == nm):
            ov = min(e, e2) - max(s, s2)
            if ov > bt: best, bt = short(nm2), ov
    by[best].append((e - s) / 1e3)
print('# finalize / coeffs launches of %d steps: %d launches, %.1f us mean\n' % (a.steps, len(fin), sum((e - s) for _, s, e, _ in fin) / 1e3 / max(len(fin), 1)))
print('| kernel covering most of the launch | launches per step | mean us | median us | max us | ms per step |\n|---|---|---|---|---|---|')
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print('| `%s` | %.1f | %.1f | %.1f | %.1f | %.3f |' % (k, len(v) / a.steps, sum(v) / len(v), v2[len(v2) // 2], v2[-1], sum(v) / a.steps / 1e3))
