#!/usr/bin/env python3
"""Per template instance: launches per step, average duration and a caller-supplied byte count -> GB/s, from a rocprofv3 kernel trace.
    python tools/per_instance.py trace.csv --steps 4 --skip-last 21 [--prefix lec::]"""
import argparse, collections, csv, re


def main():
    ap = argparse.ArgumentParser(); ap.add_argument('csv'); ap.add_argument('--steps', type=int, default=4); ap.add_argument('--skip-last', type=int, default=21)
    ap.add_argument('--prefix', default='lec::')
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.csv)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(rows) if 'joint_loss_kernel' in r['Kernel_Name']]
    marks = marks[:len(marks) - a.skip_last] if a.skip_last else marks
    use = marks[-(a.steps + 1):]
    lo, hi = use[0], use[-1]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows[lo:hi]:
        n = r['Kernel_Name']
        if a.prefix not in n:
            continue
        short = re.sub(r'\(.*$', '', n[n.find(a.prefix):])[:110]
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        agg[short][0] += 1; agg[short][1] += d
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('%-112s %5.1f/step  avg %8.1f us  total %7.3f ms/step' % (k, c / a.steps, t / c, t / a.steps / 1e3))


if __name__ == '__main__':
    main()
