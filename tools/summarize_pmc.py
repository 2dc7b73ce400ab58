#!/usr/bin/env python3
"""Sum a rocprofv3 --pmc counter (FETCH_SIZE or WRITE_SIZE, reported in KiB) per kernel family and per bench step.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o f -- python3 bench.py --steps 2 --warmup 1 --no-graph ...
    python tools/summarize_pmc.py out/f_counter_collection.csv [--prefix lec::bn_]

Steps are counted by the launches of the fused loss kernel (one per step)."""
import argparse, collections, csv, json, sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('csv')
    ap.add_argument('--prefix', default='lec::bn_')
    ap.add_argument('--marker', default='joint_loss_kernel')
    a = ap.parse_args()
    tot = collections.defaultdict(float); steps = 0; counter = None
    seen = set()
    with open(a.csv) as f:
        for row in csv.DictReader(f):
            name = row['Kernel_Name']
            if a.marker in name and (row['Dispatch_Id'], row['Counter_Name']) not in seen:
                seen.add((row['Dispatch_Id'], row['Counter_Name'])); steps += 1
            i = name.find(a.prefix)
            if i < 0:
                continue
            short = name[i:].split('<')[0].split('(')[0]
            counter = row['Counter_Name']
            tot[short] += float(row['Counter_Value'])
    steps = max(steps, 1)
    out = {'counter': counter, 'steps': steps, 'bytes_per_step': {k: v * 1024.0 / steps for k, v in sorted(tot.items())}}
    out['total_bytes_per_step'] = sum(out['bytes_per_step'].values())
    json.dump(out, sys.stdout, indent=1); print()


if __name__ == '__main__':
    main()
