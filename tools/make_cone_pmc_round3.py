#!/usr/bin/env python3
"""profiles/r03_cone_pmc.md/.json from the passes of tools/prof_cone_round3.sh: per shape the joint_loss_kernel's average duration
(kernel trace) and its HBM traffic per launch (FETCH_SIZE x 2 + WRITE_SIZE, KiB counters) next to the algorithmic bytes of SURVEY.md 8(d).
usage: make_cone_pmc_round3.py <dir> > r03_cone_pmc.md   (also writes <dir>/r03_cone_pmc.json)"""
import csv, glob, json, os, sys

d = sys.argv[1]
KERNEL = 'joint_loss_kernel'


def alg_bytes(B, K, D):
    return B * ((2 + 2 * K) * (2 * D * 4 + 4 * D + 4) + (1 + 2 * K) * 8)


def find(pattern):
    hits = glob.glob(os.path.join(d, pattern), recursive=True)
    return hits[0] if hits else None


def counter_per_launch(sub, name):
    f = find('%s/**/*counter_collection.csv' % sub)
    if not f:
        return None, 0
    tot, n = 0.0, set()
    for row in csv.DictReader(open(f)):
        if KERNEL in row['Kernel_Name'] and row['Counter_Name'] == name:
            tot += float(row['Counter_Value']); n.add(row['Dispatch_Id'])
    return (tot * 1024.0 / len(n), len(n)) if n else (None, 0)


def duration_us(sub):
    f = find('%s/**/*kernel_trace.csv' % sub)
    if not f:
        return None, 0
    ds = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(f)) if KERNEL in r['Kernel_Name']]
    ds = sorted(ds)[:max(1, len(ds) - 6)] if len(ds) > 8 else ds           # (the first launches after the upload run with cold caches)
    return (sum(ds) / len(ds) / 1e3, len(ds)) if ds else (None, 0)


out = {}
print('# joint_loss_kernel (fused cone loss fwd + bwd, csrc/joint_loss.hip) under rocprofv3, round 3\n')
print('`bash tools/prof_cone_round3.sh` on one MI355X: per shape one `--kernel-trace --stats` pass and SEPARATE `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of')
print('`tools/prof_cone.py B K D N` (the program directly after `--`).  HBM read = FETCH_SIZE x 2 (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md), write = WRITE_SIZE;')
print('algorithmic bytes = SURVEY.md 8(d): (2 + 2K)(2 D 4 + 4 D + 4) + (1 + 2K) 8 per positive, rows de-duplicated per group.\n')
print('| B | K | D | N (table MB) | launches | avg us | algorithmic MB | HBM read MB | HBM write MB | traffic / algorithmic | algorithmic GB/s | frac of 8 TB/s |')
print('|---|---|---|---|---|---|---|---|---|---|---|---|')
for sub in sorted(glob.glob(os.path.join(d, 'kt_*'))):
    if not os.path.isdir(sub):
        continue
    tag = os.path.basename(sub)[3:]
    B, K, D, N = (int(v) for v in tag.split('_'))
    us, n = duration_us('kt_' + tag)
    rd, _ = counter_per_launch('fs_' + tag, 'FETCH_SIZE'); wr, _ = counter_per_launch('ws_' + tag, 'WRITE_SIZE')
    ab = alg_bytes(B, K, D)
    traffic = (2 * rd if rd is not None else 0) + (wr or 0) if (rd is not None or wr is not None) else None
    out[tag] = {'B': B, 'K': K, 'D': D, 'N': N, 'launches': n, 'avg_us': us, 'alg_bytes': ab, 'hbm_read_bytes': None if rd is None else 2 * rd,
                'hbm_write_bytes': wr, 'traffic_bytes': traffic}
    print('| %d | %d | %d | %d (%.1f) | %d | %s | %.2f | %s | %s | %s | %s | %s |' % (
        B, K, D, N, N * D * 4 / 1e6, n, '%.1f' % us if us else '-', ab / 1e6, '%.2f' % (2 * rd / 1e6) if rd is not None else '-',
        '%.2f' % (wr / 1e6) if wr is not None else '-', '%.2f' % (traffic / ab) if traffic else '-',
        '%.1f' % (ab / us / 1e3) if us else '-', '%.4f' % (ab / us / 1e3 / 8000.0) if us else '-'))
print('\nReading: the label table (0.08 MB at N = 2 000, 2 MB at 50 000 x 10, 25.6 MB at D = 128) and the gradient table live in L2 / the Infinity Cache: HBM traffic below the')
print('algorithmic bytes means the gathered rows and the float-atomic scatter never reach HBM; the launch is bound by dependent cache round trips and the atomic rate')
print('(bench.py `roofline_stress.second_bounds`), not by HBM bandwidth.')
json.dump(out, open(os.path.join(d, 'r03_cone_pmc.json'), 'w'), indent=1)
