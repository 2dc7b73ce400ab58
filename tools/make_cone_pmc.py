#!/usr/bin/env python3
"""profiles/<name>.md/.json from the passes of tools/prof_cone_pmc.sh: per shape joint_loss_kernel's average duration (kernel trace),
its HBM traffic per launch (FETCH_SIZE x 2 + WRITE_SIZE, KiB counters) next to the algorithmic bytes of SURVEY.md 8(d), and per-launch
averages of the SQ / TCC / TCP counters collected in the other passes.  usage: make_cone_pmc.py <dir> [<name>] > <name>.md  (the .json goes to <dir>/<name>.json)"""
import csv, glob, hashlib, json, os, sys

d = sys.argv[1]
KERNEL = 'joint_loss_kernel'
CS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'learning_embeddings_amd', 'csrc')


def alg_bytes(B, K, D):
    return B * ((2 + 2 * K) * (2 * D * 4 + 4 * D + 4) + (1 + 2 * K) * 8)


def counters(tag):
    """{counter: mean per launch} over every pm*_<tag> pass (first 3 launches dropped: cold caches)."""
    out = {}
    for f in glob.glob(os.path.join(d, 'pm*_%s' % tag, '**', '*counter_collection.csv'), recursive=True):
        per = {}
        for row in csv.DictReader(open(f)):
            if KERNEL in row['Kernel_Name']:
                per.setdefault(row['Counter_Name'], {}).setdefault(int(row['Dispatch_Id']), 0.0)
                per[row['Counter_Name']][int(row['Dispatch_Id'])] += float(row['Counter_Value'])
        for name, by in per.items():
            vals = [by[k] for k in sorted(by)][3:] or list(by.values())
            out[name] = sum(vals) / len(vals)
    return out


def duration_us(tag):
    hits = glob.glob(os.path.join(d, 'kt_%s' % tag, '**', '*kernel_trace.csv'), recursive=True)
    if not hits:
        return None, 0
    ds = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(hits[0])) if KERNEL in r['Kernel_Name']]
    ds = sorted(ds)[:max(1, len(ds) - 6)] if len(ds) > 8 else ds
    return (sum(ds) / len(ds) / 1e3, len(ds)) if ds else (None, 0)


out = {'kernel_sources_sha256': {f: hashlib.sha256(open(os.path.join(CS, f), 'rb').read()).hexdigest() for f in ('joint_loss.hip', 'lec_common.h')}, 'shapes': {}}
rows = []
for sub in sorted(glob.glob(os.path.join(d, 'kt_*'))):
    if not os.path.isdir(sub):
        continue
    tag = os.path.basename(sub)[3:]
    B, K, D, N = (int(v) for v in tag.split('_'))
    us, n = duration_us(tag)
    c = counters(tag)
    rd = c.get('FETCH_SIZE'); wr = c.get('WRITE_SIZE')
    rd = None if rd is None else rd * 1024.0 * 2; wr = None if wr is None else wr * 1024.0
    ab = alg_bytes(B, K, D)
    traffic = None if (rd is None and wr is None) else (rd or 0) + (wr or 0)
    rec = {'B': B, 'K': K, 'D': D, 'N': N, 'launches': n, 'avg_us': us, 'alg_bytes': ab, 'hbm_read_bytes': rd, 'hbm_write_bytes': wr, 'traffic_bytes': traffic,
           'counters_per_launch': {k: v for k, v in sorted(c.items()) if k not in ('FETCH_SIZE', 'WRITE_SIZE')}}
    out['shapes'][tag] = rec; rows.append(rec)
print('# joint_loss_kernel (fused cone loss fwd + bwd, csrc/joint_loss.hip) under rocprofv3, round 4\n')
print('`bash tools/prof_cone_round4.sh` on one MI355X: per shape one `--kernel-trace --stats` pass and SEPARATE `--pmc` passes of `tools/prof_cone.py B K D N` (the program directly')
print('after `--`).  HBM read = FETCH_SIZE x 2 (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md), write = WRITE_SIZE; algorithmic bytes = SURVEY.md 8(d):')
print('(2 + 2K)(2 D 4 + 4 D + 4) + (1 + 2K) 8 per positive, rows de-duplicated per group.  Kernel geometry: the round-4 wave-count rule (joint_loss.hip joint_geometry).\n')
print('| B | K | D | N (table MB) | avg us | algorithmic MB | HBM read MB | HBM write MB | traffic / algorithmic | algorithmic GB/s | frac of 8 TB/s |')
print('|---|---|---|---|---|---|---|---|---|---|---|')
f2 = lambda v, s=1e6: '-' if v is None else '%.2f' % (v / s)
for r in rows:
    us = r['avg_us']
    print('| %d | %d | %d | %d (%.1f) | %s | %.2f | %s | %s | %s | %s | %s |' % (
        r['B'], r['K'], r['D'], r['N'], r['N'] * r['D'] * 4 / 1e6, '%.1f' % us if us else '-', r['alg_bytes'] / 1e6, f2(r['hbm_read_bytes']), f2(r['hbm_write_bytes']),
        '%.2f' % (r['traffic_bytes'] / r['alg_bytes']) if r['traffic_bytes'] else '-', '%.1f' % (r['alg_bytes'] / us / 1e3) if us else '-',
        '%.4f' % (r['alg_bytes'] / us / 1e3 / 8000.0) if us else '-'))
print('\n## what the launch waits for (per-launch counter averages; SQ_* are summed over the chip)\n')
print('| B x K x D | waves | VALU instructions per wave | SQ_ACTIVE_INST_VALU | SQ_WAIT_INST_ANY | wait : VALU-active | SQ_ACTIVE_INST_VMEM | L2 requests | L2 hit rate | L2 atomic requests | atomic requests / us | gradient elements a launch COULD scatter (rows x D) | TCC_EA0_ATOMIC |')
print('|---|---|---|---|---|---|---|---|---|---|---|---|---|')
for r in rows:
    c = r['counters_per_launch']; us = r['avg_us'] or 0
    g = lambda k: c.get(k)
    num = lambda v: '-' if v is None else ('%.3g' % v)
    ratio = lambda a, b: '-' if (a is None or not b) else '%.2f' % (a / b)
    n_elems = r['B'] * (2 + 2 * r['K']) * r['D']
    print('| %d x %d x %d | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %.3g | %s |' % (
        r['B'], r['K'], r['D'], num(g('SQ_WAVES')), ratio(g('SQ_INSTS_VALU'), g('SQ_WAVES')), num(g('SQ_ACTIVE_INST_VALU')), num(g('SQ_WAIT_INST_ANY')),
        ratio(g('SQ_WAIT_INST_ANY'), g('SQ_ACTIVE_INST_VALU')), num(g('SQ_ACTIVE_INST_VMEM')), num(g('TCC_REQ_sum')),
        '-' if not g('TCC_REQ_sum') else '%.0f %%' % (100.0 * (g('TCC_HIT_sum') or 0) / g('TCC_REQ_sum')), num(g('TCC_ATOMIC')),
        '-' if (g('TCC_ATOMIC') is None or not us) else '%.0f' % (g('TCC_ATOMIC') / us), n_elems, num(g('TCC_EA0_ATOMIC'))))
print('\nReading.  HBM: traffic is 0.14 - 0.8 x the algorithmic bytes at D = 10 (the 2 MB table and its gradient live in L2: 72 - 94 % hit rate) -- bandwidth is not the')
print('bound at any of these sizes.  Atomics: only pairs whose loss term is live scatter a gradient (positives, and negatives inside the margin: a few percent of the')
print('2K B pairs once energies spread), so the L2 sees 26 k - 83 k atomic requests per launch, ~1 000 per microsecond -- two orders of magnitude under the rate the')
print('guide quotes for whole-row segments; the "float-atomic bound" of round 3 was a guess the counters do not support.  What the waves do: vector-ALU issue cycles are')
print('about TWICE the cycles spent waiting on any instruction at the K = 256 shapes (the cone energy is ~50 correctly rounded fp32 operations per pair with acosf,')
print('asinf, sqrtf and divisions, evaluated for every pair; -ffp-contract=off, no fast math: the reference\'s op-by-op rounding), and at the north-star size (256 x 5 x 10: 256 waves on 1 024')
print('SIMDs, 9 us) the launch is a dependent chain on a quarter-full chip: latency, not throughput.  Binding resource: the vector ALU at K = 256, launch + dependent-load')
print('latency at K = 5.')
json.dump(out, open(os.path.join(d, (sys.argv[2] if len(sys.argv) > 2 else 'cone_pmc') + '.json'), 'w'), indent=1)
