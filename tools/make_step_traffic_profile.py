#!/usr/bin/env python3
"""profiles/r01_step_traffic.{json,md}: FETCH_SIZE / WRITE_SIZE of EVERY kernel of a bench step, by family, from the two summaries
`tools/summarize_pmc.py --prefix ""` writes (gpurun_out/pmc_all_FETCH_SIZE.json, pmc_all_WRITE_SIZE.json).

    python tools/make_step_traffic_profile.py gpurun_out <ms per step of the un-profiled bench run>"""
import collections, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fam(k):
    if 'lec::bn_' in k: return 'liblecone BatchNorm family'
    if 'lec::conv' in k or 'lec::wgrad' in k: return 'liblecone MFMA convolutions'
    if 'lec::' in k: return 'liblecone other (loss, pooling, Adam, ...)'
    if 'wrw' in k.lower(): return 'library weight-gradient convolutions (second stream)'
    if 'igemm' in k or 'ck' in k.lower() or 'Cijk' in k or 'gemm' in k.lower() or 'conv' in k.lower(): return 'library forward / data-gradient convolutions and GEMMs'
    return 'elementwise / copies / fills'


def main():
    d, ms = sys.argv[1], float(sys.argv[2])
    F = json.load(open(os.path.join(d, 'pmc_all_FETCH_SIZE.json')))['bytes_per_step']
    Wr = json.load(open(os.path.join(d, 'pmc_all_WRITE_SIZE.json')))['bytes_per_step']
    f = collections.defaultdict(float); w = collections.defaultdict(float)
    for k, v in F.items(): f[fam(k)] += v
    for k, v in Wr.items(): w[fam(k)] += v
    tf = sum(f.values()) / 1e9; tw = sum(w.values()) / 1e9
    own = sum(v for k, v in f.items() if 'liblecone' in k) / 1e9
    lo, hi = tf + own + tw, 2 * tf + tw
    md = ['# HBM traffic of the whole bench step, all kernels (rocprofv3 PMC, round 1, MI355X, final build)', '',
          '`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (two passes) over `python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-stress`, every '
          'kernel of the run summed per step by `tools/summarize_pmc.py --prefix ""`, assembled by `tools/make_step_traffic_profile.py`.  FETCH_SIZE on gfx950 counts '
          'half of the bytes of wide (16 B/lane) streaming reads (MI355X guide, HBM section; verified on this repo\'s kernels against their algorithmic bytes): the '
          'x2 column applies that correction, which is exact for liblecone\'s kernels and an upper bound for the library kernels (narrower loads are counted in full).', '',
          '| kernel family | FETCH_SIZE raw GB | read GB (x2) | WRITE_SIZE GB |', '|---|---|---|---|']
    for k in sorted(set(f) | set(w), key=lambda k: -(2 * f[k] + w[k])):
        md.append('| %s | %.1f | %.1f | %.1f |' % (k, f[k] / 1e9, 2 * f[k] / 1e9, w[k] / 1e9))
    md += ['| **total** | %.1f | %.1f | %.1f |' % (tf, 2 * tf, tw), '',
           'Per step the GPU moves between %.0f GB (library reads taken at face value) and %.0f GB (all reads corrected) through HBM in %.1f ms: **%.1f-%.1f TB/s averaged '
           'over the whole step, both streams together**, against the 8 TB/s peak and the 6.29 TB/s the guide measures for a plain float4 copy.  The step as a whole is '
           'HBM-bound: the experiments of this round agree (faster weight-gradient kernels on the second stream moving the same bytes changed nothing; every change that '
           'removed bytes -- statistics in convolution epilogues, the single-write residual gradient, BatchNorm-backward pass 1 in data-gradient epilogues, the BatchNorm '
           'apply pass inside a second run of conv3 -- moved the step by 0.1-0.2 ms per GB).' % (lo, hi, ms, lo / ms, hi / ms), '']
    open(os.path.join(ROOT, 'profiles', 'r01_step_traffic.md'), 'w').write('\n'.join(md))
    json.dump({'fetch_raw_bytes_per_step': F, 'write_bytes_per_step': Wr}, open(os.path.join(ROOT, 'profiles', 'r01_step_traffic.json'), 'w'), indent=1)
    print('step traffic %.0f-%.0f GB, %.1f-%.1f TB/s' % (lo, hi, lo / ms, hi / ms))


if __name__ == '__main__':
    main()
