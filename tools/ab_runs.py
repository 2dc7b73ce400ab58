#!/usr/bin/env python3
"""Run bench.py several times back to back with different environment settings and print one compact line per run.

    python tools/ab_runs.py LEC_CONV1X1_GEMM=0 LEC_CONV1X1_GEMM=1 LEC_CONV1X1_GEMM=0 -- --no-cpu-baseline --no-stress
Each positional token before `--` is one run: comma-separated NAME=VALUE pairs ('-' = no change)."""
import json, os, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    argv = sys.argv[1:]
    extra = []
    if '--' in argv:
        i = argv.index('--'); extra = argv[i + 1:]; argv = argv[:i]
    for n, tok in enumerate(argv):
        env = dict(os.environ)
        if tok != '-':
            for kv in tok.split(','):
                k, v = kv.split('=', 1); env[k] = v
        t0 = time.time()
        p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + extra, env=env, capture_output=True, text=True)
        wall = time.time() - t0
        try:
            r = json.loads(p.stdout.strip().splitlines()[-1])
            print('run %d [%s] %.1f img/s %.2f ms/step wall %.1fs phases %s' % (n, tok, r['value'], r['ms_per_step'], wall, r.get('phases_ms')), flush=True)
        except Exception as e:                                  # noqa: BLE001
            print('run %d [%s] FAILED rc=%d (%s)\n%s' % (n, tok, p.returncode, e, p.stderr[-2000:]), flush=True)
        warn = [l for l in p.stderr.splitlines() if 'bench' not in l and 'amdgpu.ids' not in l]
        if warn:
            print('   stderr: ' + ' | '.join(warn[:6])[:600], flush=True)


if __name__ == '__main__':
    main()
