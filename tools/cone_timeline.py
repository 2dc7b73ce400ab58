#!/usr/bin/env python3
"""Where a launch of the fused cone-loss kernel spends its time: a per-wave, per-phase timeline.

Builds csrc/joint_loss.hip ALONE with -DLEC_JL_STAMP into tools/microbench/libjl_stamp.so (instrumentation: s_memrealtime at wave entry / exit,
s_memtime deltas around the phases; arithmetic untouched), launches it through the same C entry point as the product, and prints a markdown report:

    python tools/cone_timeline.py [--out profiles/r05_cone_timeline.md]           (on the MI355X)

Per shape: launch duration (HIP events, uninstrumented product library), the ramp (first / last wave entry), the tail (last exits), and the median wave's
cycles per phase: u_b / v_b gather + projection, rows of the iterations (arrival of the prefetched rows + projection), energies (the acos / asin / sqrt /
divide chain at the reference's rounding), backward (coefficients, Jacobian chain, atomics issued), final u_b / v_b reduction + row adds, loss hand-off
(block reduction, partial, agent-scope ticket)."""
import argparse, ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

CSRC = os.path.join(ROOT, 'learning_embeddings_amd', 'csrc')
SO = os.path.join(ROOT, 'tools', 'microbench', 'libjl_stamp.so')


def build():
    cmd = ['/opt/rocm/bin/hipcc', '-DLEC_JL_STAMP', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-munsafe-fp-atomics',
           '-shared', os.path.join(CSRC, 'joint_loss.hip'), os.path.join(CSRC, 'abi.cpp'), '-o', SO]
    subprocess.run(cmd, check=True)


def shape_inputs(B, K, D, N, M, seed=0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    W = torch.randn(N, D, generator=g); W = (W / W.norm(dim=1, keepdim=True) * (0.1 + 0.05 * torch.rand(N, 1, generator=g))).cuda()
    R = (torch.randn(M, D, generator=g) * 0.3).cuda()
    frm = torch.randint(0, N, (B,), generator=g, dtype=torch.int32).cuda()
    to = (-1 - torch.randint(0, M, (B,), generator=g, dtype=torch.int32)).cuda()
    neg = torch.randint(0, N, (B, 2 * K), generator=g, dtype=torch.int32).cuda()
    return W, R, frm, to, neg


def run(lib, B, K, D, N, grad=True, reps=7):
    M = B
    W, R, frm, to, neg = shape_inputs(B, K, D, N, M)
    gt = torch.zeros_like(W); gf = torch.zeros_like(R)
    e_pos = torch.empty(B, device='cuda'); e_neg = torch.empty(B, 2 * K, device='cuda'); loss = torch.empty(1, device='cuda')
    ws = torch.zeros(1 << 20, dtype=torch.uint8, device='cuda')
    T_, E_, it_, tpg, waves = (C.c_int() for _ in range(5))
    assert lib.lec_jl_geometry(B, K, D, C.byref(T_), C.byref(E_), C.byref(it_), C.byref(tpg), C.byref(waves)) == 0
    nw = ((waves.value + 7) // 8) * 8
    stamps = torch.zeros(nw * 10, dtype=torch.int64, device='cuda')
    lib.lec_jl_set_stamp_buffer(C.c_void_p(stamps.data_ptr()))
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    recs = []
    for r in range(reps):
        stamps.zero_(); torch.cuda.synchronize()
        rc = lib.lec_joint_loss_fwd_bwd(0, 1, 1, p(W), C.c_int64(D), N, p(R), C.c_int64(D), M, p(frm), p(to), p(neg), None, B, K, D, C.c_float(0.1), C.c_float(0.01),
                                        p(e_pos), p(e_neg), p(loss), p(gt) if grad else None, p(gf) if grad else None, p(ws), C.c_int64(ws.numel()), st)
        assert rc == 0, lib.lec_last_error()
        torch.cuda.synchronize()
        recs.append(stamps.cpu().numpy().reshape(nw, 10)[:waves.value].copy())
    return dict(T=T_.value, EPL=E_.value, iters=it_.value, tasks_per_group=tpg.value, waves=waves.value), recs[-1], recs


def product_us(B, K, D, N, grad=True):
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import bench_cone
    return bench_cone.time_joint(B, K, D, N, B, iters=30, grad=grad)['us']


def report(B, K, D, N, lib, out):
    geo, rec, recs = run(lib, B, K, D, N)
    us = product_us(B, K, D, N); us_f = product_us(B, K, D, N, grad=False)
    rt0, rt1 = rec[:, 0].astype(np.int64), rec[:, 1].astype(np.int64)
    t0 = rt0.min()
    start = (rt0 - t0) / 100.0; end = (rt1 - t0) / 100.0                      # 100 MHz -> us
    total_c = rec[:, 8].astype(np.float64)
    life = end - start
    mhz = np.median(total_c / np.maximum(life, 1e-3))                         # shader cycles per us, from the waves themselves
    ph = rec[:, 2:8].astype(np.float64) / mhz
    names = ['u_b, v_b: gather + projection', 'iteration rows: arrival + projection', 'energies (cone forward)', 'backward (coefficients, Jacobian, atomics issued)',
             'u_b / v_b gradient: wave reduction + row adds', 'loss hand-off (block sum, partial, ticket)']
    xcc = (rec[:, 9] >> 32) & 0xf
    w = out.write
    w('\n### %d x %d x %d  (N = %d)\n\n' % (B, K, D, N))
    w('geometry: T = %d lanes per pair, %d row elements per lane, %d iteration(s) of %d pairs per wave, %d wave(s) per positive: **%d waves** on 1 024 SIMDs (%.2f per SIMD)\n\n'
      % (geo['T'], geo['EPL'], geo['iters'], 64 // geo['T'], geo['tasks_per_group'], geo['waves'], geo['waves'] / 1024.0))
    w('launch duration (product library, HIP events over 30 graph-replayed launches): **%.1f us** forward + backward, %.1f us forward only\n\n' % (us, us_f))
    w('| | us |\n|---|---|\n')
    w('| first wave enters -> last wave enters (dispatch ramp) | %.2f |\n' % start.max())
    w('| first wave enters -> median wave exits | %.2f |\n' % np.median(end))
    w('| first wave enters -> last wave exits (the instrumented launch, kernel body only) | %.2f |\n' % end.max())
    w('| a wave\'s life, median / p90 / max | %.2f / %.2f / %.2f |\n' % (np.median(life), np.percentile(life, 90), life.max()))
    w('| shader clock seen by the waves | %.0f MHz |\n\n' % mhz)
    w('| phase of a wave | median us | p90 us | share of the median wave |\n|---|---|---|---|\n')
    med = np.median(ph, axis=0); p90 = np.percentile(ph, 90, axis=0)
    for i, n in enumerate(names):
        w('| %s | %.2f | %.2f | %.0f %% |\n' % (n, med[i], p90[i], 100 * med[i] / max(np.median(life), 1e-9)))
    w('| (sum of the phases) | %.2f | | |\n\n' % med.sum())
    w('waves per XCD: %s; wave entry by XCD (median us): %s\n' % (np.bincount(xcc, minlength=8).tolist(), [round(float(np.median(start[xcc == x])), 2) if (xcc == x).any() else None for x in range(8)]))
    return dict(us=us, us_fwd=us_f, life=float(np.median(life)), ramp=float(start.max()), med=med.tolist(), waves=geo['waves'])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=None)
    ap.add_argument('--no-build', action='store_true')
    a = ap.parse_args()
    if not a.no_build:
        build()
    from learning_embeddings_amd import _lib            # torch's HIP runtime first, then the product library (bench_cone uses it)
    lib = C.CDLL(SO)
    lib.lec_last_error.restype = C.c_char_p
    out = open(a.out, 'w') if a.out else sys.stdout
    out.write('# joint_loss_kernel: where a launch\'s time goes (per-wave stamps, `python tools/cone_timeline.py`)\n\n'
              'Instrumentation build of csrc/joint_loss.hip (`-DLEC_JL_STAMP`: `s_memrealtime` at wave entry / exit, `s_memtime` around the phases; same arithmetic, '
              'a few scalar instructions per phase).  Launch durations are the PRODUCT library\'s (HIP events).  One MI355X.\n')
    res = {}
    for shape in [(256, 256, 10, 50000), (4096, 256, 10, 50000), (256, 256, 128, 50000), (256, 5, 10, 2000)]:
        res[shape] = report(*shape, lib, out)
    if a.out:
        out.close()


if __name__ == '__main__':
    main()
