#!/usr/bin/env python3
"""bench.py -- images/sec of the joint CNN + hyperbolic cone-loss training step on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2] [--dtype bf16|fp32]

N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
(one process per GPU, RCCL).  A step = one pass of the hot path over one batch of B positives per GPU (weak scaling):
sampler (host, bit-exact) + ResNet fwd/bwd + fused cone loss fwd/bwd + gradient all-reduce + table/CNN optimizer steps,
inputs resident in HBM.  Rank 0 prints ONE JSON line.
"""
import argparse, json, os, sys, time

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def cone_alg_bytes(B, K, D):
    """fwd+bwd algorithmic bytes per positive, rows de-duplicated inside a group, fp32."""
    return B * ((2 + 2 * K) * (2 * D * 4 + 4 * D + 4) + (1 + 2 * K) * 8)


def cpu_baseline(eng, budget_s=25.0):
    """The pinned oracle (oracle/cone_oracle.py, kind "port") timed on this box's host cores, rank 0 only: the restated
    loss path (dense-matrix sampler + numpy cone loss fwd/bwd + table step) at the full batch, plus torch-CPU ResNet
    fwd+bwd on a bounded sample of images; images/sec = B / (t_loss + B_rows * t_cnn_per_image)."""
    import numpy as np, torch
    from oracle import cone_oracle as O
    from learning_embeddings_amd.resnet import resnet18, resnet50
    B, K, D, N = eng.B, eng.K, eng.D, eng.N
    M = min(eng.M, 2048)                                       # dense (N+M)^2 bool matrix must stay small
    lm = eng.labelmap
    leaf = [lm.level_start[-1] + (j % lm.levels[-1]) for j in range(M)]
    A = O.dense_negative_adjacency(N, sorted(lm.edges), leaf)
    smp = O.DenseSampler(A, lm.levels, pick_per_level=True, seed=0)
    W = eng.table.cpu().numpy().copy()
    rs = np.random.RandomState(0)
    R = (rs.randn(M, D) * 0.3).astype(np.float32)
    frm, to = eng.positives(0)
    frm = frm[:B]; to = N + ((to[:B] - N) % M)
    t0 = time.time(); reps = 0
    m = np.zeros_like(W); v = np.zeros_like(W)
    while reps < 3 and time.time() - t0 < budget_s * 0.4:
        neg = smp.draw_batch(frm, to, K)
        loss, e_pos, e_neg, gW, gR = O.joint_loss_fwd_bwd(W, R, frm, to, neg, eng.alpha, eng.K_cone)
        W2, m, v = O.table_step_adam(W, gW.astype(np.float32), m, v, reps + 1, eng.lr, eng.K_cone)
        reps += 1
    t_loss = (time.time() - t0) / max(reps, 1)
    cores = min(os.cpu_count() or 1, 32)                      # threads actually used (more only adds sync overhead at this size)
    torch.set_num_threads(cores)
    net = (resnet50 if eng.arch == 'resnet50' else resnet18)(num_classes=D)
    n_s = 8
    x = torch.rand(n_s, 3, eng.hw, eng.hw)
    tw = time.time()
    net(x[:2]).sum().backward()                               # untimed warm-up (oneDNN primitive creation)
    tw = time.time() - tw
    t1 = time.time(); r2 = 0
    while r2 < 2 and time.time() - t1 < budget_s * 0.6 and (r2 == 0 or tw < budget_s):
        net.zero_grad(); net(x).sum().backward(); r2 += 1
    t_cnn = (time.time() - t1) / max(r2, 1) / n_s
    rows = eng.n_rows
    ips = B / (t_loss + rows * t_cnn)
    return {'value': round(ips, 3), 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
            'sample': 'oracle loss path (dense sampler + numpy cone fwd/bwd + table step) x%d at B=%d K=%d; torch-CPU %s fwd+bwd on %d images x%d, scaled to the %d CNN rows of a step'
                      % (reps, B, K, eng.arch, n_s, r2, rows),
            'loss_path_s_per_step': round(t_loss, 4), 'cnn_s_per_image': round(t_cnn, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='cfg3')
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--batch', type=int, default=None)
    ap.add_argument('--sampler', default='replicated', choices=['replicated', 'per_rank'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-stress', action='store_true')
    ap.add_argument('--no-overlap-wgrad', action='store_true', help='keep the conv weight-gradient kernels on the main stream (default: second HIP stream, +3.9 %%)')
    ap.add_argument('--check-replicas', action='store_true', help='after the run, assert that every rank holds identical parameters')
    ap.add_argument('--no-graph', action='store_true', help='launch every kernel eagerly instead of replaying the captured hipGraph of forward+loss+backward')
    ap.add_argument('--cudnn-benchmark', action='store_true', help='let MIOpen benchmark every solver per conv shape (slow start)')
    args = ap.parse_args()

    # stdout carries ONE line, the JSON result.  Everything else this process or its libraries write to file descriptor 1
    # (the reference-style banners of the host mirror, RCCL's version banner -- printed through C stdio, which would
    # otherwise land AFTER the JSON line when stdout is a pipe) goes to stderr until the result is ready.
    sys.stdout.flush()
    saved_stdout_fd = os.dup(1)
    os.dup2(2, 1)

    t_start = time.time()
    def stamp(what):
        if int(os.environ.get('RANK', 0)) == 0:
            print('[bench %7.1f s] %s' % (time.time() - t_start, what), file=sys.stderr, flush=True)
    from learning_embeddings_amd import miopen_tuning
    miopen_tuning.setup()                                       # before the first convolution
    import torch
    import torch.distributed as dist
    stamp('torch imported')
    from learning_embeddings_amd import ops, _lib, parallel
    from learning_embeddings_amd.engine import StepEngine, WORKLOADS
    from learning_embeddings_amd.resnet import conv_macs

    torch.backends.cudnn.benchmark = bool(args.cudnn_benchmark)
    rank, local_rank, world = parallel.init_process_group()
    if world != args.gpus and rank == 0:
        print('warning: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world), file=sys.stderr)
    eng = StepEngine(args.workload, dtype=args.dtype, sampler_mode=args.sampler, batch=args.batch, overlap_wgrad=not args.no_overlap_wgrad,
                     use_graph=not args.no_graph)
    dev = eng.device
    stamp('engine built')
    for i in range(args.warmup):
        eng.step()
        if i < 2:
            torch.cuda.synchronize(); stamp('warm-up step %d done' % i)
    while eng.use_graph and eng.hip_graph is None and eng.graph_error is None:
        eng.step()                                              # fewer warm-up steps than the capture needs: finish them untimed
    stamp('launch mode: %s' % ('hipGraph replay' if eng.hip_graph is not None else 'eager'))
    eng.enable_timers()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host_s = 0.0
    eng.host_wait_s = 0.0
    for _ in range(args.steps):
        th = time.perf_counter()
        eng.step()
        host_s += time.perf_counter() - th          # host time to enqueue one step (no synchronisation inside)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    host_busy_s = host_s - eng.host_wait_s
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())
    stamp('timed steps done')
    if args.check_replicas and world > 1:
        for name, t in (('label table', eng.table), ('cnn arena', eng.arena.data)):
            ref = t.clone(); dist.broadcast(ref, 0)
            same = torch.tensor([float(torch.equal(ref, t))], device=dev); dist.all_reduce(same, op=dist.ReduceOp.MIN)
            if rank == 0:
                print('[bench] replicas identical (%s): %s' % (name, bool(same.item())), file=sys.stderr)
            assert same.item() == 1.0, 'replicas diverged: ' + name
    graph_mode = eng.hip_graph is not None
    if graph_mode:
        # A replayed graph cannot carry timing events, so the per-kernel durations behind `roofline` come from eager
        # launches of the SAME step (same kernels, same two-stream overlap) right after the timed region.
        eng.set_launch_mode(False)
        def probe_step():
            # two replays first: while the GPU chews on them the host enqueues the whole eager step behind them, so the
            # event-bracketed durations carry no host launch gaps (the replays recompute the current batch's gradients
            # and change no training state; BatchNorm running statistics see the batch again)
            eng._graph_saved.replay(); eng._graph_saved.replay()
            eng.step()
        probe_step(); torch.cuda.synchronize()
        eng.timers['records'] = [r for r in eng.timers['records'] if len(r) == 4]
        ops.BN_TIMER = []
        for _ in range(5):
            probe_step()
        torch.cuda.synchronize()
    phases = eng.timer_summary()
    # The weight-gradient kernels run on a second stream next to the BatchNorm kernels, so per-kernel durations inside the
    # timed region include that sharing.  Three more steps with everything on ONE stream give the family's own duration.
    bn_isolated = None
    if eng.overlap is not None and eng.overlap.side is not None and 'fused_bn' in phases:
        side = eng.overlap.side
        eng.overlap.side = None
        ops.BN_TIMER = []
        n_iso = 3
        for _ in range(n_iso):
            eng.step()
        torch.cuda.synchronize()
        bn_isolated = sum(a.elapsed_time(b) for a, b, _ in ops.BN_TIMER) / n_iso
        eng.overlap.side = side
        ops.BN_TIMER = None
    if graph_mode:
        eng.set_launch_mode(True)
    loss_mean = float(eng.loss_acc.item()) / max(eng.step_no, 1)

    if rank == 0:
        B, K, D = eng.B, eng.K, eng.D
        ips = world * B * args.steps / dt
        # ---- roofline of the hand-written hot kernel (HBM-bound gather/scatter): algorithmic bytes / measured duration
        cone_s = phases['cone_loss'] * 1e-3
        ab = cone_alg_bytes(B, K, D)
        pmc_all = {}
        try:                                                    # HBM bytes per launch from rocprofv3 --pmc passes (profiles/)
            pmc_all = json.load(open(os.path.join(ROOT, 'profiles', 'r01_cone_stress_pmc.json')))
        except Exception:
            pass
        traffic = pmc_all.get('%d_%d_%d_%d' % (B, K, D, eng.N), {}).get('traffic_bytes_fetch_x2')
        roof_bn = None
        if 'fused_bn' in phases and getattr(eng, 'bn_bytes_per_step', 0):
            bn_s = phases['fused_bn'] * 1e-3
            bn_traffic = None
            try:                                                # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command (profiles/r01_bn_pmc.md)
                if args.workload == 'cfg3' and B == 256:
                    bn_traffic = json.load(open(os.path.join(ROOT, 'profiles', 'r01_bn_pmc.json')))['traffic_bytes_per_step_fetch_x2']
            except Exception:
                pass
            roof_bn = {'kernel': 'fused BatchNorm(+residual)(+ReLU) family (bn_stats/bn_apply/bn_bwd_reduce/bn_bwd_apply, bn.hip): all %d layer launches of the step' % int(eng.bn_launch_groups_per_step),
                       'bound': 'hbm', 'achieved': round(eng.bn_bytes_per_step / bn_s / 1e9, 1), 'peak': 8000.0, 'unit': 'GB/s',
                       'frac': round(eng.bn_bytes_per_step / bn_s / 8e12, 4), 'traffic': bn_traffic,
                       'alg_bytes_per_step': int(eng.bn_bytes_per_step), 'ms_per_step': round(phases['fused_bn'], 3)}
            if bn_isolated is not None:
                roof_bn['note'] = (('the timed region replays a hipGraph, which cannot carry timing events: these durations are HIP-event timings of 5 eager steps of the same workload run right after it; ' if graph_mode else '') +
                                   'inside the step the conv weight-gradient kernels run concurrently on a second stream and share HBM with this family; '
                                   'alone on the GPU (3 extra steps, single stream) the same launches take %.2f ms = %.0f GB/s = %.3f of peak'
                                   % (bn_isolated, eng.bn_bytes_per_step / bn_isolated / 1e6, eng.bn_bytes_per_step / bn_isolated / 1e6 / 8000.0))
                roof_bn['isolated'] = {'ms_per_step': round(bn_isolated, 3), 'achieved': round(eng.bn_bytes_per_step / bn_isolated / 1e6, 1),
                                       'frac': round(eng.bn_bytes_per_step / bn_isolated / 1e6 / 8000.0, 4)}
        roof_cone = {'kernel': 'joint_loss_kernel (fused cone loss fwd+bwd)', 'bound': 'hbm',
                     'achieved': round(ab / cone_s / 1e9, 3), 'peak': 8000.0, 'unit': 'GB/s',
                     'frac': round(ab / cone_s / 8e12, 6), 'traffic': traffic, 'alg_bytes_per_launch': ab,
                     'avg_launch_us': round(cone_s * 1e6, 2),
                     'note': 'at the north-star size (B=%d, K=%d, D=%d: %.2f MB per launch) the launch is latency-bound, not HBM-bound; see roofline_stress' % (B, K, D, ab / 1e6)}
        # ---- the step's dominant component: ResNet fwd+bwd (MFMA-bound), analytic flops / measured fwd+bwd time
        macs = conv_macs(eng.img_feat_net.model, eng.hw)
        flops = 3 * 2 * macs * eng.n_rows
        # graph mode: the replay's own duration (forward + 17 us of loss kernel + backward, no host gaps)
        cnn_s = (phases['graph_fwd_loss_bwd'] if graph_mode else phases['cnn_fwd'] + phases['cnn_bwd']) * 1e-3
        peak_tf = 2500.0 if args.dtype in ('bf16', 'fp16') else 157.3
        roof_cnn = {'kernel': '%s conv stack fwd+bwd (MIOpen/hipBLASLt via PyTorch)' % eng.arch, 'bound': 'mfma',
                    'achieved': round(flops / cnn_s / 1e12, 3), 'peak': peak_tf, 'unit': 'TFLOP/s',
                    'frac': round(flops / cnn_s / 1e12 / peak_tf, 5), 'traffic': None,
                    'gflop_per_image_fwd_bwd': round(6 * macs / 1e9, 3), 'cnn_rows_per_step': eng.n_rows}
        out = {'metric': 'images/sec (joint CNN+cone-loss step)', 'value': round(ips, 2), 'unit': 'images/sec',
               'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
               'config': {'workload': '%s: %s hierarchy (%d labels, %d levels) + %d synthetic images, %s, hyperbolic cone loss, B=%d positives/GPU, K=%d, D=%d, %dx%d'
                                      % (args.workload, WORKLOADS[args.workload][0], eng.N, eng.L, eng.M, eng.arch, B, K, D, eng.hw, eng.hw),
                          'global_batch': B * world, 'cnn_rows_per_step_per_gpu': eng.n_rows, 'cone_loss_dtype': 'f32',
                          'parallelism': 'dp%d' % world, 'sampler': args.sampler,
                          'hbm_peak_allocated_gb': round(torch.cuda.max_memory_allocated() / 1e9, 1),
                          'launch_mode': 'hipgraph (forward + loss + backward of a step replayed as one graph)' if graph_mode else 'eager', 'mean_loss': round(loss_mean, 4)},
               'phases_ms': dict({('eager_probe_' + k if graph_mode and k in ('cnn_fwd', 'cone_loss', 'cnn_bwd', 'allreduce_wait', 'fused_bn') else k): round(v, 3)
                                  for k, v in phases.items()}, host_enqueue=round(host_busy_s / args.steps * 1e3, 3)),
               # `roofline`: the hand-written kernel family that dominates the step's time (HBM-bound BatchNorm passes);
               # `roofline_cone`: the fused cone-loss kernel the metric also names; `roofline_cnn`: the whole backbone pass
               'roofline': roof_bn if roof_bn is not None else roof_cone, 'roofline_cone': roof_cone, 'roofline_cnn': roof_cnn}
        if not args.no_stress:
            # the same kernel where it IS bandwidth-bound: config 5's label-embedding stress shape (K=256) and a D=128 table
            sys.path.insert(0, os.path.join(ROOT, 'tools'))
            import bench_cone
            st = {}
            pmc = {}
            try:                                                # HBM bytes per launch measured with rocprofv3 --pmc (profiles/)
                pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r01_cone_stress_pmc.json')))
            except Exception:
                pass
            for tag, (b_, k_, d_, n_) in {'cfg5_B256_K256_D10': (256, 256, 10, 50000), 'B256_K256_D128': (256, 256, 128, 50000),
                                          'B4096_K256_D10': (4096, 256, 10, 50000)}.items():
                r = bench_cone.time_joint(b_, k_, d_, n_, b_, iters=30)
                rec = pmc.get('%d_%d_%d_%d' % (b_, k_, d_, n_), {})
                st[tag] = {'bound': 'hbm', 'achieved': round(r['GBps'], 1), 'peak': 8000.0, 'unit': 'GB/s', 'frac': round(r['GBps'] / 8000.0, 4),
                           'traffic': rec.get('traffic_bytes_fetch_x2'), 'avg_launch_us': round(r['us'], 1), 'pairs': r['pairs'],
                           'alg_bytes_per_launch': int(r['alg_MB'] * 1e6)}
            out['roofline_stress'] = st
            rb = bench_cone.time_bn(512, 256, 56, 56, True, iters=10)
            out['roofline_bn_standalone'] = {'kernel': 'fused BatchNorm+residual+ReLU fwd+bwd through the C ABI, ResNet-50 layer1 output shape [512,256,56,56] bf16 NHWC',
                                  'bound': 'hbm', 'achieved': round(rb['GBps'], 1), 'peak': 8000.0, 'unit': 'GB/s',
                                  'frac': round(rb['GBps'] / 8000.0, 4), 'traffic': None, 'alg_bytes_per_launch': int(rb['alg_MB'] * 1e6),
                                  'avg_launch_us': round(rb['us_fwd_bwd'], 1)}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(eng)
    eng.close()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)                           # C stdio buffers (RCCL / MIOpen banners) out through stderr
    except Exception:
        pass
    os.dup2(saved_stdout_fd, 1)
    os.close(saved_stdout_fd)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
