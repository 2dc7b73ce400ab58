#!/usr/bin/env python3
"""N data-parallel ranks of the FILE-BACKED drop-in trainer on however many GPUs the box has (VERDICT r03 "next round" #4a: de-risk N = 8 on the
one GPU a gpurun box offers).  `python tools/dp8_trainer_files.py --ranks 8` starts the ranks itself (bench.self_launch: children of
torch.distributed.run, never an exec of a process that touched the GPU; with fewer GPUs than ranks they share devices and exchange
gradients over gloo).  Every rank builds JointEmbeddings over the SAME image files with its own HBM image store, DataLoader workers and
decode pool, and runs `train_epoch` for a few epochs at a per-rank batch of B positives: the sampler walks the GLOBAL batch's stream
(N x B positives, 'replicated'), the host contends for cores the way it will on an 8-GPU node.

What a shared GPU can and cannot show: the GPU time of a step is N x a real step's (the ranks take turns), so wall time per step says
nothing; HOST cost does not depend on whose GPU runs the kernels.  Reported per rank: CPU seconds consumed per step by the training
process (all its threads: launches, autograd, lookahead, decode pool) and by its DataLoader workers, against the 131 ms a real step takes
on a GPU of its own; replicas identical after the run (label table + CNN arena, broadcast compare).  Rank 0 prints ONE JSON line."""
import argparse, json, os, shutil, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ranks', type=int, default=8); ap.add_argument('--workload', default='cfg2'); ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--images', type=int, default=4096); ap.add_argument('--epochs', type=int, default=3); ap.add_argument('--workers', type=int, default=2)
    ap.add_argument('--dir', default=None)
    a = ap.parse_args()
    if 'LOCAL_RANK' not in os.environ and a.ranks > 1:
        d = tempfile.mkdtemp(prefix='lec_dp8_')                 # the files once, before any rank starts (nothing has touched the GPU)
        bench.write_image_files(d, a.images + 80)
        try:
            sys.exit(bench.self_launch(a.ranks, script=os.path.abspath(__file__), argv=sys.argv[1:] + ['--dir', d]))
        finally:
            shutil.rmtree(d, ignore_errors=True)
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get('LEC_DP8_WATCHDOG_S', '150')), exit=True)      # a hung rank says where, and the box is released
    import numpy as np, torch, torch.distributed as dist
    import psutil
    from learning_embeddings_amd import parallel
    d = a.dir or tempfile.mkdtemp(prefix='lec_dp8_')
    if a.dir is None:
        bench.write_image_files(d, a.images + 80)
    paths = [os.path.join(d, 'img_%06d.jpg' % j) for j in range(a.images + 80)]
    rank, local_rank, world = parallel.init_process_group()
    t_start = time.time()
    def stamp(msg):
        print('[rank %d %6.1f s] %s' % (rank, time.time() - t_start, msg), file=sys.stderr, flush=True)
    tr, crit, dl, cfg = bench._bench_trainer(a, 'fp32', a.images, 64, lambda j: paths[j], a.workers)
    B = cfg[2]
    stamp('trainer built')
    me = psutil.Process()
    rows = []
    for ep in range(a.epochs):
        tr.epoch = ep
        kids0 = sum((c.cpu_times().user + c.cpu_times().system) for c in me.children(recursive=True))
        c0 = time.process_time(); torch.cuda.synchronize(); dist.barrier(); t0 = time.perf_counter(); tr.host_wait_s = 0.0
        running, steps = tr.train_epoch(on_step=lambda s_: stamp('epoch %d step %d' % (ep + 1, s_)) if (s_ <= 2 or s_ % 8 == 0) else None)
        faulthandler.cancel_dump_traceback_later(); faulthandler.dump_traceback_later(int(os.environ.get('LEC_DP8_WATCHDOG_S', '150')), exit=True)
        t_loop = time.perf_counter(); w_loop = tr.host_wait_s
        torch.cuda.synchronize(); t1 = time.perf_counter(); c1 = time.process_time()
        kids1 = sum((c.cpu_times().user + c.cpu_times().system) for c in me.children(recursive=True))
        rows.append({'epoch': ep + 1, 'steps': steps, 'wall_ms_per_step_shared_gpu': round((t1 - t0) / steps * 1e3, 1),
                     'host_busy_ms_per_step_training_thread': round((t_loop - t0 - w_loop) / steps * 1e3, 2),
                     'host_cpu_ms_per_step_training_process': round((c1 - c0) / steps * 1e3, 2),
                     'host_cpu_ms_per_step_dataloader_workers': round(max(kids1 - kids0, 0.0) / steps * 1e3, 2),
                     'files_decoded': tr.image_store.stats['decoded_here'] + tr.image_store.stats['decoded_by_workers']})
    identical = True
    for t in (tr.model.embeddings.weight.data, tr.arena.data):
        ref = t.clone(); dist.broadcast(ref, 0)
        same = torch.tensor([float(torch.equal(ref, t))], device=t.device); dist.all_reduce(same, op=dist.ReduceOp.MIN)
        identical = identical and bool(same.item() == 1.0)
    one = torch.ones(1, device=tr.device); dist.all_reduce(one)
    gathered = [None] * world
    dist.all_gather_object(gathered, rows)
    if rank == 0:
        warm = [[r[-1]['host_busy_ms_per_step_training_thread'], r[-1]['host_cpu_ms_per_step_training_process']] for r in gathered]
        real_step_ms = 131.0
        out = {'what': 'file-backed JointEmbeddings.train_epoch, %d ranks on %d GPU(s)' % (world, torch.cuda.device_count()), 'ranks': int(one.item()),
               'backend': dist.get_backend(), 'workload': a.workload, 'arch': cfg[1], 'batch_per_rank': B, 'global_batch': B * world,
               'sampler': 'replicated: every rank walks the global batch\'s stream (%d positives x 2K = %d draws per step per rank)' % (B * world, B * world * 2 * cfg[3]),
               'image_files': a.images, 'dataloader_workers_per_rank': a.workers, 'host_cores': os.cpu_count(),
               'replicas_identical_after_run': identical, 'per_rank_epochs': gathered,
               'last_epoch_host_ms_per_step': {'training_thread_busy_max_over_ranks': max(w[0] for w in warm), 'process_cpu_max_over_ranks': max(w[1] for w in warm),
                                                   'fraction_of_a_real_step': round(max(w[0] for w in warm) / real_step_ms, 3),
                                                   'real_step_ms': real_step_ms,
                                                   'note': 'training_thread_busy: wall time of the epoch loop minus the time the training thread spent waiting (run-ahead bound on the GPU, '
                                                           'gradient all-reduce) -- what the host needs per step to feed a GPU of its own; process_cpu: CPU seconds of every thread of the process '
                                                           '(time.process_time), which also counts the runtime\'s and gloo\'s spin-waits; last epoch (every image resident in the HBM store). '
                                                           'The GPU is shared by the ranks here, so wall time per step is N x a real step and is not the figure of merit'}}
        print(json.dumps(out), flush=True)
    faulthandler.cancel_dump_traceback_later()
    dist.barrier(); dist.destroy_process_group()
    if a.dir is None:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == '__main__':
    main()
