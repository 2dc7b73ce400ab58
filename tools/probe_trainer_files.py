#!/usr/bin/env python3
"""Where the host's time goes in JointEmbeddings.train_epoch fed from image files (bench.py's through_trainer_files leg): wall time of the
DataLoader iterator's creation, of waiting for each batch, of the negative lookahead, and of the criterion / backward / update calls of
train_step (host enqueue time: nothing synchronises), per epoch.  usage: python tools/probe_trainer_files.py [--images 1024] [--workers 8]"""
import argparse, os, sys, tempfile, time, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

def main():
    global T, EV, MAIN, STAMPS
    STAMPS = []
    ap = argparse.ArgumentParser()
    ap.add_argument('--images', type=int, default=1024); ap.add_argument('--workers', type=int, default=8); ap.add_argument('--epochs', type=int, default=3)
    ap.add_argument('--no-thp', type=int, default=0); ap.add_argument('--ctx', default=None)
    ap.add_argument('--workload', default='cfg3'); ap.add_argument('--batch', type=int, default=None); ap.add_argument('--persistent', type=int, default=0)
    a = ap.parse_args()
    if a.no_thp:
        import ctypes
        print('prctl(PR_SET_THP_DISABLE) ->', ctypes.CDLL(None).prctl(41, 1, 0, 0, 0))
    try:
        print('THP:', open('/sys/kernel/mm/transparent_hugepage/enabled').read().strip())
    except Exception as e:
        print('THP: ?', e)
    import torch
    d = tempfile.mkdtemp(prefix='lec_probe_')
    paths = bench.write_image_files(d, a.images + 80)
    tr, crit, dl, cfg = bench._bench_trainer(a, 'fp32', a.images, 64, lambda j: paths[j], a.workers)
    B = cfg[2]
    T = {}
    def timed(name, fn):
        def w(*x, **k):
            t = time.perf_counter(); r = fn(*x, **k); T[name] = T.get(name, 0.0) + time.perf_counter() - t; return r
        return w
    crit.forward = timed('criterion.forward (host)', crit.forward)
    tr.apply_updates = timed('apply_updates (host)', tr.apply_updates)
    tr.image_store.resolve = timed('store.resolve', tr.image_store.resolve)
    tr.image_store.gather = timed('store.gather', tr.image_store.gather)
    tr.img_feat_net.forward_raw = timed('forward_raw (host)', tr.img_feat_net.forward_raw)
    orig_step = tr.train_step
    EV = []
    def step_with_events(item):
        a_ = torch.cuda.Event(enable_timing=True); b_ = torch.cuda.Event(enable_timing=True)
        a_.record(); r = orig_step(item); b_.record(); EV.append((a_, b_)); STAMPS.append(time.perf_counter()); return r
    tr.train_step = timed('train_step (host)', step_with_events)
    orig_bwd = torch.Tensor.backward
    torch.Tensor.backward = timed('loss.backward (host: waits for the autograd thread to enqueue)', orig_bwd)
    orig_fin = tr.reducer.finish
    tr.reducer.finish = timed('reducer.finish', orig_fin)
    import torch.utils.data.dataloader as dlm
    orig_iter = torch.utils.data.DataLoader.__iter__
    torch.utils.data.DataLoader.__iter__ = timed('DataLoader.__iter__ (worker start)', orig_iter)
    if a.ctx:
        tr.worker_context = a.ctx
    if a.persistent or a.ctx:
        tr.persistent_workers = bool(a.persistent); tr._make_train_loader()
    import threading, collections, traceback
    MAIN = threading.get_ident()
    def sampler(stop, hist):
        while not stop.is_set():
            fr = sys._current_frames().get(MAIN)
            if fr is not None:
                st = traceback.extract_stack(fr)[-4:]
                hist[' <- '.join('%s:%d %s' % (os.path.basename(f.filename), f.lineno, f.name) for f in reversed(st))] += 1
            time.sleep(0.004)
    for ep in range(a.epochs):
        T.clear(); tr.epoch = ep
        stop = threading.Event(); hist = collections.Counter()
        th = threading.Thread(target=sampler, args=(stop, hist), daemon=True); th.start()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        running, steps = tr.train_epoch()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print('epoch %d: %d steps, %.1f ms/step wall (host loop %.1f ms/step, final GPU drain %.1f ms)' % (ep + 1, steps, (t2 - t0) / steps * 1e3, (t1 - t0) / steps * 1e3, (t2 - t1) * 1e3))
        gpu = [x.elapsed_time(y) for x, y in EV]; EV.clear()
        print('    host wall between train_step returns (ms):', ' '.join('%.0f' % ((STAMPS[i + 1] - STAMPS[i]) * 1e3) for i in range(len(STAMPS) - 1))); STAMPS.clear()
        print('    GPU time between the events around train_step: mean %.1f ms, min %.1f, max %.1f' % (sum(gpu) / len(gpu), min(gpu), max(gpu)))
        for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
            print('    %-40s %8.1f ms total  %7.2f ms/step' % (k, v * 1e3, v / steps * 1e3))
        stop.set(); th.join()
        tot = sum(hist.values())
        for k, v in hist.most_common(6):
            print('    [main thread %4.1f %%] %s' % (100.0 * v / tot, k))
    shutil.rmtree(d, ignore_errors=True)


if __name__ == '__main__':
    main()
