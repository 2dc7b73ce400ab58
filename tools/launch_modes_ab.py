"""Same box, same engine: ms/step of the cfg3 fp32 step replayed as a hipGraph against eager launches, in alternating blocks, and
(optionally) the trainer-API step.  Prints per-block means and the per-step wall times of the eager blocks (host jitter shows there).
    python tools/launch_modes_ab.py [--blocks 3] [--steps 12] [--trainer]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--blocks', type=int, default=3)
    ap.add_argument('--steps', type=int, default=12)
    ap.add_argument('--dtype', default='fp32')
    ap.add_argument('--trainer', action='store_true')
    ap.add_argument('--batch', type=int, default=None)
    ap.add_argument('--side-priority', type=int, default=None)
    ap.add_argument('--main-high', action='store_true', help='run every step on a high-priority stream')
    a = ap.parse_args()
    from learning_embeddings_amd.engine import StepEngine
    eng = StepEngine('cfg3', dtype=a.dtype, use_graph=True, batch=a.batch)
    print('rows per step', eng.n_rows, flush=True)
    if a.side_priority is not None and eng.overlap is not None:
        eng.overlap.side = torch.cuda.Stream(priority=a.side_priority)
    if a.main_high:
        hs = torch.cuda.Stream(priority=-1)
        hs.wait_stream(torch.cuda.current_stream())
        ctx = torch.cuda.stream(hs); ctx.__enter__()
    for _ in range(8):
        eng.step()
    torch.cuda.synchronize()
    assert eng.hip_graph is not None, eng.graph_error
    def block(graph):
        eng.set_launch_mode(graph)
        for _ in range(2):
            eng.step()
        torch.cuda.synchronize()
        per = []
        t0 = time.perf_counter()
        for _ in range(a.steps):
            th = time.perf_counter(); eng.step(); per.append((time.perf_counter() - th) * 1e3)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.steps * 1e3, per
    for b in range(a.blocks):
        for graph in (True, False):
            ms, per = block(graph)
            print('block %d %-6s %8.2f ms/step   host enqueue per step: mean %.1f max %.1f ms' % (b, 'graph' if graph else 'eager', ms, sum(per) / len(per), max(per)), flush=True)
    eng.set_launch_mode(True)
    eng.close()
    if a.trainer:
        import bench
        ns = argparse.Namespace(workload='cfg3', batch=None)
        for _ in range(2):
            r = bench.measure_trainer(ns, a.dtype, lambda s: None, a.steps, 3)
            print('trainer eager %8.2f ms/step  rows %.0f  -> %.2f ms per 512 rows' % (r['ms_per_step'], r['cnn_rows_per_step'], r['ms_per_step'] * 512 / r['cnn_rows_per_step']), flush=True)


if __name__ == '__main__':
    main()
