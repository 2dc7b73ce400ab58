"""Engine built WITHOUT a graph capture, eager steps only: ms/step (compare with tools/launch_modes_ab.py's eager blocks)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ap = argparse.ArgumentParser(); ap.add_argument('--batch', type=int, default=None); ap.add_argument('--no-reducer-hooks', action='store_true'); ap.add_argument('--flips', type=int, default=0); ap.add_argument('--sleep-us', type=int, default=0)
a = ap.parse_args()
from learning_embeddings_amd.engine import StepEngine
eng = StepEngine('cfg3', dtype='fp32', use_graph=False, batch=a.batch)
if a.no_reducer_hooks:
    eng.reducer.live = False
for _ in range(6):
    eng.step()
torch.cuda.synchronize()
imgs = torch.rand(128, 3, 224, 224, device=eng.device)
def pre():
    for i in range(a.flips):
        imgs[i % 128].flip(-1)
    if a.sleep_us:
        torch.cuda._sleep(int(a.sleep_us * 2000))
for b in range(3):
    t0 = time.perf_counter()
    for _ in range(12):
        pre()
        eng.step()
    torch.cuda.synchronize()
    print('pure eager block %d rows %d: %.2f ms/step' % (b, eng.n_rows, (time.perf_counter() - t0) / 12 * 1e3), flush=True)
st = torch.cuda.memory_stats()
print('reserved %.1f GB, peak active %.1f GB' % (st['reserved_bytes.all.current'] / 1e9, st['active_bytes.all.peak'] / 1e9))
eng.close()
