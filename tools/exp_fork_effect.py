#!/usr/bin/env python3
"""Does fork() of a process with a live HIP context slow its later steps down, and on which side?  StepEngine steps timed in three launch modes
(hipGraph replay: the host does almost nothing; eager: ~25 ms of Python / ctypes launches per step) before and after forking 8 children that
just sleep (what starting DataLoader workers does).  usage: python tools/exp_fork_effect.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning_embeddings_amd import miopen_tuning
miopen_tuning.setup()
from learning_embeddings_amd.engine import StepEngine


def timeit(eng, n=8):
    for _ in range(2):
        eng.step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        eng.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def minor_faults():
    with open('/proc/self/stat') as f:
        return int(f.read().split()[9])

eng = StepEngine('cfg3', dtype='fp32', use_graph=True)
for _ in range(6):
    eng.step()
while eng.hip_graph is None and eng.graph_error is None:
    eng.step()
res = {}
for phase in ('before fork', 'after fork (children alive)', 'children gone'):
    if phase.startswith('after'):
        kids = []
        for _ in range(8):
            pid = os.fork()
            if pid == 0:
                time.sleep(30); os._exit(0)
            kids.append(pid)
    if phase.startswith('children gone'):
        import signal
        for pid in kids:
            os.kill(pid, signal.SIGKILL); os.waitpid(pid, 0)
    for mode, graph in (('hipGraph replay', True), ('eager launches', False)):
        eng.set_launch_mode(graph)
        for rep in range(2):
            f0 = minor_faults(); ms = timeit(eng); f1 = minor_faults()
            print('%-28s %-16s run %d: %7.1f ms/step, %6d minor page faults per step' % (phase, mode, rep, ms, (f1 - f0) / 10), flush=True)
eng.close()
