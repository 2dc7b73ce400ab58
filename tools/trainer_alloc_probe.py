"""Through-trainer fp32 step, three measurements in one process, with the caching allocator's counters after each: shows whether the
varying CNN row count per batch (500-508 rows) keeps the allocator going back to hipMalloc."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
ns = argparse.Namespace(workload='cfg3', batch=None)
for i in range(3):
    r = bench.measure_trainer(ns, 'fp32', lambda s: None, 8, 3)
    st = torch.cuda.memory_stats()
    print('run %d: %.2f ms/step rows %.0f | device allocs %d frees %d retries %d reserved %.1f GB active %.1f GB' % (
        i, r['ms_per_step'], r['cnn_rows_per_step'], st['num_device_alloc'], st['num_device_free'], st['num_alloc_retries'],
        st['reserved_bytes.all.current'] / 1e9, st['active_bytes.all.peak'] / 1e9), flush=True)
