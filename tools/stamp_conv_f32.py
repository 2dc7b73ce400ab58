#!/usr/bin/env python3
"""Diagnostic (variants/liblecone_stamp.so: a build of conv_f32.hip with s_memtime stamps around the four segments of a K-loop iteration):
where a chunk's cycles go.  usage: LEC_LIB_PATH=variants/liblecone_stamp.so python tools/stamp_conv_f32.py [cin hw cout k stride pad]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning_embeddings_amd import ops
from learning_embeddings_amd._lib import lib
shape = [int(v) for v in sys.argv[1:7]] if len(sys.argv) >= 7 else [256, 14, 256, 3, 1, 1]
cin, hw, cout, r, st, pad = shape
N = 512
x = torch.randn(N, cin, hw, hw, device='cuda').contiguous(memory_format=torch.channels_last)
w = (torch.randn(cout, cin, r, r, device='cuda') / (cin * r * r) ** 0.5).contiguous(memory_format=torch.channels_last)
ho = (hw + 2 * pad - r) // st + 1
dy = torch.randn(N, cout, ho, ho, device='cuda').contiguous(memory_format=torch.channels_last)
out = (C.c_ulonglong * 8)()
for what in ('fwd', 'dgrad'):
    fn = (lambda: ops.conv_f32_fwd(x, w, st, pad)) if what == 'fwd' else (lambda: ops.conv_f32_dgrad(dy, w, x.shape, st, pad))
    fn(); torch.cuda.synchronize()
    lib.lec_debug_cf_stamps(out, 1)
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize()
    lib.lec_debug_cf_stamps(out, 1)
    ld, mm, stt, ba, n, tot, wgs = [int(v) for v in out[:7]]
    print('%s %s: %.0f us; per chunk (wave 0 of every workgroup, s_memtime ticks): issue loads %.0f, MFMA block %.0f, wait + LDS stores %.0f, barrier %.0f = %.0f; chunks %d, workgroups %d, ticks per workgroup %.0f (in the loop %.0f)'
          % (what, shape, a.elapsed_time(b) * 1e3, ld / n, mm / n, stt / n, ba / n, (ld + mm + stt + ba) / n, n, wgs, tot / wgs, (ld + mm + stt + ba) / wgs))
