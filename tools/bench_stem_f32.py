"""The fp32 stem forward kernel against the generic one (us per launch at 256 / 512 rows of 224 x 224 images): python tools/bench_stem_f32.py"""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learning_embeddings_amd import ops
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)*1000/iters
for N in (256, 512):
    x=torch.rand(N,4,224,224,device='cuda').contiguous(memory_format=torch.channels_last); x[:,3]=0
    w=(torch.randn(64,4,7,7,device='cuda')/12).contiguous(memory_format=torch.channels_last); w[:,3]=0
    t1=timeit(lambda: ops.conv_f32_stem_fwd(x,w,want_stats=True)); ops._BN_WS_OWNER[0]=0
    t0=timeit(lambda: ops.conv_f32_fwd(x,w,2,3,want_stats=True)); ops._BN_WS_OWNER[0]=0
    print('rows %d: stem kernel %.1f us, generic %.1f us' % (N, t1, t0))
for N in (256, 512):
    x=torch.rand(N,4,224,224,device='cuda').contiguous(memory_format=torch.channels_last); x[:,3]=0
    dy=torch.randn(N,64,112,112,device='cuda').contiguous(memory_format=torch.channels_last)
    dw=torch.zeros(64,3,7,7,device='cuda').contiguous(memory_format=torch.channels_last)
    print('rows %d: weight gradient (lec_conv_f32_wgrad_c3; LEC_CF_STEM=0 for the generic kernel) %.1f us' % (N, timeit(lambda: ops.conv_f32_wgrad_c3(dy, x, dw, 2, 3))))
