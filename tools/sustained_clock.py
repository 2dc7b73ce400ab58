import os, sys, torch
sys.path.insert(0, os.getcwd())
from learning_embeddings_amd import ops
cin, hw, cout, r, st, pad = 128, 28, 128, 3, 1, 1
x = torch.randn(512, cin, hw, hw, device='cuda').contiguous(memory_format=torch.channels_last)
w = (torch.randn(cout, cin, r, r, device='cuda') / (cin * r * r) ** 0.5).contiguous(memory_format=torch.channels_last)
pl = ops.conv_f32x3_split_weights(w)
for name, fn in (('x3 fwd', lambda: ops.conv_f32x3_fwd(x, pl, st, pad)), ('native fwd', lambda: ops.conv_f32_fwd(x, w, st, pad))):
    torch.cuda.synchronize(); import time; time.sleep(1.0)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(201)]
    ev[0].record()
    for i in range(200):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    d = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(200)]
    print(name, 'us per launch: first 5', [round(v) for v in d[:5]], ' 20-25', [round(v) for v in d[20:25]], ' 100-105', [round(v) for v in d[100:105]], ' last 5', [round(v) for v in d[-5:]])
