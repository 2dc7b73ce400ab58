import torch, time
dev='cuda'
P, n, H = 512, 512, 224
pool4 = torch.rand(P, H, H, 4, device=dev)
pool_cl = pool4.permute(0,3,1,2)   # channels_last view
idx = torch.randint(0, P, (n,), device=dev)
def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/it*1e3
print('index_select on channels_last 4d      %.0f us' % t(lambda: pool_cl.index_select(0, idx)), pool_cl.index_select(0, idx).is_contiguous(memory_format=torch.channels_last))
print('index_select on plain [P,H,W,4]       %.0f us' % t(lambda: pool4.index_select(0, idx)))
p2 = pool4.view(P, -1)
print('index_select on [P, HW4]              %.0f us' % t(lambda: p2.index_select(0, idx)))
print('advanced indexing pool4[idx]          %.0f us' % t(lambda: pool4[idx]))
p16 = pool4.view(P, -1, 4)
print('index_select on [P, HW, 4]            %.0f us' % t(lambda: p16.index_select(0, idx)))
out = torch.empty(n, H, H, 4, device=dev)
print('index_select out=                     %.0f us' % t(lambda: torch.index_select(pool4, 0, idx, out=out)))
print('copy 411 MB (clone)                   %.0f us' % t(lambda: out.copy_(pool4)))
