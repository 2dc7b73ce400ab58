#!/usr/bin/env python3
"""The balanced (stream-K) form of the fp32 forward / stride-1 data gradient against the tile walk, layer by layer: ResNet-50's convolution
shapes at the rows one pass of the step carries (default 256: the bench's 512 rows run as two concurrent passes).  Per layer: tiles, rounds of
the 512 workgroup slots, microseconds and TFLOP/s with schedule 0 (tile walk), 2 (balanced wherever it applies) and (1) (the
launcher's choice).  usage: python tools/bench_conv_f32_balanced.py [--rows 256] [--iters 10] [--json out.json]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning_embeddings_amd import ops

ap = argparse.ArgumentParser(); ap.add_argument('--rows', type=int, default=256); ap.add_argument('--iters', type=int, default=10)
ap.add_argument('--json', default=None); ap.add_argument('--net', action='store_true', help='only the end-to-end ResNet-50 forward (the evaluation phase embeds images on ONE stream)')
a = ap.parse_args()
SHAPES = [('l1.c1a', 64, 56, 64, 1, 1, 0), ('l1.c2', 64, 56, 64, 3, 1, 1), ('l1.c3', 64, 56, 256, 1, 1, 0), ('l1.c1', 256, 56, 64, 1, 1, 0),
          ('l2.c1a', 256, 56, 128, 1, 1, 0), ('l2.c2s', 128, 56, 128, 3, 2, 1), ('l2.c3', 128, 28, 512, 1, 1, 0), ('l2.ds', 256, 56, 512, 1, 2, 0),
          ('l2.c1', 512, 28, 128, 1, 1, 0), ('l2.c2', 128, 28, 128, 3, 1, 1),
          ('l3.c1a', 512, 28, 256, 1, 1, 0), ('l3.c2s', 256, 28, 256, 3, 2, 1), ('l3.c3', 256, 14, 1024, 1, 1, 0), ('l3.ds', 512, 28, 1024, 1, 2, 0),
          ('l3.c1', 1024, 14, 256, 1, 1, 0), ('l3.c2', 256, 14, 256, 3, 1, 1),
          ('l4.c1a', 1024, 14, 512, 1, 1, 0), ('l4.c2s', 512, 14, 512, 3, 2, 1), ('l4.c3', 512, 7, 2048, 1, 1, 0), ('l4.ds', 1024, 14, 2048, 1, 2, 0),
          ('l4.c1', 2048, 7, 512, 1, 1, 0), ('l4.c2', 512, 7, 512, 3, 1, 1)]
# how often each shape occurs in ResNet-50 (forward; the data gradient of the same layer as often)
COUNT = {'l1.c1a': 1, 'l1.c2': 3, 'l1.c3': 4, 'l1.c1': 2, 'l2.c1a': 1, 'l2.c2s': 1, 'l2.c3': 4, 'l2.ds': 1, 'l2.c1': 3, 'l2.c2': 3,
         'l3.c1a': 1, 'l3.c2s': 1, 'l3.c3': 6, 'l3.ds': 1, 'l3.c1': 5, 'l3.c2': 5, 'l4.c1a': 1, 'l4.c2s': 1, 'l4.c3': 3, 'l4.ds': 1, 'l4.c1': 2, 'l4.c2': 2}


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3




def net_forward():
    """ResNet-50 -> D = 10, fp32, no_grad, one stream: images / s of the embedding forward with the tile walk and with the launcher's choice."""
    from learning_embeddings_amd.oe_h import FeatCNN
    torch.manual_seed(0)
    net = FeatCNN(None, output_dim=10, K=0.1).cuda()
    out = {}
    for rows, train in ((250, False), (10, True), (256, True), (512, True)):      # (10, train): the reference's own chunks of the 'train' phase (oe_h.py:1972)          # eval-mode batches of 250 (reference_exact_eval=False); train-mode statistics
        net.train(train)
        x = torch.rand(rows, 3, 224, 224, device='cuda')
        with torch.no_grad():
            timeit(lambda: net(x), 3)
            tm = {0: [], 1: []}
            for rep in range(3):
                for mode in (0, 1):
                    net.model.conv_schedule = mode                    # ResNet.conv_schedule -> the `schedule` argument of every launch of this forward
                    tm[mode].append(timeit(lambda: net(x), 5))
                    net.model.conv_schedule = -1
        a0, a1 = sorted(tm[0])[1], sorted(tm[1])[1]
        # the same forward replayed as a hipGraph (JointEmbeddings.embed_images does this from a chunk shape's third occurrence on)
        st = torch.cuda.Stream()
        with torch.no_grad():
            with torch.cuda.stream(st):
                net(x); net(x)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                y = net(x)
            gr = timeit(lambda: g.replay(), 10)
            del g, y
        out['rows_%d_%s' % (rows, 'train' if train else 'eval')] = {'tile_walk_ms': round(a0 / 1e3, 2), 'balanced_ms': round(a1 / 1e3, 2),
                                                                   'images_per_s_tile_walk': round(rows / a0 * 1e6), 'images_per_s_balanced': round(rows / a1 * 1e6),
                                                                   'replayed_ms': round(gr / 1e3, 2), 'images_per_s_replayed': round(rows / gr * 1e6)}
    return out


if a.net:
    r = net_forward(); print(json.dumps(r))
    if a.json:
        json.dump(r, open(a.json, 'w'), indent=1)
    sys.exit(0)
res = []; tot = {}
for name, cin, hw, cout, r, st, pad in SHAPES:
    N = a.rows
    x = torch.randn(N, cin, hw, hw, device='cuda').contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, r, r, device='cuda') / (cin * r * r) ** 0.5).contiguous(memory_format=torch.channels_last)
    ho = (hw + 2 * pad - r) // st + 1
    dy = torch.randn(N, cout, ho, ho, device='cuda').contiguous(memory_format=torch.channels_last)
    flops = 2.0 * N * ho * ho * cout * cin * r * r
    tf = -(-N * ho * ho // 128) * -(-cout // (64 if cout <= 64 else 128)); td = -(-N * hw * hw // 128) * -(-cin // (64 if cin <= 64 else 128))
    row = {'layer': name, 'gflop': round(flops / 1e9, 1), 'tiles_fwd': tf, 'rounds_fwd': round(tf / 512, 2), 'tiles_dgrad': td, 'rounds_dgrad': round(td / 512, 2)}
    # warm the clocks on this shape, then alternate the modes and keep each mode's median (the first launches after an idle gap run 5 - 9 % slow)
    timeit(lambda: ops.conv_f32_fwd(x, w, st, pad, want_stats=True), 2 * a.iters)
    tm = {m: {'f': [], 'd': []} for m in (0, 2, 1)}
    for rep in range(3):
        for mode in (0, 2, 1):
            prev = ops.fusion().schedule; ops.fusion().schedule = mode
            tm[mode]['f'].append(timeit(lambda: ops.conv_f32_fwd(x, w, st, pad, want_stats=True), a.iters))
            tm[mode]['d'].append(timeit(lambda: ops.conv_f32_dgrad(dy, w, x.shape, st, pad), a.iters) if st == 1 else float('nan'))
            ops.fusion().schedule = prev
    for mode in (0, 2, 1):
        t_f = sorted(tm[mode]['f'])[1]; t_d = sorted(tm[mode]['d'])[1]
        row['fwd_us_%d' % mode] = round(t_f, 1); row['dgrad_us_%d' % mode] = round(t_d, 1)
        row['fwd_tf_%d' % mode] = round(flops / t_f / 1e6, 1); row['dgrad_tf_%d' % mode] = round(flops / t_d / 1e6, 1)
        tot[mode] = tot.get(mode, 0.0) + COUNT[name] * (t_f + (t_d if st == 1 else 0.0))
    res.append(row); print(json.dumps(row), flush=True)
    del x, w, dy; torch.cuda.empty_cache()
print(json.dumps({'weighted_total_ms': {('mode_%d' % m): round(v / 1e3, 2) for m, v in tot.items()}, 'rows': a.rows}))
if a.json:
    json.dump({'layers': res, 'weighted_total_ms': {('mode_%d' % m): round(v / 1e3, 2) for m, v in tot.items()}, 'rows': a.rows}, open(a.json, 'w'), indent=1)
