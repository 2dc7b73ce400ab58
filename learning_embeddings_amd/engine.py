"""StepEngine: the joint training step on synthetic inputs, shaped for MI355X (used by bench.py, smoke() and tests).

One step = oe_h.py:1734-1774 for one batch of B (label, image) positives per GPU:

    host   : negatives of step s+1 drawn on a thread (bit-exact MT19937 stream) while the GPU runs step s
    stream : images[B,3,H,W] -> ResNet fwd (NHWC, bf16 autocast, fp32 master weights) -> raw feats [B,D] fp32
             -> ONE fused kernel (projections + (1+2K)B cone energies + hinge + loss + d/d table + d/d feats)
             -> ResNet bwd  -> bucketed SUM all-reduce (RCCL, overlapped with bwd; table gradient rides along)
             -> ONE table-step kernel (lambda-rescale + Adam + clip) + ONE flat-arena Adam kernel
    no host<->device synchronisation anywhere in the step.

Launch mode: after a few eager steps the whole forward + loss + backward (both HIP streams, ~700 launches) is captured
into ONE hipGraph and replayed per step (`use_graph`); index uploads, the gradient all-reduce and the two optimizer
launches stay outside (the Adam step number is a launch argument).  Replays make the step immune to host-side jitter
(the eager step needs 18-26 ms of host time against 49 ms of GPU time, and a busy host turns that around).

Workloads: hierarchy S3 = [8,64,384,1544] (or the real ETHEC DAG), image j hangs under leaf j mod n_leaf,
positive b of a step pairs image (step*B_global + b) mod M with its ancestor at level b mod L.
"""
import os
import time
import numpy as np
import torch

from . import _lib, ops, parallel
from .hierarchy import NegativeGraph, SyntheticLabelMap, SYNTHETIC
from .oe_h import Embedder, FeatCNN18, FeatCNN, EuclideanConesWithImagesHypernymLoss
from .resnet import WgradOverlap
from . import resnet as resnet_mod

# BASELINE.json configs[3]: ETHEC, resnet50, multi-level cross-entropy head over the 723 labels, batch 512 per GPU
CLASSIFIER_WORKLOADS = {'cfg4': ('ETHEC', 'resnet50', 512, 224), 'tiny4': ('ETHEC', 'resnet18', 8, 32)}

WORKLOADS = {
    # name: (hierarchy, arch, per-GPU batch, K negatives ratio, D, image hw)
    'cfg2': ('ETHEC', 'resnet18', 128, 5, 10, 224),     # BASELINE.json configs[1]
    'cfg3': ('S3', 'resnet50', 256, 5, 10, 224),        # configs[2]: the config the headline metric is quoted on (hierarchy S3)
    'cfg3_ethec': ('ETHEC', 'resnet50', 256, 5, 10, 224),   # the same step over the real ETHEC label DAG (723 labels) instead of S3's 2 000
    'cfg5': ('S5', 'resnet50', 256, 256, 10, 224),      # configs[4]: 50k-node hierarchy, 256 negatives / positive
    'tiny': ('S1x', 'resnet18', 8, 4, 10, 32),          # smoke / tests
}


def max_rows_f32(hw):
    """Rows per CNN pass the fp32 convolutions take (32-bit byte offsets: every activation below 2 GiB; the stem's output is the largest)."""
    return ((1 << 31) - 1) // (max(hw // 2, 1) ** 2 * 64 * 4)


def make_labelmap(name):
    if name == 'ETHEC':
        return SyntheticLabelMap.ethec()
    if name == 'S1x':
        return SyntheticLabelMap([2, 4, 8])
    return SyntheticLabelMap(SYNTHETIC[name])



def _resident_pool(pool, device, compute_dtype, backbone, rows):
    """The synthetic image pool as it sits in HBM: channels_last, in the compute dtype.  For the fp32 backbone on liblecone's convolutions the pixels
    are stored as FOUR floats -- the 16-byte pixel with a zero 4th channel that the f32 stem reads (resnet._pad_c4; the image store's gather emits the
    same layout, image_store.gather(c_out=4)) -- so a step's `index_select` produces the stem's operand itself.  Before, every pass started with a strided
    zero-fill and a strided 3 -> 4 channel copy of its rows behind the gather: ~0.6 ms at the head of the step with nothing to overlap with."""
    from . import resnet
    pool = pool.to(device).contiguous(memory_format=torch.channels_last)
    if compute_dtype != torch.float32:
        pool = pool.to(compute_dtype)                         # the backbone's first op would cast it anyway; same values
        conv1 = getattr(backbone, 'conv1', None)
        if (compute_dtype == torch.bfloat16 and resnet.CONV_BF16 != '0' and isinstance(conv1, resnet.Conv2d) and conv1.in_channels == 3 and conv1.bias is None
                and getattr(backbone, 'wgrad_overlap', None) not in (None, False) and rows * pool.shape[2] * pool.shape[3] * 16 < (1 << 31)):
            pool = resnet._pad_c8(pool)                       # the bf16 stem's operand itself: 16-byte pixels, zero channels 3..7 (no pad / cast kernels per step)
        return _rows_first(pool)
    conv1 = getattr(backbone, 'conv1', None)
    ov = getattr(backbone, 'wgrad_overlap', None)
    if ov is None:
        ov = resnet.WgradOverlap.instance                       # (the process default, as ops.overlap() resolves it)
    own_stem = (resnet.MFMA_F32 and isinstance(conv1, resnet.Conv2d) and conv1.in_channels == 3 and conv1.bias is None and ov not in (None, False)
                and ov.enabled and rows * pool.shape[2] * pool.shape[3] * 16 < (1 << 31) and os.environ.get('LEC_POOL_C4', '1') != '0')     # (0: A/B runs)
    return _rows_first(resnet._pad_c4(pool) if own_stem else pool)


def _rows_first(pool_cl):
    """[P, C, H, W] channels_last -> the same memory as a plain [P, H, W, C] tensor: what `_gather_images` selects rows of."""
    return pool_cl.permute(0, 2, 3, 1)


def _gather_images(pool, idx):
    """The images `idx` of the resident pool as an [n, C, H, W] channels_last batch.  `index_select` on the plain [P, H, W, C] tensor is a row copy
    (132 us for 512 x 224 x 224 x 4 floats, 6.2 TB/s, `tools/microbench/gather_rows_bench.py`); on the channels_last [P, C, H, W] view of the same memory it takes
    365 us AND returns an NCHW-contiguous tensor that the backbone then converts back (another ~150 us per pass)."""
    if os.environ.get('LEC_GATHER_ROWS', '1') == '0':             # A/B: the former path
        return pool.permute(0, 3, 1, 2).index_select(0, idx)
    return pool.index_select(0, idx).permute(0, 3, 1, 2)

class StepEngine:
    def __init__(self, workload='cfg3', n_images=4096, pool_images=None, dtype='bf16', lr=1e-4, alpha=0.01, K_cone=0.1,
                 sampler_mode='replicated', seed=0, batch=None, device=None, overlap_wgrad=None, use_graph=False,
                 graph_after=3, table_dtype='fp32', cnn_chunk=None, passes=None):
        hier, arch, B, K, D, hw = WORKLOADS[workload]
        self.workload, self.arch, self.B, self.K, self.D, self.hw = workload, arch, batch or B, K, D, hw
        self.rank, self.local_rank, self.world = parallel.init_process_group()
        self.device = device or torch.device('cuda', self.local_rank % max(torch.cuda.device_count(), 1))
        torch.cuda.set_device(self.device)
        self.lr, self.alpha, self.K_cone = lr, alpha, K_cone
        self.compute_dtype = {'bf16': torch.bfloat16, 'fp16': torch.float16, 'fp32': torch.float32}[dtype]
        self.labelmap = lm = make_labelmap(hier)
        self.N, self.L = lm.n_classes, len(lm.levels)
        self.M = n_images
        # image j hangs under leaf j mod n_leaf.  With far fewer images than leaves (config 5: 39 078 leaves) that
        # rule puts every image under the first top-level node, whose slot-L draw then has NO candidate (random.choice
        # raises in the reference too): spread the images over the leaves instead.
        nleaf_ = lm.levels[-1]
        jj = np.arange(n_images, dtype=np.int64)
        self.img_leaf = (jj % nleaf_) if 2 * n_images > nleaf_ else (jj * nleaf_) // n_images
        self.graph = NegativeGraph.from_labelmap(lm, image_leaf=lm.level_start[-1] + self.img_leaf, pick_per_level=True,
                                                 seed=seed + (0 if sampler_mode == 'replicated' else self.rank))
        # ancestors of every leaf per level (for the positives)
        par = lm.parents()
        leaf0, nleaf = lm.level_start[-1], lm.levels[-1]
        self.leaf_anc = np.zeros((nleaf, self.L), dtype=np.int32)
        for i in range(nleaf):
            v = leaf0 + i; chain = [v]
            while v in par:
                v = par[v][0]; chain.append(v)
            if len(chain) != self.L:
                raise ValueError('leaf %d does not have one ancestor per level' % (leaf0 + i))
            self.leaf_anc[i] = np.array(chain[::-1], dtype=np.int32)
        # With pick_per_level the `u`-fixed draw of pass p lands in slot p % (L+1); slot L with a label `u` draws an IMAGE
        # (oe_h.py:893-898), so every positive brings `cnt` extra images through the CNN each step.
        self.img_passes = [p for p in range(K) if p % (self.L + 1) == self.L]
        self.cnt = len(self.img_passes)
        self.n_rows = self.B * (1 + self.cnt)                            # CNN batch per step (fixed shape)
        # one BatchNorm batch per step means every activation of the n_rows images is alive at once: 0.045 GB per image
        # measured for ResNet-50 at 224x224 in bf16 (22.9 GB at 512 rows, 80.3 GB at 1 856).  Config 5 (K = 256 over 8 levels draws 28
        # image negatives per positive: 7 424 rows at B = 256) does not fit 288 GB; say so instead of dying in hipMalloc.
        per_row_gb = (0.05 if arch == 'resnet50' else 0.015) * (hw / 224.0) ** 2 * (2 if dtype == 'fp32' else 1)
        # cnn_chunk: push the step's CNN rows through the backbone `cnn_chunk` rows at a time -- a forward of every chunk without
        # saved activations, the loss on all raw outputs, then per chunk a second forward + backward (activations of ONE chunk alive at a
        # time; BatchNorm statistics per chunk, as the reference's own separate forwards of positives and negatives have them,
        # oe_h.py:980-985,1003-1009).  Picked automatically when the whole batch would not fit.
        self.cnn_chunk = cnn_chunk
        if self.cnn_chunk is None and self.n_rows * per_row_gb > 240:
            # Chunk size by measurement of the own kernels (round 6, config 5 = 7 424 rows, bf16, same box): 15 x 512 rows (256 padding rows) 603.8 ms,
            # 16 x 464 587.6, 8 x 928 553.2 -- larger chunks amortise the per-chunk launch boundaries and fill the chip better, and a chunk size that
            # DIVIDES the row count leaves no padding rows in the last chunk's BatchNorm batch.  Bounds: the 2 GiB tensor limit of the
            # 32-bit offsets (bf16: 1 337 rows of ResNet-50 at 224 x 224) and ~50 GB of activations per chunk.
            cap = 1024 if dtype != 'fp32' else min(512, max_rows_f32(hw))
            divs = [d for d in range(cap, 255, -1) if self.n_rows % d == 0]
            self.cnn_chunk = divs[0] if divs else min(cap, 512)
        # liblecone's fp32 convolutions address a tensor with 32-bit byte offsets (conv_geo.h conv_check): the largest activation of the
        # backbone (the stem's output, (hw/2)^2 x 64 floats per row) must stay below 2 GiB -- 668 rows of ResNet at 224 x 224
        rows_2g = max_rows_f32(hw)
        if dtype == 'fp32' and self.cnn_chunk is None and self.n_rows > rows_2g:
            self.cnn_chunk = min(512, rows_2g)
        if self.cnn_chunk is not None and self.cnn_chunk >= self.n_rows:
            self.cnn_chunk = None
        # chunks are all the same size (one set of kernel shapes): the row list is padded with repeats of POOL IMAGE 0 (idx_dev is
        # zero-initialised), whose outputs no pair references (zero gradient).  They do enter the last chunk's BatchNorm batch
        # statistics and running statistics, as any image of that chunk does.
        self.n_rows_pad = self.n_rows if self.cnn_chunk is None else -(-self.n_rows // self.cnn_chunk) * self.cnn_chunk
        if self.cnn_chunk is None and self.n_rows * per_row_gb > 240:
            raise ValueError('workload %s at B=%d pushes %d images through %s per step (%d image negatives per positive): about %.0f GB of '
                             'activations, more than one MI355X holds; pass a smaller `batch` (e.g. %d)'
                             % (workload, self.B, self.n_rows, arch, self.cnt, self.n_rows * per_row_gb, max(1, int(200 / per_row_gb / (1 + self.cnt)))))
        # Concurrent passes (fp32): the step's CNN rows go through the backbone as `passes` equal parts -- with one image negative per
        # positive: the positives' images | the image negatives, the reference's own split into separate forwards (oe_h.py:980-985,
        # 1003-1009: separate BatchNorm batches) -- each on its own HIP stream.  At fp32 the convolutions are bound by the matrix pipe
        # and the BatchNorm passes by HBM: with ONE pass they alternate on one stream and the step is the sum of its kernels; with two,
        # one pass's BatchNorm traffic runs under the other's convolutions (ResNet-50 fwd + bwd of 512 rows: 145.2 -> 133.7 ms,
        # tools/exp_two_streams_f32.py).  The weight gradients then run in line (a third stream adds nothing once the matrix pipe is
        # busy all the time: 137.1 ms).  The bf16 stack is HBM-bound everywhere and keeps one pass (measured in round 1: 53.5 vs 51.8 ms).
        if passes is None:
            # (bf16 too since round 6: with the own bf16 convolution family -- bound by L2 request throughput in layer3 / layer4, not by HBM -- two concurrent
            # half-batch passes hide BatchNorm traffic as at fp32: 40.35 -> 38.56 ms per step, same box; the round-1 measurement on library kernels said the opposite)
            passes = 2 if (dtype in ('fp32', 'bf16') and self.cnn_chunk is None and self.n_rows % 2 == 0 and self.n_rows >= 16
                           and not (dtype == 'bf16' and overlap_wgrad)) else 1      # (bf16 with side-stream weight gradients asked for: the one-pass arrangement)
        if passes > 1 and (self.cnn_chunk is not None or self.n_rows % passes):
            raise ValueError('passes=%d needs an unchunked step whose %d CNN rows divide evenly' % (passes, self.n_rows))
        self.passes = int(passes)
        if overlap_wgrad is None:
            # (the chunked step keeps its weight gradients in line: same box, config 5, 645.4 / 646.0 ms with the side stream, 639.3 without -- inside a replayed chunk
            # graph the side stream's kernels mostly stretch the BatchNorm finalize launches of the main chain)
            overlap_wgrad = self.passes == 1 and self.cnn_chunk is None
        torch.manual_seed(0)                                              # oe_h.py:1338: table init from seed 0
        self.criterion = EuclideanConesWithImagesHypernymLoss(lm, K, {}, alpha, pick_per_level=True, K=K_cone, use_CNN=True)
        self.model = Embedder(D, lm, None, K=K_cone).to(self.device)
        cls = FeatCNN if arch == 'resnet50' else FeatCNN18
        self.img_feat_net = cls(image_dir='', output_dim=D, K=K_cone, compute_dtype=self.compute_dtype).to(self.device)
        self.img_feat_net.train(); self.model.train()
        self.arena = parallel.FlatArena(self.img_feat_net.parameters(), self.device)
        w = self.model.embeddings.weight
        self.table = w.data
        self.table_grad = torch.zeros_like(self.table)
        self.table_m = torch.zeros_like(self.table); self.table_v = torch.zeros_like(self.table); self.table_step = 0
        # BASELINE.json config 5 ("fp16+fp32-master"): the loss kernel reads the label rows from a 2-byte shadow; gradients, Adam
        # moments and the master stay fp32, the table-step kernel refreshes the shadow
        self.table_h = self.table.to(torch.float16) if table_dtype == 'fp16' else None
        self.reducer = parallel.GradientReducer(self.arena, extra=[self.table_grad])
        if self.world > 1:
            torch.distributed.broadcast(self.arena.data, 0); torch.distributed.broadcast(self.table, 0)
        # low-precision shadow weights (no per-layer cast kernels) + gradients written straight into the arena; the
        # weight-gradient kernels optionally run on a second stream (overlap_wgrad)
        self.overlap = None
        if self.compute_dtype in (torch.bfloat16, torch.float32):
            if self.compute_dtype == torch.bfloat16:
                self.arena.enable_lowp_transposed()     # bf16 shadow + its transposed twin (the data gradients' operand)
            # fp32 (the reference's precision): the convolutions read the arena's master weights directly
            self.overlap = WgradOverlap(self.reducer, self.arena, side_stream=overlap_wgrad)
        # this engine's settings travel with ITS backbone (resnet.ResNet.wgrad_overlap / bn_grad_accumulate / conv_schedule -> the
        # FusionContext of every forward): no process-wide switch is touched, a second engine or trainer in the process keeps its own
        self.backbone = self.img_feat_net.model
        self.backbone.wgrad_overlap = self.overlap if self.overlap is not None else False
        # synthetic image pool resident in HBM: torch.rand in [0,1) like ToTensor output (oe_h.py:1463-1471), seed 0
        P = pool_images or min(n_images, 2 * self.B)
        g = torch.Generator(device='cpu').manual_seed(1234 + self.rank)
        pool = torch.rand(P, 3, hw, hw, generator=g)
        self.pool = _resident_pool(pool, self.device, self.compute_dtype, self.backbone, self.cnn_chunk or self.n_rows_pad)    # (rows of ONE gathered batch: a chunk, or the step)
        self.P = P
        self.gfeat = torch.zeros(self.n_rows_pad, D, device=self.device)
        self.pin = [torch.empty((self.B, 2 + 2 * K), dtype=torch.int32).pin_memory() for _ in range(2)]
        self.pin_img = [torch.empty(self.n_rows, dtype=torch.int64).pin_memory() for _ in range(2)]
        self.pin_ev = [None, None]
        self.step_no = 0
        self.host_wait_s = 0.0                       # host time spent waiting for the GPU (run-ahead bound), for bench.py
        self._in_flight = []                         # completion events of the steps the host has enqueued and the GPU has not finished
        self.loss_acc = torch.zeros((), device=self.device)
        self.prefetch = parallel.NegativePrefetcher(self.graph, self.positives, K, mode=sampler_mode)
        self.timers = None
        # static device inputs of the step (the captured graph reads these addresses)
        self.codes_dev = torch.zeros((self.B, 2 + 2 * K), dtype=torch.int32, device=self.device)
        self.idx_dev = torch.zeros(self.n_rows_pad, dtype=torch.int64, device=self.device)
        self.use_graph = bool(use_graph) and self.cnn_chunk is None      # the chunked step replays ONE graph per chunk instead (use_chunk_graph)
        self.use_chunk_graph = bool(use_graph) and self.cnn_chunk is not None
        self.chunk_graph = None
        if self.passes > 1 and self.overlap is not None and self.overlap.side is not None:
            # concurrent passes AND a weight-gradient side stream (an opt-in combination that measures slower: 130.5 against 125 ms): capturing it
            # aborts inside the runtime (the side stream is forked from two capturing pass streams at once), so this combination launches eagerly
            self.use_graph = False
        # A chunk of the chunked step as `chunk_lanes` concurrent parts (one HIP stream and one BatchNorm batch each), like the passes of the unchunked step: bf16,
        # even chunk sizes.  LEC_CHUNK_LANES=1: one part per chunk (A/B runs).
        self.chunk_lanes = 1
        if self.cnn_chunk is not None and self.compute_dtype == torch.bfloat16 and self.cnn_chunk % 2 == 0 and self.overlap is not None and self.overlap.side is None:
            self.chunk_lanes = int(os.environ.get('LEC_CHUNK_LANES', '2'))
        if self.cnn_chunk is not None and self.overlap is not None:
            self.overlap.accumulate = self.chunk_lanes == 1          # (lanes: BatchNorm parameter gradients are ADDED by the kernels themselves, atomics, as in the multi-pass step)
        self.pass_streams = [torch.cuda.Stream() for _ in range(max(self.passes, self.chunk_lanes))] if max(self.passes, self.chunk_lanes) > 1 else []
        if self.chunk_lanes > 1:
            self.backbone.bn_grad_accumulate = True
        if self.passes > 1:
            # several backward passes add into the same gradient slots from concurrent streams: BatchNorm's d gamma / d beta are ADDED
            # with atomics (like the weight gradients); the arena is zeroed once per step
            self.backbone.bn_grad_accumulate = True
        self.graph_after = graph_after
        self.hip_graph = None
        self.graph_out = None
        self.graph_error = None if (self.use_graph or not use_graph or self.cnn_chunk is not None) else 'not captured: concurrent passes with a weight-gradient side stream launch eagerly'
        self._graph_saved = None
        # With liblecone's own RCCL layer (LEC_DP_BACKEND=lecone) the bucket all-reduces are plain stream-ordered RCCL calls and are
        # captured INTO the step graph on the reducer's launch stream, where they overlap the rest of backward; torch.distributed's
        # collectives cannot be (this PyTorch build refuses the external events that would order them against a replay), so with
        # that backend the buckets are reduced after the replay.
        self.graph_reduces = self.reducer.comm is not None

    # global positives of step s: image (s*Bg + b) mod M with its ancestor at level b mod L   
    def positives(self, s):
        Bg = self.B * self.world
        b = np.arange(Bg, dtype=np.int64)
        j = (s * Bg + b) % self.M
        leaf = self.img_leaf[j]
        frm = self.leaf_anc[leaf, b % self.L].astype(np.int32)
        to = (self.N + j).astype(np.int32)
        return frm, to

    def enable_timers(self):
        mk = lambda: torch.cuda.Event(enable_timing=True)
        self.timers = {'records': [], 'mk': mk}
        self.kernel_timers(True)

    def kernel_timers(self, on):
        """Per-launch HIP events of this engine's BatchNorm / fp32 convolution kernels (ResNet.step_timers): True = start with empty lists,
        False = stop.  Returns the lists' dict (or None)."""
        self.backbone.step_timers = {'bn': [], 'conv': []} if on else None
        return self.backbone.step_timers

    def step(self):
        B, K = self.B, self.K
        if self.step_no == 0:
            resnet_mod.library_launches(reset=True)             # step 0 is always launched eagerly: its count is the step's (a replayed graph repeats it)
        frm, to, neg = self.prefetch.next()
        slot = self.step_no & 1
        if self.pin_ev[slot] is not None:
            t_w = time.perf_counter()
            self.pin_ev[slot].synchronize()          # the H2D copies of step s-2 have left this pinned buffer (also bounds run-ahead)
            self.host_wait_s += time.perf_counter() - t_w
        if len(self._in_flight) >= 2:
            t_w = time.perf_counter()
            self._in_flight.pop(0).synchronize()
            self.host_wait_s += time.perf_counter() - t_w
        pin = self.pin[slot]
        # CNN batch rows: [0, B) the positives' images; row B + b*cnt + i the image drawn as negative in pass img_passes[i]
        hp = pin.numpy()
        hp[:, 0] = frm
        hp[:, 1] = -1 - np.arange(B, dtype=np.int32)
        hp[:, 2:] = neg
        rows = self.pin_img[self.step_no & 1].numpy()
        rows[:B] = (to - self.N) % self.P
        is_img = neg >= self.N
        if self.cnt:
            cols = np.asarray(self.img_passes)
            sub = neg[:, cols]
            if not (sub >= self.N).all() or int(is_img.sum()) != B * self.cnt:
                raise RuntimeError('unexpected negative layout: image negatives outside the slot-L passes')
            rows[B:] = ((sub - self.N) % self.P).reshape(-1)
            hp[:, 2 + cols] = -1 - (B + np.arange(B, dtype=np.int32)[:, None] * self.cnt + np.arange(self.cnt, dtype=np.int32)[None, :])
        elif is_img.any():
            raise RuntimeError('unexpected image negative')
        self.codes_dev.copy_(pin, non_blocking=True)
        self.idx_dev[:self.n_rows].copy_(self.pin_img[self.step_no & 1], non_blocking=True)
        self.pin_ev[slot] = torch.cuda.Event(); self.pin_ev[slot].record()

        T = self.timers
        if (self.use_graph and self.hip_graph is None and self._graph_saved is None and self.graph_error is None
                and self.step_no >= self.graph_after):
            self._capture()
        if self.hip_graph is not None:
            ev = [T['mk']() for _ in range(4)] if T is not None else None
            if ev: ev[0].record()
            self.hip_graph.replay()
            if ev: ev[1].record()
            loss, e_pos, e_neg = self.graph_out
            if not self.graph_reduces:
                self.reducer.reduce_now()
            if ev: ev[2].record()
        else:
            ev = [T['mk']() for _ in range(6)] if T is not None else None
            loss, e_pos, e_neg = self._core(ev)
            self.reducer.finish()
            if ev: ev[4].record()
        self.table_step += 1
        ops.table_step_adam(self.table, self.table_grad, self.table_m, self.table_v, self.table_step, self.lr, self.K_cone,
                            table_f16=self.table_h)
        self.arena.adam_step(self.lr)
        if ev:
            ev[-1].record(); T['records'].append(ev)
        self.loss_acc += loss[0]
        if self.step_no == 0:
            self.library_conv_launches_per_step = dict(resnet_mod.library_launches())    # convolutions / GEMMs handed to MIOpen / hipBLASLt by this step: 0 on liblecone's paths
        self.step_no += 1
        self.last = (loss, e_pos, e_neg, frm, to, neg)
        # at most two whole steps in flight (the pinned-buffer wait above bounds the run-ahead by the START of step s - 2 only): what the side
        # stream has touched returns to the allocator when its events pass, so every extra step of run-ahead holds one more step of activations
        done = torch.cuda.Event(); done.record(); self._in_flight.append(done)
        return loss

    def _chunk_body(self):
        """ONE chunk of the chunked step on static device inputs (idx_chunk, win_dev): gather, backbone forward, the fused loss over the pairs whose image lies
        in the chunk (lec_joint_loss_fwd_bwd_window reading its window from device memory), backbone backward with the gradient that launch left.  This is the region
        the per-chunk hipGraph captures: the same graph is replayed for every chunk of every step."""
        codes = self.codes_dev
        pos_from = codes[:, 0].contiguous(); pos_to = codes[:, 1].contiguous(); negc = codes[:, 2:].contiguous()
        if self.chunk_lanes > 1:
            return self._chunk_body_lanes(pos_from, pos_to, negc)
        f = self.img_feat_net.forward_raw(_gather_images(self.pool, self.idx_chunk))
        self.feats_c.copy_(f.detach())
        self.gfeat_c.zero_()
        l_c, _, _ = ops.joint_loss_raw(self.table, self.feats_c, pos_from, pos_to, negc, None, self.K_cone, self.alpha, _lib.ENERGY_HYP_CONE,
                                       _lib.LABEL_HYP, _lib.IMAGE_SOFTCLIP, self.table_grad, self.gfeat_c, table_f16=self.table_h,
                                       window_dev=self.win_dev, out=self._chunk_out)
        self.loss_buf += l_c
        f.backward(self.gfeat_c)
        if self.overlap is not None:
            self.overlap.join()

    def _chunk_body_lanes(self, pos_from, pos_to, negc):
        """`_chunk_body` with the chunk's rows as `chunk_lanes` concurrent parts (the structure of `_core_passes`): backbone forward of every part up to the pooled
        features on its own stream, join, the fully connected layer over the chunk's rows + the windowed loss launch + the layer's backward on the main stream,
        backbone backward of every part, join.  BatchNorm batch = a part; convolution / BatchNorm parameter gradients add up in the arena (atomics), the fully connected
        layer's go through AccumulateGrad on ONE stream."""
        cur = torch.cuda.current_stream()
        images = _gather_images(self.pool, self.idx_chunk)
        L = self.chunk_lanes
        h = self.cnn_chunk // L
        parts, order = [], {}
        lanes = self.pass_streams[:L]
        for p, st in enumerate(lanes):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                f = self.img_feat_net.forward_pooled(images[p * h:(p + 1) * h], pass_order=(order, p))
                parts.append(f)
                f.record_stream(cur)
        for st in lanes:
            images.record_stream(st)
        for st in lanes:
            cur.wait_stream(st)
        pooled = torch.cat([f.detach() for f in parts]).requires_grad_(True)
        feats = self.img_feat_net.head(pooled)
        self.feats_c.copy_(feats.detach())
        self.gfeat_c.zero_()
        l_c, _, _ = ops.joint_loss_raw(self.table, self.feats_c, pos_from, pos_to, negc, None, self.K_cone, self.alpha, _lib.ENERGY_HYP_CONE,
                                       _lib.LABEL_HYP, _lib.IMAGE_SOFTCLIP, self.table_grad, self.gfeat_c, table_f16=self.table_h,
                                       window_dev=self.win_dev, out=self._chunk_out)
        self.loss_buf += l_c
        feats.backward(self.gfeat_c)
        gp = pooled.grad
        for p, st in enumerate(lanes):
            st.wait_stream(cur)
            gp.record_stream(st)
            with torch.cuda.stream(st):
                parts[p].backward(gp[p * h:(p + 1) * h])
                if self.overlap is not None:
                    self.overlap.join()
        for st in lanes:
            cur.wait_stream(st)

    def _capture_chunk(self):
        """Capture `_chunk_body` into a hipGraph (after eager steps have sized every workspace).  On any capture error the engine keeps launching eagerly and says so."""
        saved_timers = self.backbone.step_timers
        try:
            torch.cuda.synchronize()
            self.backbone.step_timers = None
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                self._chunk_body()
            torch.cuda.synchronize()
            self.chunk_graph = g
        except Exception as e:                                     # noqa: BLE001  (launch mode only; the kernels are the same)
            self.graph_error = '%s: %s' % (type(e).__name__, e)
            self.chunk_graph = None
            import sys
            print('[StepEngine] per-chunk hipGraph capture failed, staying in eager launch mode: %s' % self.graph_error, file=sys.stderr)
            torch.cuda.synchronize()
        finally:
            self.backbone.step_timers = saved_timers

    def _core_chunked(self, ev=None):
        """The same step with the CNN rows in chunks (see `cnn_chunk`), ONE forward per chunk: forward of a chunk, the fused loss over the pairs whose image lies
        in that chunk (lec_joint_loss_fwd_bwd_window: every pair of the step has at most one image end point -- `step()` checks the negative layout -- so a
        chunk's loss terms need that chunk's embeddings and the label table only), backward of the chunk with the gradient the launch left, next chunk.
        Activations of one chunk are alive at a time; BatchNorm batch statistics are per chunk, as the reference's own separate forwards of positives and
        negatives have them (oe_h.py:980-985, 1003-1009).  (Round 4 ran every chunk's forward TWICE -- once without saved activations to have all embeddings
        before one loss launch, once more in front of its backward: a quarter of the step.)  Launch mode: the chunk is ONE hipGraph, captured once and replayed
        for every chunk of every step (the host rewrites the chunk's image indices and four integers of window): ~2 000 launches per chunk -> 3."""
        self.arena.zero_grad(); self.table_grad.zero_()
        if ev: ev[0].record()
        R, C = self.n_rows_pad, self.cnn_chunk
        if getattr(self, 'feats_c', None) is None:
            dev = self.device
            self.feats_c = torch.zeros(C, self.D, device=dev); self.gfeat_c = torch.zeros(C, self.D, device=dev)
            self.idx_chunk = torch.zeros(C, dtype=torch.int64, device=dev)
            self.win_dev = torch.zeros(4, dtype=torch.int32, device=dev)
            self.win_table = torch.tensor([[lo, lo + C, 1 if lo == 0 else 0, lo] for lo in range(0, R, C)], dtype=torch.int32, device=dev)
            self._chunk_feats = torch.zeros(R, self.D, device=dev)                   # every chunk's raw CNN outputs of the step (tests, smoke)
            self._chunk_out = (torch.zeros(self.B, device=dev), torch.zeros(self.B, 2 * self.K, device=dev))
            self.loss_buf = torch.zeros(1, device=dev)
        self.loss_buf.zero_()
        # Several backward passes add into the same gradient slots: the reducer's per-parameter hooks stay muted for all of them (a
        # parameter reports in EVERY chunk; a bucket launched after chunk 0 would reduce a partial sum and race with the later chunks'
        # atomics) and the buckets are reduced once, by step()'s reducer.finish(), after the last chunk's weight gradients have joined.
        live = self.reducer.live
        self.reducer.live = False
        try:
            if (self.use_chunk_graph and self.chunk_graph is None and self.graph_error is None and self.step_no >= self.graph_after):
                self._capture_chunk()                  # (hooks muted: a collective must not be issued into the capture)
            for i, lo in enumerate(range(0, R, C)):
                self.idx_chunk.copy_(self.idx_dev[lo:lo + C]); self.win_dev.copy_(self.win_table[i])
                if self.chunk_graph is not None:
                    self.chunk_graph.replay()
                else:
                    self._chunk_body()
                self._chunk_feats[lo:lo + C].copy_(self.feats_c)
        finally:
            self.reducer.live = live
            self.reducer.reset()                       # nothing launched: finish() reduces every bucket and the table gradient
        self.last_feats = self._chunk_feats
        if ev:
            ev[1].record(); ev[2].record(); ev[3].record()        # (forward, loss and backward interleave per chunk: the whole step is "cnn_fwd" in phases_ms)
        return self.loss_buf.clone(), self._chunk_out[0], self._chunk_out[1]

    def _core(self, ev=None):
        """Forward + fused loss + backward of one step on the static device inputs (codes_dev, idx_dev).  This is the
        region the hipGraph captures."""
        if self.cnn_chunk is not None:
            return self._core_chunked(ev)
        if self.passes > 1:
            return self._core_passes(ev)
        codes = self.codes_dev
        pos_from = codes[:, 0].contiguous(); pos_to = codes[:, 1].contiguous(); negc = codes[:, 2:].contiguous()
        images = _gather_images(self.pool, self.idx_dev)
        self.arena.zero_grad(); self.table_grad.zero_(); self.gfeat.zero_()
        if ev: ev[0].record()
        feats = self.img_feat_net.forward_raw(images)
        self.last_feats = feats.detach()              # raw CNN outputs the loss kernel consumes (a static buffer under graph replay)
        if ev: ev[1].record()
        loss, e_pos, e_neg = ops.joint_loss_raw(self.table, feats.detach(), pos_from, pos_to, negc, None, self.K_cone,
                                                self.alpha, _lib.ENERGY_HYP_CONE, _lib.LABEL_HYP, _lib.IMAGE_SOFTCLIP,
                                                self.table_grad, self.gfeat, table_f16=self.table_h)
        if ev: ev[2].record()
        feats.backward(self.gfeat)
        if self.overlap is not None:
            self.overlap.join()                       # weight gradients from the side stream
        if self.graph_reduces and torch.cuda.is_current_stream_capturing():
            self.reducer.finish()                     # in-graph: the buckets launched by the hooks during backward join here
        if ev: ev[3].record()
        return loss, e_pos, e_neg

    def _core_passes(self, ev=None):
        """`_core` with the CNN rows as `passes` concurrent parts, one HIP stream each (see __init__): backbone forward of every part up
        to the pooled features, join, the fully connected layer over all rows + ONE fused loss launch + the layer's backward on the main
        stream, backbone backward of every part, join.  BatchNorm statistics are per part; the running statistics are updated in part
        order (ResNet.forward's pass_order); the convolutions' and BatchNorms' parameter gradients of the parts add up in the arena (atomics); the
        fully connected layer -- the only parameters whose gradients go through autograd's AccumulateGrad -- stays on ONE stream."""
        codes = self.codes_dev
        pos_from = codes[:, 0].contiguous(); pos_to = codes[:, 1].contiguous(); negc = codes[:, 2:].contiguous()
        images = _gather_images(self.pool, self.idx_dev)
        self.arena.zero_grad(); self.table_grad.zero_(); self.gfeat.zero_()
        if ev: ev[0].record()
        cur = torch.cuda.current_stream()
        h = self.n_rows // self.passes
        parts, order = [], {}
        live = self.reducer.live
        self.reducer.live = False                      # every parameter reports once per part: the buckets are reduced after the last part
        # concurrent parts fill each other's last rounds of workgroup slots: the balanced (stream-K) convolution kernel buys nothing here (131.0 ms
        # with the tile walk, 131.5 with it) and moves 32 GB more per step (slabs, operand re-reads: profiles/r03_conv_f32_balanced.md): tile walk
        schedule = self.backbone.conv_schedule
        self.backbone.conv_schedule = int(os.environ.get('LEC_PASS_SCHEDULE', _lib.SCHEDULE_TILE_WALK))    # (A/B runs: 1 = balanced where it pays, 2 = wherever it applies)
        try:
            for p, st in enumerate(self.pass_streams):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    f = self.img_feat_net.forward_pooled(images[p * h:(p + 1) * h], pass_order=(order, p))
                    parts.append(f)
                    f.record_stream(cur)
            for st in self.pass_streams:
                images.record_stream(st)
            for st in self.pass_streams:
                cur.wait_stream(st)
            pooled = torch.cat([f.detach() for f in parts]).requires_grad_(True)
            feats = self.img_feat_net.head(pooled)
            self.last_feats = feats.detach()
            if ev: ev[1].record()
            loss, e_pos, e_neg = ops.joint_loss_raw(self.table, feats.detach(), pos_from, pos_to, negc, None, self.K_cone,
                                                    self.alpha, _lib.ENERGY_HYP_CONE, _lib.LABEL_HYP, _lib.IMAGE_SOFTCLIP,
                                                    self.table_grad, self.gfeat, table_f16=self.table_h)
            if ev: ev[2].record()
            feats.backward(self.gfeat)                 # fc: d weight, d bias (AccumulateGrad on this stream), d pooled
            gp = pooled.grad
            for p, st in enumerate(self.pass_streams):
                st.wait_stream(cur)
                gp.record_stream(st)
                with torch.cuda.stream(st):
                    parts[p].backward(gp[p * h:(p + 1) * h])
                    if self.overlap is not None:
                        self.overlap.join()
            for st in self.pass_streams:
                cur.wait_stream(st)
        finally:
            self.backbone.conv_schedule = schedule
            self.reducer.live = live
            self.reducer.reset()
        if self.graph_reduces and torch.cuda.is_current_stream_capturing():
            self.reducer.reduce_now()
        if ev: ev[3].record()
        return loss, e_pos, e_neg

    def _capture(self):
        """Capture `_core` into a hipGraph (after eager steps have sized every workspace and MIOpen has settled on its
        solvers).  Under capture the gradient reducer's per-parameter hooks are muted: the all-reduce runs after the
        replay.  On any capture error the engine stays in eager launch mode and says so."""
        saved_timers = self.backbone.step_timers
        try:
            torch.cuda.synchronize()
            self.reducer.live = self.graph_reduces    # hooks muted unless the collectives are captured with the step
            self.reducer.reset()
            self.backbone.step_timers = None           # a captured launch cannot carry timing events
            g = torch.cuda.CUDAGraph()
            # thread_local: RCCL's watchdog thread polls events while we capture; only THIS thread's calls may fail the capture
            # (capturing the main chain on a HIGH-priority stream, so that its nodes outrank the side stream's weight gradients, was
            # measured on the fp32 step: 171 ms against 157 ms -- dropped)
            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                out = self._core(None)
            torch.cuda.synchronize()
            self.hip_graph, self.graph_out = g, out
            self.reducer.live = False                 # after the capture every gradient of a step comes out of a replay
        except Exception as e:                                     # noqa: BLE001  (launch mode only; the kernels are the same)
            self.graph_error = '%s: %s' % (type(e).__name__, e)
            self.hip_graph = None
            self.reducer.live = True
            import sys
            print('[StepEngine] hipGraph capture failed, staying in eager launch mode: %s' % self.graph_error, file=sys.stderr)
            torch.cuda.synchronize()
        finally:
            self.backbone.step_timers = saved_timers

    def set_launch_mode(self, graph):
        """Switch between replaying the captured graph and eager launches (bench.py times per-kernel phases with HIP
        events on eager steps: a replay cannot carry timing events)."""
        if graph and self._graph_saved is not None:
            self.hip_graph, self._graph_saved = self._graph_saved, None
            self.reducer.live = False
        elif not graph and self.hip_graph is not None:
            self._graph_saved, self.hip_graph = self.hip_graph, None
            self.reducer.live = True

    def timer_summary(self):
        """Mean milliseconds per phase over the recorded steps (call after a synchronize)."""
        names = ['cnn_fwd', 'cone_loss', 'cnn_bwd', 'allreduce_wait', 'optimizer']
        gnames = ['graph_fwd_loss_bwd', 'allreduce', 'optimizer']
        recs = [r for r in self.timers['records'] if len(r) == 6]
        grecs = [r for r in self.timers['records'] if len(r) == 4]
        res = {}
        for rs, ns in ((recs, names), (grecs, gnames)):
            if rs:
                for i, n in enumerate(ns):
                    res[n] = sum(ev[i].elapsed_time(ev[i + 1]) for ev in rs) / len(rs)
        def busy(timer):
            """Milliseconds per step during which AT LEAST ONE launch of the family was running: the union of the launches' intervals over
            all streams (event timestamps share one clock).  With concurrent streams the plain sum counts shared time twice."""
            ref = timer[0][0]
            iv = sorted((ref.elapsed_time(a), ref.elapsed_time(b)) for a, b, _ in timer)
            tot, lo, hi = 0.0, iv[0][0], iv[0][1]
            for s_, e_ in iv[1:]:
                if s_ > hi:
                    tot += hi - lo; lo, hi = s_, e_
                else:
                    hi = max(hi, e_)
            return (tot + hi - lo) / max(len(recs), 1)
        tm = self.backbone.step_timers or {'bn': [], 'conv': []}
        bn_t, conv_t = tm['bn'], tm['conv']
        if bn_t:
            res['fused_bn_busy'] = busy(bn_t)
            res['fused_bn'] = sum(a.elapsed_time(b) for a, b, _ in bn_t) / max(len(recs), 1)
            self.bn_bytes_per_step = sum(n for _, _, n in bn_t) / max(len(recs), 1)
            self.bn_launch_groups_per_step = len(bn_t) / max(len(recs), 1)
        if conv_t:
            res['conv_f32_busy'] = busy(conv_t)
            res['conv_f32'] = sum(a.elapsed_time(b) for a, b, _ in conv_t) / max(len(recs), 1)
            self.conv_flops_per_step = sum(n for _, _, n in conv_t) / max(len(recs), 1)
            self.conv_launches_per_step = len(conv_t) / max(len(recs), 1)
        return res

    def close(self):
        self.prefetch.close()
        _release_graphs(self)


def _release_graphs(eng):
    """Destroy an engine's captured graphs NOW (hipGraphExecDestroy, the graph's private memory pool), not whenever Python's cycle
    collector gets to the engine: the next engine of the process (bench.py builds three in a row, the tests dozens) starts from a
    clean runtime state and from freed HBM."""
    import gc
    if getattr(eng, 'hip_graph', None) is not None or getattr(eng, '_graph_saved', None) is not None:
        torch.cuda.synchronize()
    eng.hip_graph = None
    if getattr(eng, 'chunk_graph', None) is not None:
        torch.cuda.synchronize()
        eng.chunk_graph = None
    if hasattr(eng, '_graph_saved'):
        eng._graph_saved = None
    eng.graph_out = None
    gc.collect()


class ClassifierEngine:
    """Config 4 (ethec_experiments.py:243-383 + finetuner.py:199-246) on synthetic inputs: experiment.ETHECExperiment's step --
    ResNet forward, MultiLevelCELoss (one fused launch), backward, gradient all-reduce, flat Adam -- on an image pool resident in
    HBM, forward + loss + backward replayed as ONE hipGraph after a few eager steps.  Image j carries the label chain of leaf
    j mod n_leaf."""

    def __init__(self, workload='cfg4', dtype='fp32', batch=None, lr=1e-4, use_graph=True, graph_after=3, overlap_wgrad=True):
        import tempfile
        from .experiment import ETHECExperiment
        from .loss import MultiLevelCELoss
        hier, arch, B, hw = CLASSIFIER_WORKLOADS[workload]
        self.workload, self.arch, self.B, self.hw = workload, arch, batch or B, hw
        self.rank, self.local_rank, self.world = parallel.init_process_group()
        self.device = torch.device('cuda', self.local_rank % max(torch.cuda.device_count(), 1))
        torch.cuda.set_device(self.device)
        self.labelmap = lm = make_labelmap(hier)
        self.compute_dtype = {'bf16': torch.bfloat16, 'fp32': torch.float32}[dtype]
        torch.manual_seed(0)
        self.exp = ETHECExperiment({}, lm, MultiLevelCELoss(lm), lr=lr, batch_size=self.B, model_name=arch, experiment_dir=tempfile.mkdtemp(prefix='lec_cfg4_'),
                                   compute_dtype=self.compute_dtype)
        if not overlap_wgrad and self.exp.overlap is not None:
            self.exp.overlap.side = None
        if self.exp.overlap is not None and self.exp.overlap.side is not None and self.compute_dtype == torch.float32:
            self.exp.overlap.antiphase = os.environ.get('LEC_WGRAD_ANTIPHASE', '1') != '0'      # (0: A/B runs)
        self.exp.model.train()
        L = len(lm.levels)
        par = lm.parents()
        nleaf = lm.levels[-1]
        chains = np.zeros((nleaf, L), dtype=np.int64)
        for i in range(nleaf):
            v = lm.level_start[-1] + i; c = [v]
            while c[-1] in par:
                c.append(par[c[-1]][0])
            c = c[::-1]
            chains[i] = [c[l] - lm.level_start[l] for l in range(L)]
        self.P = P = self.B
        g = torch.Generator(device='cpu').manual_seed(4321 + self.rank)
        self.pool = _resident_pool(torch.rand(P, 3, hw, hw, generator=g), self.device, self.compute_dtype, getattr(self.exp.model, 'module', self.exp.model), self.B)
        self.pool_levels = torch.from_numpy(chains[np.arange(P) % nleaf]).to(self.device)
        self.idx_dev = torch.zeros(self.B, dtype=torch.int64, device=self.device)
        self.pin = [torch.empty(self.B, dtype=torch.int64).pin_memory() for _ in range(2)]
        self.pin_ev = [None, None]
        self.use_graph, self.graph_after = bool(use_graph), graph_after
        self.hip_graph = self.graph_out = self.graph_error = None
        self.step_no = 0
        self.loss_acc = torch.zeros((), device=self.device)

    def _core(self):
        images = _gather_images(self.pool, self.idx_dev)
        lvl = self.pool_levels.index_select(0, self.idx_dev)
        return self.exp.fwd_bwd(images, lvl)

    def step(self):
        slot = self.step_no & 1
        if self.step_no == 0:
            resnet_mod.library_launches(reset=True)             # (step 0 is launched eagerly: its count is the step's)
        if self.pin_ev[slot] is not None:
            self.pin_ev[slot].synchronize()
        rs = np.random.RandomState(self.step_no * 7919 + self.rank)
        self.pin[slot].numpy()[:] = rs.permutation(self.P)[:self.B] if self.P > self.B else rs.permutation(self.P)
        self.idx_dev.copy_(self.pin[slot], non_blocking=True)
        self.pin_ev[slot] = torch.cuda.Event(); self.pin_ev[slot].record()
        if (self.use_graph and self.hip_graph is None and getattr(self, '_graph_saved', None) is None and self.graph_error is None
                and self.step_no >= self.graph_after):
            try:
                torch.cuda.synchronize()
                self.exp.reducer.live = False
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode='thread_local'):
                    out = self._core()
                torch.cuda.synchronize()
                self.hip_graph, self.graph_out = g, out
            except Exception as e:                                  # noqa: BLE001 (launch mode only)
                self.graph_error = '%s: %s' % (type(e).__name__, e)
                self.exp.reducer.live = True
                import sys
                print('[ClassifierEngine] hipGraph capture failed, staying in eager launch mode: %s' % self.graph_error, file=sys.stderr)
                torch.cuda.synchronize()
        if self.hip_graph is not None:
            self.hip_graph.replay()
            loss, outputs = self.graph_out
            self.exp.reducer.reduce_now()
        else:
            loss, outputs = self._core()
            self.exp.reducer.finish()
        self.exp.arena.adam_step(self.exp.lr, grad_scale=1.0 / self.world)
        self.loss_acc += loss
        if self.step_no == 0:
            self.library_conv_launches_per_step = dict(resnet_mod.library_launches())
        self.step_no += 1
        self.last = (loss, outputs)
        return loss

    def set_launch_mode(self, graph):
        """Replay the captured graph (True) or launch eagerly (False); same contract as StepEngine.set_launch_mode."""
        saved = getattr(self, '_graph_saved', None)
        if graph and saved is not None:
            self.hip_graph, self._graph_saved = saved, None
            self.exp.reducer.live = False
        elif not graph and self.hip_graph is not None:
            self._graph_saved, self.hip_graph = self.hip_graph, None
            self.exp.reducer.live = True

    def close(self):
        _release_graphs(self)
