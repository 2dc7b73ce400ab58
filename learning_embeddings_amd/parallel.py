"""Data parallelism, MI355X-first: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm).

Replaces the reference's single-process `nn.DataParallel` (oe_h.py:301,1434,1439; ethec_experiments.py:240), which
re-broadcasts all parameters every forward, gathers outputs on device 0 and reduce-adds replica gradients there.  Here
every rank owns a full replica whose parameters AND gradients live in one flat fp32 arena each, so that

  * the gradient exchange is a handful of large bucketed SUM all-reduces over contiguous slices of one buffer (the node
    is a full xGMI mesh: large messages let RCCL drive all 7 links), launched from autograd hooks while backward is
    still running, and
  * the optimizer is ONE Adam launch over the arena (ops.adam_flat) instead of ~160 per-tensor launches.

Reduction is SUM, not mean: the reference loss is a plain sum over pairs (oe_h.py:843-846), so the gradient of the
global batch is the sum of the shard gradients.  Gloo (CPU) works for the same code path and is what the CPU tests use.
"""
import os
import threading
import queue
import torch
import torch.distributed as dist

_ALIGN = 64            # elements: every parameter starts on a 256-byte boundary inside the arena


class FlatArena:
    """Re-homes `params` into one flat fp32 buffer (`data`) with a twin gradient buffer (`grad`)."""

    def __init__(self, params, device=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('FlatArena: no trainable parameters')
        device = device or self.params[0].device
        self.offsets = []
        off = 0
        for p in self.params:
            if p.dtype != torch.float32:
                raise TypeError('FlatArena holds fp32 master parameters only')
            self.offsets.append(off)
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = off
        self.data = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device)
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            view = self.data[o:o + n].view(p.shape)
            view.copy_(p.data)
            if p.data.dim() == 4 and p.data.is_contiguous(memory_format=torch.channels_last) and not p.data.is_contiguous():
                # keep NHWC-strided conv weights: store the permuted bytes, expose the same logical strides
                nhwc = self.data[o:o + n].view(p.shape[0], p.shape[2], p.shape[3], p.shape[1]).permute(0, 3, 1, 2)
                nhwc.copy_(p.data); view = nhwc
                gview = self.grad[o:o + n].view(p.shape[0], p.shape[2], p.shape[3], p.shape[1]).permute(0, 3, 1, 2)
            else:
                gview = self.grad[o:o + n].view(p.shape)
            p.data = view
            p.grad = gview
        self.exp_avg = None; self.exp_avg_sq = None; self.step = 0
        self.data_lp = None

    def enable_lowp_shadow(self):
        """Keep a flat bf16 copy of the parameters, refreshed by the Adam kernel itself; `lowp_view(p)` is parameter p's
        slot in it (same shape and memory layout as p)."""
        if self.data_lp is None:
            self.data_lp = self.data.to(torch.bfloat16)
            self._lp_views = {id(p): self.view_of(self.data_lp, k) for k, p in enumerate(self.params)}
        return self

    def lowp_view(self, p):
        return self._lp_views.get(id(p)) if self.data_lp is not None else None

    def enable_lowp_transposed(self):
        """Keep, beside the bf16 shadow, a TRANSPOSED twin of every convolution weight -- memory [Cin][RS][Cout], the k-contiguous operand of the
        bf16 data-gradient kernels (lec_conv_bf16_dgrad) -- refreshed by ONE launch right after the Adam kernel (lec_conv_bf16_wt_transpose_flat).
        `lowp_t_view(p)` is parameter p's slot: a channels_last [Cin, Cout, R, S] tensor."""
        self.enable_lowp_shadow()
        if getattr(self, 'data_lp_t', None) is not None:
            return self
        rows = []; views = {}; tile = 0
        self.data_lp_t = torch.zeros_like(self.data_lp)
        for k, p in enumerate(self.params):
            if p.dim() != 4:
                continue
            cout, cin, r, s_ = p.shape
            nhwc = p.data.is_contiguous(memory_format=torch.channels_last) or (r == 1 and s_ == 1)
            if not nhwc or cin % 8 or cout % 64:
                continue
            o = self.offsets[k]
            rows.append([o, cout, r * s_, cin, tile])
            tile += r * s_ * ((cout + 63) // 64) * ((cin + 63) // 64)
            views[id(p)] = self.data_lp_t[o:o + p.numel()].view(cin, r, s_, cout).permute(0, 3, 1, 2)
        self._lpt_views = views
        self._lpt_table = torch.tensor(rows, dtype=torch.int32, device=self.data.device).contiguous() if rows else None
        self._lpt_tiles = tile
        self.refresh_lowp_t()
        return self

    def lowp_t_view(self, p):
        return self._lpt_views.get(id(p)) if getattr(self, 'data_lp_t', None) is not None else None

    def refresh_lowp_t(self):
        if getattr(self, 'data_lp_t', None) is not None and self._lpt_table is not None:
            from . import _lib
            _lib.check(_lib.lib.lec_conv_bf16_wt_transpose_flat(_lib.dptr(self.data_lp), _lib.dptr(self.data_lp_t), _lib.dptr(self._lpt_table),
                                                                int(self._lpt_table.shape[0]), int(self._lpt_tiles), _lib.stream_ptr()))

    def refresh_lowp(self):
        if self.data_lp is not None:
            self.data_lp.copy_(self.data)
            self.refresh_lowp_t()

    def zero_grad(self):
        self.grad.zero_()

    def slice_of(self, p):
        i = next(k for k, q in enumerate(self.params) if q is p)
        return self.offsets[i], self.params[i].numel()

    def view_of(self, buf, k):
        """View of flat buffer `buf` over parameter k's slot with the parameter's own shape AND memory layout (conv weights
        kept channels_last are stored as their NHWC bytes)."""
        p, o = self.params[k], self.offsets[k]
        flat = buf[o:o + p.numel()]
        if p.dim() == 4 and not p.data.is_contiguous() and p.data.is_contiguous(memory_format=torch.channels_last):
            return flat.view(p.shape[0], p.shape[2], p.shape[3], p.shape[1]).permute(0, 3, 1, 2)
        return flat.view(p.shape)

    # ---- interchange with torch.optim.Adam checkpoints (the reference saves optimizer_labels.state_dict(), oe_h.py:1880)
    def export_adam_state(self, first_index=0):
        """{param index: {'step', 'exp_avg', 'exp_avg_sq'}} in this arena's parameter order, torch.optim.Adam layout."""
        out = {}
        for k, p in enumerate(self.params):
            if self.exp_avg is not None:
                m = self.view_of(self.exp_avg, k).clone(); v = self.view_of(self.exp_avg_sq, k).clone()
            else:
                m = torch.zeros_like(p.data); v = torch.zeros_like(p.data)
            out[first_index + k] = {'step': torch.tensor(float(self.step)), 'exp_avg': m, 'exp_avg_sq': v}
        return out

    def import_adam_state(self, state, first_index=0):
        if self.exp_avg is None:
            self.exp_avg = torch.zeros_like(self.data); self.exp_avg_sq = torch.zeros_like(self.data)
        steps = set()
        for k, p in enumerate(self.params):
            st = state.get(first_index + k)
            if st is None:
                continue
            self.view_of(self.exp_avg, k).copy_(st['exp_avg']); self.view_of(self.exp_avg_sq, k).copy_(st['exp_avg_sq'])
            steps.add(int(float(st['step'])))
        if len(steps) > 1:
            raise ValueError('the flat Adam keeps ONE step counter; the checkpoint has %s' % sorted(steps))
        if steps:
            self.step = steps.pop()

    def adam_step(self, lr, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
        from . import ops
        if self.exp_avg is None:
            self.exp_avg = torch.zeros_like(self.data); self.exp_avg_sq = torch.zeros_like(self.data)
        self.step += 1
        ops.adam_flat(self.data, self.grad, self.exp_avg, self.exp_avg_sq, self.step, lr, betas, eps, grad_scale, self.data_lp)
        self.refresh_lowp_t()                                   # the data gradients' transposed bf16 weights follow (one launch; no-op without them)


# ------------------------------------------------------------------------------------------------ process group
class PinnedRing:
    """Small host -> device uploads of a step (index arrays: node codes, image slots, flip bits) through PINNED memory: a ring of `slots`
    page-locked buffers, one per step in flight, each guarded by an event recorded after the step's last copy was enqueued.  A copy from
    pageable memory (`torch.from_numpy(a).to(device)`) goes through the runtime's own staging and is not asynchronous to the host; from a
    pinned buffer it is one DMA the stream orders, and the host moves on."""

    def __init__(self, slots=4, nbytes=1 << 16):
        self.bufs = [None] * slots; self.events = [None] * slots
        self.nbytes = nbytes; self.cur = 0; self.off = 0

    def begin_step(self):
        self.cur = (self.cur + 1) % len(self.bufs); self.off = 0
        ev = self.events[self.cur]
        if ev is not None:
            ev.synchronize()                                   # copies of `slots` steps ago: long done, this only makes reuse safe by construction

    def upload(self, arr, device):
        """numpy array -> device tensor of the same dtype / shape (asynchronous copy on the current stream)."""
        import numpy as np, torch
        arr = np.ascontiguousarray(arr)
        n = arr.nbytes
        start = (self.off + 15) & ~15
        buf = self.bufs[self.cur]
        if buf is None or start + n > buf.numel():
            if buf is not None and self.off > 0:               # the slot's buffer is in use by earlier uploads of this step: leave it, take a new one
                self._keep = getattr(self, '_keep', []) + [buf]
            buf = self.bufs[self.cur] = torch.empty(max(self.nbytes, 2 * (start + n)), dtype=torch.uint8).pin_memory()
            start = 0
        view = buf[start:start + n].view(torch.from_numpy(arr[:0]).dtype if n else torch.uint8)
        if n:
            view.numpy()[...] = arr.reshape(-1)
        self.off = start + n
        return view.to(device, non_blocking=True).view(arr.shape)

    def end_step(self):
        import torch
        ev = self.events[self.cur]
        if ev is None:
            ev = self.events[self.cur] = torch.cuda.Event()
        ev.record()
        self._keep = []


def env_rank():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))


def init_process_group(backend=None):
    """Idempotent.  Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torchrun contract)."""
    rank, local_rank, world = env_rank()
    # LEC_FORCE_DIST=1: bring the process group (and the gradient reducer) up even for ONE rank -- exercises the RCCL
    # all-reduce path, its streams and its interplay with the hipGraph replay on a single-GPU box
    if (world > 1 or os.environ.get('LEC_FORCE_DIST')) and not dist.is_initialized():
        if backend is None:
            # LEC_DIST_BACKEND=gloo lets several ranks share ONE GPU (tests on a 1-GPU box): gloo reduces CUDA tensors
            backend = os.environ.get('LEC_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
    return rank, local_rank, world


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


class DpComm:
    """The C ABI's RCCL layer (include/lecone.h 5b: lec_dp_unique_id / lec_dp_init / lec_dp_allreduce_sum / lec_dp_destroy) as the
    gradient exchange: what a maintainer of the reference binds in place of nn.DataParallel.  Rank 0 draws the 128-byte unique id;
    it travels to the other ranks through the already-initialised torch.distributed group (any side channel would do).
    `LEC_DP_BACKEND=lecone` makes GradientReducer use it instead of torch.distributed's own all-reduce."""

    def __init__(self, device=None):
        import ctypes as C
        from ._lib import lib, check
        self._lib, self._check, self._C = lib, check, C
        self.rank, self.world = rank(), world_size()
        dev = torch.cuda.current_device() if device is None else (device.index if isinstance(device, torch.device) else int(device))
        buf = (C.c_char * 128)()
        if self.rank == 0:
            check(lib.lec_dp_unique_id(C.cast(buf, C.c_void_p)))
        if self.world > 1:
            box = [bytes(buf)]
            dist.broadcast_object_list(box, src=0)
            buf = (C.c_char * 128).from_buffer_copy(box[0])
        h = C.c_void_p()
        check(lib.lec_dp_init(C.byref(h), self.rank, self.world, C.cast(buf, C.c_void_p), dev))
        self._h = h

    def allreduce_sum_(self, t, stream=None):
        """In-place SUM all-reduce of a contiguous fp32 / bf16 device tensor, asynchronous on `stream` (default: the current one)."""
        if not (t.is_cuda and t.is_contiguous()) or t.dtype not in (torch.float32, torch.bfloat16):
            raise ValueError('DpComm.allreduce_sum_: contiguous fp32 / bf16 device tensor required')
        st = (stream or torch.cuda.current_stream()).cuda_stream
        self._check(self._lib.lec_dp_allreduce_sum(self._h, self._C.c_void_p(t.data_ptr()), t.numel(), 0 if t.dtype == torch.float32 else 1,
                                                   self._C.c_void_p(st)))
        return t

    def close(self):
        if getattr(self, '_h', None):
            self._lib.lec_dp_destroy(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _StreamHandle:
    """What dist.all_reduce(async_op=True) returns, for a collective enqueued on a HIP stream through the C ABI."""

    def __init__(self, stream):
        self.ev = torch.cuda.Event(); self.ev.record(stream)

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


class GradientReducer:
    """Bucketed, backward-overlapped SUM all-reduce of a FlatArena's gradient (plus any extra flat tensors, e.g. the
    label table's dense gradient).  Buckets are contiguous slices of the arena in REVERSE parameter order, so the first
    bucket to fill is the one backward finishes first."""

    def __init__(self, arena, bucket_mb=32.0, extra=()):
        self.arena = arena
        self.extra = list(extra)
        self.enabled = world_size() > 1 or (dist.is_initialized() and bool(os.environ.get('LEC_FORCE_DIST')))
        # LEC_DP_BACKEND=lecone: the collective goes through liblecone's own RCCL layer (lec_dp_allreduce_sum) instead of
        # torch.distributed's -- same RCCL underneath, reached through the C ABI a reference maintainer would bind
        self.comm = DpComm() if (self.enabled and os.environ.get('LEC_DP_BACKEND') == 'lecone' and arena.grad.is_cuda) else None
        self.live = True             # False: hooks are muted (gradients produced by a hipGraph replay; see reduce_now)
        self.side_streams = []       # streams other than the autograd one that write gradients (WgradOverlap registers its own)
        self._launch = None          # stream the bucket all-reduces are issued from (waits for every producer stream)
        self.handles = []
        self.buckets = []            # (start, end, [param indices])
        cap = int(bucket_mb * 1024 * 1024 / 4)
        idxs = list(range(len(arena.params)))[::-1]
        cur, cur_n = [], 0
        for i in idxs:
            n = arena.params[i].numel()
            cur.append(i); cur_n += n
            if cur_n >= cap:
                self.buckets.append(cur); cur, cur_n = [], 0
        if cur:
            self.buckets.append(cur)
        self._spans = []
        for b in self.buckets:
            lo = min(arena.offsets[i] for i in b)
            hi = max(arena.offsets[i] + (arena.params[i].numel() + _ALIGN - 1) // _ALIGN * _ALIGN for i in b)
            self._spans.append((lo, min(hi, arena.numel)))
        self._bucket_of = {}
        for bi, b in enumerate(self.buckets):
            for i in b:
                self._bucket_of[i] = bi
        self._pending = [0] * len(self.buckets)
        self._index_of = {id(p): i for i, p in enumerate(arena.params)}
        self._hooks = [self._make_hook(i) for i in range(len(arena.params))]
        if self.enabled:
            for i, p in enumerate(arena.params):
                p.register_post_accumulate_grad_hook(self._hooks[i])
        self.reset()

    def reset(self):
        self._pending = [len(b) for b in self.buckets]
        self._ready = set()
        self.handles = []

    def _make_hook(self, i):
        def hook(_p):
            if not self.live:
                return
            if i in self._ready:             # a parameter counts once per step, whoever reports it (autograd's
                return                       # AccumulateGrad hook, a manual mark_ready from a fused backward, or both)
            self._ready.add(i)
            bi = self._bucket_of[i]
            self._pending[bi] -= 1
            if self._pending[bi] == 0:
                lo, hi = self._spans[bi]
                self.handles.append(self._launch_bucket(self.arena.grad[lo:hi]))
        return hook

    def _launch_bucket(self, flat):
        """Issue one bucket's all-reduce.  The bucket's gradients come from more than one stream (BatchNorm / fc gradients
        from the autograd stream, convolution weight gradients from WgradOverlap's side stream) and the parameter that
        completes the bucket may report from either: the collective is issued from a third stream that waits for all of
        them, so neither producer stream stalls and the reduction never reads a gradient still being written."""
        if not (flat.is_cuda and (self.side_streams or self.comm is not None)):
            return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
        if self._launch is None:
            self._launch = torch.cuda.Stream()
        self._launch.wait_stream(torch.cuda.current_stream())
        for s in self.side_streams:
            self._launch.wait_stream(s)
        with torch.cuda.stream(self._launch):
            return self._allreduce(flat)

    def _allreduce(self, t):
        """One SUM all-reduce on the current stream; returns a handle with .wait()."""
        if self.comm is not None:
            self.comm.allreduce_sum_(t)
            return _StreamHandle(torch.cuda.current_stream())
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)

    def mark_ready(self, p):
        """Manual form of the post-accumulate hook, for gradients written outside autograd (e.g. weight gradients computed
        on a side stream): call it on the stream that produced the gradient."""
        if not self.enabled:
            return
        i = self._index_of.get(id(p))
        if i is not None:
            self._hooks[i](p)

    def reduce_now(self):
        """All-reduce every bucket and the extras now (gradients already complete: after a hipGraph replay)."""
        if self.enabled:
            self._reduce_now()

    def finish(self):
        """Reduce whatever has not been launched by hooks (unused parameters, the extras) and wait for everything."""
        if not self.enabled:
            return
        for bi, left in enumerate(self._pending):
            if left > 0:
                lo, hi = self._spans[bi]
                self.handles.append(self._allreduce(self.arena.grad[lo:hi]))
        for t in self.extra:
            self.handles.append(self._allreduce(t))
        for h in self.handles:
            h.wait()
        self.reset()


    def time_buckets(self, reps=5):
        """Each bucket's (and each extra tensor's) SUM all-reduce alone, HIP-event timed on the current stream: [{'mb', 'ms'}].  The
        buffers are saved and restored (the sums overflow harmlessly in between): training state is untouched.  Collective: every rank
        must call it."""
        out = []
        if not self.enabled:
            return out
        spans = [self.arena.grad[lo:hi] for lo, hi in self._spans] + list(self.extra)
        for t in spans:
            keep = t.clone()
            h = self._allreduce(t); h.wait()                                         # untimed first call
            if t.is_cuda:
                a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(reps):
                    self._allreduce(t).wait()
                b.record(); b.synchronize()
                ms = a.elapsed_time(b) / reps
            else:
                import time
                t0 = time.perf_counter()
                for _ in range(reps):
                    self._allreduce(t).wait()
                ms = (time.perf_counter() - t0) / reps * 1e3
            t.copy_(keep)
            out.append({'mb': round(t.numel() * t.element_size() / 1e6, 2), 'ms': round(ms, 4)})
        return out

    def _reduce_now(self):
        for lo, hi in self._spans:
            self.handles.append(self._allreduce(self.arena.grad[lo:hi]))
        for t in self.extra:
            self.handles.append(self._allreduce(t))
        for h in self.handles:
            h.wait()
        self.reset()


def allreduce_sum_(t):
    if world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def shard_range(n_global, r=None, w=None):
    """Contiguous shard of a global batch for this rank."""
    r = rank() if r is None else r; w = world_size() if w is None else w
    per = n_global // w
    if per * w != n_global:
        raise ValueError('global batch %d is not divisible by world size %d' % (n_global, w))
    return r * per, (r + 1) * per


class NegativePrefetcher:
    """Draws the negatives of step s+1 on a host thread while the GPU works on step s.  The ctypes call releases the
    GIL, so the MT19937 walk genuinely overlaps.  Draw order is the reference's (oe_h.py:940-957); in 'replicated' mode
    every rank walks the GLOBAL batch's stream and keeps its shard, which makes the indices bit-identical to the
    single-process run whatever the world size, with no collective."""

    def __init__(self, graph, positives_fn, K, mode='replicated', depth=2, on_item=None, rank=None, world=None):
        """positives_fn(s) -> (from, to) global positives of step s, or None when there is no step s (the thread ends).  on_item(item): called
        on the producer thread with every (from, to, neg) shard before it is queued (the trainer starts image decodes there)."""
        self.graph, self.positives_fn, self.K, self.mode = graph, positives_fn, K, mode
        self.on_item, self.rank, self.world = on_item, rank, world
        self.q = queue.Queue(maxsize=depth)
        self.step = 0
        self._stop = False
        self.th = threading.Thread(target=self._run, daemon=True)
        self.th.start()

    def _put(self, x):
        while not self._stop:
            try:
                self.q.put(x, timeout=0.1); return
            except queue.Full:
                pass

    def _run(self):
        s = 0
        try:
            while not self._stop:
                pos = self.positives_fn(s)                     # global positives of step s (numpy int32)
                if pos is None:
                    self._put((-2, None)); return
                frm, to = pos
                lo, hi = shard_range(len(frm), self.rank, self.world)
                if self.mode == 'replicated':
                    neg = self.graph.draw_batch(frm, to, self.K)
                    item = (frm[lo:hi], to[lo:hi], neg[lo:hi])
                else:                                          # 'per_rank': independent stream per rank
                    item = (frm[lo:hi], to[lo:hi], self.graph.draw_batch(frm[lo:hi], to[lo:hi], self.K))
                if self.on_item is not None:
                    self.on_item(item)
                self._put((s, item))
                s += 1
        except Exception as e:                                 # surface sampler errors in the consumer
            self._put((-1, e))

    def next(self):
        s, item = self.q.get()
        if s == -2:
            raise StopIteration('no more steps to draw negatives for')
        if s < 0:
            raise item
        return item

    def close(self):
        self._stop = True
        try:
            while True:
                self.q.get_nowait()
        except queue.Empty:
            pass
        if self.th is not threading.current_thread():
            self.th.join(timeout=5.0)                           # the sampler is the caller's again only once the producer has left it
