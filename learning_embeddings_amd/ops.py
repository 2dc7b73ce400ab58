"""Tensor-level wrappers + autograd Functions over the C ABI (include/lecone.h).  PyTorch is plumbing here: device
memory, the current HIP stream and autograd bookkeeping.  All arithmetic happens in liblecone.so on the MI355X."""
import ctypes as C
import numpy as np
import torch

from . import _lib
from ._lib import lib, check, dptr, stream_ptr

ENERGY = {'hyp_cone': _lib.ENERGY_HYP_CONE, 'order': _lib.ENERGY_ORDER, 'euc_cone': _lib.ENERGY_EUC_CONE}

_workspaces = {}


def _workspace(device, nbytes):
    """Scratch of the loss kernels (arrival counter + partials), one per device AND stream: launches that share it must be stream-ordered
    (lec_common.h block_publish_and_finalize), and two trainers of one process may run on different streams."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream if device.type == 'cuda' else 0)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.zeros(max(int(nbytes), 1 << 16), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def _rows(t, name):
    if t.dtype != torch.float32:
        raise TypeError('%s must be float32, got %s' % (name, t.dtype))
    if t.dim() != 2:
        raise ValueError('%s must be 2-D [rows, D]' % name)
    if t.stride(1) != 1 and t.shape[1] > 1:
        t = t.contiguous()
    return t


def _ld(t):
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))


# ------------------------------------------------------------------------------------------------ pair energies
class PairEnergyFn(torch.autograd.Function):
    """E_operator (oe_h.py:811-833 / order_embeddings.py:818-824) with its autograd."""

    @staticmethod
    def forward(ctx, x, y, K_cone, energy):
        shp = x.shape
        x2 = _rows(x.reshape(-1, shp[-1]), 'x'); y2 = _rows(y.reshape(-1, shp[-1]), 'y')
        if x2.shape != y2.shape:
            raise ValueError('x and y must have the same shape')
        P, D = x2.shape
        E = torch.empty(P, dtype=torch.float32, device=x.device)
        check(lib.lec_pair_energy_fwd(energy, dptr(x2), _ld(x2), dptr(y2), _ld(y2), P, D, float(K_cone), dptr(E), stream_ptr()))
        ctx.save_for_backward(x2, y2)
        ctx.meta = (float(K_cone), energy, shp)
        return E.view(shp[:-1])

    @staticmethod
    def backward(ctx, gE):
        x2, y2 = ctx.saved_tensors
        K_cone, energy, shp = ctx.meta
        P, D = x2.shape
        g = gE.reshape(-1).contiguous().float()
        gx = torch.empty(P, D, dtype=torch.float32, device=x2.device); gy = torch.empty_like(gx)
        check(lib.lec_pair_energy_bwd(energy, dptr(x2), _ld(x2), dptr(y2), _ld(y2), dptr(g), P, D, K_cone, dptr(gx), dptr(gy), D, stream_ptr()))
        return gx.view(shp), gy.view(shp), None, None


def pair_energy(x, y, K_cone=0.1, energy='hyp_cone'):
    return PairEnergyFn.apply(x, y, 0.0 if K_cone is None else K_cone, ENERGY[energy])


def energy_matrix(apex, points, K_cone=0.1, energy='hyp_cone'):
    """E[i, j] = E(apex_j, points_i): every image against every label (oe_h.py:2018-2036 done in one launch)."""
    a = _rows(apex, 'apex'); p = _rows(points, 'points')
    N, D = a.shape; M = p.shape[0]
    E = torch.empty(M, N, dtype=torch.float32, device=a.device)
    check(lib.lec_pair_energy_matrix(ENERGY[energy], dptr(a), _ld(a), N, dptr(p), _ld(p), M, D, float(K_cone or 0.0), dptr(E), N, stream_ptr()))
    return E


def level_topk(apex, points, level_start, k, K_cone=0.1, energy='hyp_cone'):
    """Per level l and point i, the k apexes of rows [level_start[l], level_start[l+1]) with the smallest E(apex, point_i)
    (oe_h.py:2018-2036: E_operator + torch.topk(largest=False) per level), fused: no [M, N] matrix.  `level_start`: L+1
    ints.  Returns (idx int32 [M, L, k], val float32 [M, L, k]); entries past a short level are (-1, +inf)."""
    a = _rows(apex, 'apex'); p = _rows(points, 'points')
    N, D = a.shape; M = p.shape[0]
    ls = np.ascontiguousarray(np.asarray(level_start, dtype=np.int32))
    L = ls.size - 1
    if L < 1 or int(ls[-1]) > N or int(ls[0]) < 0 or np.any(np.diff(ls) < 0):
        raise ValueError('level_start must be L+1 non-decreasing offsets into the apex rows')
    idx = torch.empty(M, L, k, dtype=torch.int32, device=a.device); val = torch.empty(M, L, k, dtype=torch.float32, device=a.device)
    check(lib.lec_level_topk(ENERGY[energy], dptr(a), _ld(a), N, dptr(p), _ld(p), M, D, ls.ctypes.data_as(C.c_void_p), L, int(k), float(K_cone or 0.0),
                             dptr(idx), dptr(val), stream_ptr()))
    return idx, val


# ------------------------------------------------------------------------------------------------ projections
class LabelProjectFn(torch.autograd.Function):
    """Embedder.forward (oe_h.py:77-104): gather + exp-map style tanh projection + straight-through clip; with
    mode=LABEL_SOFTCLIP_K the Euclidean trainer's form (oe.py:65-80): gather + x/|x| (|x| + K)."""

    @staticmethod
    def forward(ctx, weight, idx, K_cone, mode=_lib.LABEL_HYP):
        w = _rows(weight, 'weight')
        idx = idx.reshape(-1).to(torch.int64).contiguous()
        n, D = idx.numel(), w.shape[1]
        out = torch.empty(n, D, dtype=torch.float32, device=w.device)
        check(lib.lec_label_project_fwd(mode, dptr(w), _ld(w), w.shape[0], dptr(idx), n, D, float(K_cone), dptr(out), D, stream_ptr()))
        ctx.save_for_backward(w, idx); ctx.K = float(K_cone); ctx.mode = mode
        return out

    @staticmethod
    def backward(ctx, gout):
        w, idx = ctx.saved_tensors
        g = gout.contiguous().float()
        gW = torch.zeros(w.shape, dtype=torch.float32, device=w.device)        # dense (sparse=False semantics)
        check(lib.lec_label_project_bwd(ctx.mode, dptr(w), _ld(w), w.shape[0], dptr(idx), idx.numel(), w.shape[1], ctx.K, dptr(g), g.shape[1], dptr(gW), stream_ptr()))
        return gW, None, None, None


class ImageSoftClipFn(torch.autograd.Function):
    """FeatCNN18.soft_clip (oe_h.py:323-328: + r_in(K)); mode=IMAGE_SOFTCLIP_K is oe.py:235-240 (+ K itself)."""

    @staticmethod
    def forward(ctx, raw, K_cone, mode=_lib.IMAGE_SOFTCLIP):
        shp = raw.shape
        r = _rows(raw.reshape(-1, shp[-1]).float(), 'raw')
        n, D = r.shape
        out = torch.empty(n, D, dtype=torch.float32, device=r.device)
        check(lib.lec_image_softclip_fwd(mode, dptr(r), _ld(r), n, D, float(K_cone), dptr(out), D, stream_ptr()))
        ctx.save_for_backward(r); ctx.meta = (float(K_cone), shp, raw.dtype); ctx.mode = mode
        return out.view(shp)

    @staticmethod
    def backward(ctx, gout):
        (r,) = ctx.saved_tensors
        K_cone, shp, dt = ctx.meta
        g = gout.reshape(r.shape).contiguous().float()
        graw = torch.empty_like(r)
        check(lib.lec_image_softclip_bwd(ctx.mode, dptr(r), _ld(r), dptr(g), g.shape[1], r.shape[0], r.shape[1], K_cone, dptr(graw), r.shape[1], stream_ptr()))
        return graw.view(shp).to(dt), None, None


# ------------------------------------------------------------------------------------------------ fused joint loss
def joint_loss_raw(table, feat, pos_from, pos_to, neg, weights, K_cone, alpha, energy, label_proj, image_proj,
                   grad_table=None, grad_feat=None, table_f16=None, window=None, out=None, window_dev=None):
    """One launch of lec_joint_loss_fwd_bwd.  table [N,D], feat [n_feat,D] (or None): contiguous float32; index tensors:
    contiguous int32 device tensors of node codes (>= 0 label row, < 0 feature row -1-code).  Gradients are ADDED into
    grad_table / grad_feat when given (same shapes, contiguous).  Returns (loss[1], e_pos[B], e_neg[B,2K]).
    window = (row_lo, row_hi, labels_too): lec_joint_loss_fwd_bwd_window -- only the pairs whose image row lies in [row_lo, row_hi) (and the label-label pairs
    when labels_too) are evaluated; `out` = (e_pos, e_neg) buffers the launches of one step share (entries of other windows' pairs are left alone).
    window_dev: int32[4] DEVICE tensor {row_lo, row_hi, labels_too, feat_base} read by the kernel itself; feat / grad_feat are then chunk buffers whose row 0 is
    feature row feat_base (a launch that can be captured once and replayed for every chunk)."""
    def chk(t, name, like=None):
        if t.dtype != torch.float32 or t.dim() != 2 or not t.is_contiguous():
            raise ValueError('%s must be a contiguous 2-D float32 tensor' % name)
        if like is not None and t.shape != like.shape:
            raise ValueError('%s must have the shape of its parameter' % name)
    chk(table, 'table')
    N, D = table.shape
    dev = table.device
    n_feat = 0
    if feat is not None and feat.numel():
        chk(feat, 'feat'); n_feat = feat.shape[0]
        if feat.shape[1] != D:
            raise ValueError('feat rows must have the table\'s embedding_dim')
    else:
        feat = None
    if grad_table is not None:
        chk(grad_table, 'grad_table', table)
        if feat is not None:
            if grad_feat is None:
                raise ValueError('grad_feat is required when grad_table is given and image rows exist')
            chk(grad_feat, 'grad_feat', feat)
    B = pos_from.numel(); K = (neg.shape[1] // 2) if (neg is not None and neg.numel()) else 0
    for name, t in (('pos_from', pos_from), ('pos_to', pos_to), ('neg', neg)):
        if t is not None and (t.dtype != torch.int32 or not t.is_contiguous()):
            raise TypeError('%s must be a contiguous int32 tensor of node codes' % name)
    if weights is not None and (weights.dtype != torch.float32 or weights.numel() != B or not weights.is_contiguous()):
        raise ValueError('weights must be float32 [B]')
    if out is not None:
        e_pos, e_neg = out
        if (e_pos.shape != (B,) or e_neg.shape != (B, 2 * K) or e_pos.dtype != torch.float32 or e_neg.dtype != torch.float32
                or not e_pos.is_contiguous() or not e_neg.is_contiguous()):
            raise ValueError('out must be contiguous float32 (e_pos [B], e_neg [B, 2K])')
    else:
        e_pos = torch.empty(B, dtype=torch.float32, device=dev)
        e_neg = torch.empty(B, 2 * K, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    need = lib.lec_loss_workspace_bytes(B, K, D)
    if need < 0:
        check(int(need))
    ws = _workspace(dev, need)
    fn, tbl = lib.lec_joint_loss_fwd_bwd, table
    if table_f16 is not None:
        if table_f16.dtype != torch.float16 or table_f16.shape != table.shape or not table_f16.is_contiguous():
            raise ValueError('table_f16 must be a contiguous float16 tensor of the table\'s shape')
        fn, tbl = lib.lec_joint_loss_fwd_bwd_f16, table_f16
    if window_dev is not None:
        if window_dev.dtype != torch.int32 or window_dev.numel() != 4 or not window_dev.is_cuda:
            raise ValueError('window_dev must be a device int32[4] tensor {row_lo, row_hi, labels_too, feat_base}')
        window = (0, 0, 0)
    if window is not None:
        lo, hi, labels_too = window
        check(lib.lec_joint_loss_fwd_bwd_window(energy, label_proj, image_proj, dptr(table), dptr(table_f16) if table_f16 is not None else None, D, N,
                                                dptr(feat), D, n_feat, dptr(pos_from), dptr(pos_to), dptr(neg) if K else None,
                                                dptr(weights), B, K, D, float(K_cone), float(alpha), int(lo), int(hi), int(bool(labels_too)), dptr(window_dev),
                                                dptr(e_pos), dptr(e_neg) if K else None, dptr(loss),
                                                dptr(grad_table), dptr(grad_feat) if feat is not None else dptr(grad_table),
                                                dptr(ws), ws.numel(), stream_ptr()))
        return loss, e_pos, e_neg
    check(fn(energy, label_proj, image_proj, dptr(tbl), D, N,
                                     dptr(feat), D, n_feat, dptr(pos_from), dptr(pos_to), dptr(neg) if K else None,
                                     dptr(weights), B, K, D, float(K_cone), float(alpha),
                                     dptr(e_pos), dptr(e_neg) if K else None, dptr(loss),
                                     dptr(grad_table), dptr(grad_feat) if feat is not None else dptr(grad_table),
                                     dptr(ws), ws.numel(), stream_ptr()))
    return loss, e_pos, e_neg


class JointLossFn(torch.autograd.Function):
    """criterion.forward's train branch after sampling (oe_h.py:929-967) + backward, one kernel.
    The gradients are produced by the same launch as the forward and handed to autograd in backward()."""

    @staticmethod
    def forward(ctx, table, feat, pos_from, pos_to, neg, weights, K_cone, alpha, energy, label_proj, image_proj):
        need_grad = table.requires_grad or (feat is not None and feat.requires_grad)
        tb = table.detach().float().contiguous()
        ft = feat.detach().float().contiguous() if (feat is not None and feat.numel()) else None
        gt = gf = None
        if need_grad:
            gt = torch.zeros_like(tb)
            gf = torch.zeros_like(ft) if ft is not None else None
        loss, e_pos, e_neg = joint_loss_raw(tb, ft, pos_from, pos_to, neg, weights, K_cone, alpha, energy, label_proj,
                                            image_proj, gt, gf)
        ctx.grads = (gt, gf)
        ctx.dts = (table.dtype, feat.dtype if feat is not None else None)
        ctx.mark_non_differentiable(e_pos, e_neg)
        return loss.view(()), e_pos, e_neg

    @staticmethod
    def backward(ctx, gl, _gp, _gn):
        gt, gf = ctx.grads
        if gt is None:
            return (None,) * 11
        gt = (gt * gl).to(ctx.dts[0])
        if gf is not None:
            gf = (gf * gl).to(ctx.dts[1])
        return gt, gf, None, None, None, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------------ table / optimiser steps
def table_step_adam(table, grad, exp_avg, exp_avg_sq, step, lr, K_cone=0.1, betas=(0.9, 0.999), eps=1e-8,
                    riemannian=True, clip=True, table_f16=None):
    """oe_h.py:1768-1771 in one pass: grad *= (1/lambda_x)^2 -> Adam -> clip into [r_in, 1-1e-5].  In place.
    table_f16: optional fp16 shadow of the table, refreshed with the updated rows in the same pass (config 5)."""
    for t in (table, grad, exp_avg, exp_avg_sq):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.shape != table.shape:
            raise ValueError('table_step_adam: all buffers must be contiguous float32 of the table\'s shape')
    N, D = table.shape
    if table_f16 is not None:
        if table_f16.dtype != torch.float16 or table_f16.shape != table.shape or not table_f16.is_contiguous():
            raise ValueError('table_f16 must be a contiguous float16 tensor of the table\'s shape')
        check(lib.lec_table_step_adam_f16(dptr(table), dptr(grad), dptr(exp_avg), dptr(exp_avg_sq), D, N, D, float(lr),
                                          float(betas[0]), float(betas[1]), float(eps), int(step), float(K_cone or 0.0),
                                          int(bool(riemannian)), int(bool(clip)), dptr(table_f16), stream_ptr()))
        return
    check(lib.lec_table_step_adam(dptr(table), dptr(grad), dptr(exp_avg), dptr(exp_avg_sq), D, N, D, float(lr),
                                  float(betas[0]), float(betas[1]), float(eps), int(step), float(K_cone or 0.0),
                                  int(bool(riemannian)), int(bool(clip)), stream_ptr()))


def table_step_rsgd(table, grad, lr, K_cone=0.1):
    """oe_h.py:1761-1762 / order_embeddings_h.py:764-775: exp_map_x(w, -lr * (1/lambda_x)^2 grad) then clip.  In place."""
    N, D = table.shape
    if not (table.is_contiguous() and grad.is_contiguous() and table.dtype == grad.dtype == torch.float32):
        raise ValueError('table_step_rsgd: contiguous float32 buffers required')
    check(lib.lec_table_step_rsgd(dptr(table), dptr(grad), D, N, D, float(lr), float(K_cone), stream_ptr()))


def adam_flat(param, grad, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0, param_lp=None):
    """torch.optim.Adam arithmetic over one flat fp32 arena, one launch.  `param_lp`: optional flat bf16 shadow of the
    parameters, refreshed in the same pass."""
    n = param.numel()
    for t in (param, grad, exp_avg, exp_avg_sq):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n:
            raise ValueError('adam_flat: contiguous float32 buffers of equal length required')
    if param_lp is not None and (param_lp.dtype != torch.bfloat16 or param_lp.numel() != n or not param_lp.is_contiguous()):
        raise ValueError('adam_flat: param_lp must be a contiguous bf16 buffer of the same length')
    check(lib.lec_adam_flat(dptr(param), dptr(grad), dptr(exp_avg), dptr(exp_avg_sq), n, float(lr), float(betas[0]),
                            float(betas[1]), float(eps), int(step), float(grad_scale), dptr(param_lp), stream_ptr()))


# ------------------------------------------------------------------------------------------------ multi-level CE
class MultiLevelCEFn(torch.autograd.Function):
    """MultiLevelCELoss.forward (network/loss.py:29-38) + backward in one launch."""

    @staticmethod
    def forward(ctx, logits, level_labels, levels, level_weights, class_weights=None):
        z = _rows(logits.float(), 'logits')
        B, Cc = z.shape
        lab = level_labels.to(torch.int64).contiguous()
        L = len(levels)
        sizes = (C.c_int32 * L)(*[int(v) for v in levels])
        wts = (C.c_float * L)(*[float(v) for v in level_weights]) if level_weights is not None else None
        loss = torch.empty(1, dtype=torch.float32, device=z.device)
        g = torch.empty(B, Cc, dtype=torch.float32, device=z.device) if logits.requires_grad else None
        ws = _workspace(z.device, 256 + 4 * 2048)
        if class_weights is not None:
            if class_weights.dtype != torch.float32 or class_weights.numel() != Cc or not class_weights.is_contiguous() or class_weights.device != z.device:
                raise ValueError('class weights must be a contiguous float32 [n_classes] tensor on the logits\' device')
        check(lib.lec_multilevel_ce_fwd_bwd(dptr(z), _ld(z), dptr(lab), B, Cc, C.cast(sizes, C.c_void_p),
                                            C.cast(wts, C.c_void_p) if wts is not None else None, dptr(class_weights), L, dptr(loss),
                                            dptr(g), dptr(ws), ws.numel(), stream_ptr()))
        ctx.g = g; ctx.dt = logits.dtype
        return loss.view(())

    @staticmethod
    def backward(ctx, gl):
        return (ctx.g * gl).to(ctx.dt), None, None, None, None


# ------------------------------------------------------------------------------------------------ per-model fusion records
# The fused paths hand work from one autograd node to another: a convolution leaves the BatchNorm statistics of its output in the
# BatchNorm workspace, a BatchNorm leaves pass 2 of its backward to the convolution behind it, ...  Each hand-off is a record keyed by
# the data_ptr() of the tensor it travels with.  The records (and the workspace the statistics sit in) belong to ONE backbone
# instance: a `FusionContext`, owned by the ResNet module, made current for the duration of its forward and captured by every
# autograd node created there (ctx.fc), whose backward makes it current again.  Two models alive in one process (train + eval copy,
# the reference's oe.py / oe_h.py side by side) or interleaved forward / backward passes cannot see each other's records.
# Ops called outside any model (tests, tools) use the process-wide default context; `_FORKS`, `_FOLDED`, ... are ITS dictionaries.
import os as _os


class FusionContext:
    """forks: backward of a forked block output z = relu(bn(x) + residual) (two consumers: the next block's conv1 and its identity
    branch) -- pass 1 of the BatchNorm backward can run in the epilogue of conv1's data gradient (lec_conv1x1_dgrad_bnfold), which needs
    this layer's saved tensors and the identity branch's gradient: data_ptr of z -> those tensors (filled by forward, 'dres' by the
    backward of the layer that took z as its residual, dropped by this layer's backward).
    folded: data_ptr of a gradient that already is g = mask * (dy + dy2) -> number of partial rows its producer left in the workspace.
    deferred: forward of conv3 -> bn3 (+ identity, ReLU): the convolution first runs as a statistics-only pass (conv1x1_stats_rows) and
    hands BNActFn an UNWRITTEN output tensor: data_ptr of that tensor -> (input rows, weight); BNActFn.forward finalizes the
    statistics and runs the convolution again with the BatchNorm apply in its epilogue, which writes both tensors.
    lazy_ok / lazy_dx: backward of the same pair -- pass 2 of bn3's backward runs inside conv3's weight-gradient kernel
    (lec_conv1x1_wgrad_bnapply).  lazy_ok: data_ptr of a BatchNorm input whose producer convolution can do that -> its input channels;
    lazy_dx: data_ptr of the UNWRITTEN dx BNActFn.backward returned -> what the kernel needs.
    ws_owner: (data_ptr of the tensor whose statistics partials sit in this context's BatchNorm workspace, number of partial rows)."""

    def __init__(self):
        self.forks, self.folded, self.deferred, self.lazy_ok, self.lazy_dx = {}, {}, {}, {}, {}
        self.ws_owner = [0, 0]
        self._ws = {}
        # settings of the forward / backward pair this context serves, copied from the owning backbone when its forward starts (ResNet.forward).
        # They used to be process-wide switches (a C static, class attributes): two trainers in one process, or two threads, then raced.
        self.overlap = None             # resnet.WgradOverlap the convolutions / BatchNorms of this pass report to (None: the process default, False: none)
        self.accumulate = False         # BatchNorm backward ADDS d gamma / d beta (several backward passes per step share the gradient slots)
        self.schedule = _lib.SCHEDULE_DEFAULT   # LEC_SCHEDULE_* of the fp32 forward / data-gradient launches
        self.pass_order = None          # (dict, pass index): running statistics are updated in pass order (see BNActFn.forward)
        self.bn_timer = None            # lists of (start event, end event, bytes | flops) per launch when the owner profiles its step
        self.conv_timer = None
        self.in_flight = False          # a training forward has run and its backward has not reached the stem yet
        self.graph_ref = None           # weakref to the LAST autograd node of that forward (the global average pooling's): dead = the output was dropped, no backward can come

    def busy(self):
        """A forward whose backward has not finished yet owns this context (its records and its BatchNorm workspace).  A forward whose autograd graph
        has been freed without a backward (its output dropped, an exception) no longer does: the last node's weak reference tells."""
        if self.graph_ref is not None and self.graph_ref() is None:
            self.reset()
        return bool(self.in_flight or self.forks or self.folded or self.deferred or self.lazy_ok or self.lazy_dx)

    def reset(self):
        """Drop the records of a forward whose backward never ran (or raised)."""
        for d in (self.forks, self.folded, self.deferred, self.lazy_ok, self.lazy_dx):
            d.clear()
        self.ws_owner[0] = 0
        self.in_flight = False
        self.graph_ref = None

    def workspace(self, device):
        key = (device.type, device.index)
        ws = self._ws.get(key)
        if ws is None:
            ws = torch.zeros(int(lib.lec_bn_workspace_bytes(2048)), dtype=torch.uint8, device=device)
            self._ws[key] = ws
        return ws


_DEFAULT_FUSION = FusionContext()
import threading as _threading
_TLS = _threading.local()               # the stack is per THREAD: autograd runs backward on its own thread, a host may drive two trainers from two


def _stack():
    st = getattr(_TLS, 'stack', None)
    if st is None:
        st = _TLS.stack = [_DEFAULT_FUSION]
    return st


def fusion():
    """The current FusionContext of this thread (the innermost `use_fusion`, else the process-wide default)."""
    return _stack()[-1]


def overlap():
    """The WgradOverlap the current pass reports to: its context's own, else the process default `WgradOverlap.instance` (ops called outside
    a model: tests, tools)."""
    ov = fusion().overlap
    if ov is False:                     # the owner said: none (stock autograd convolutions)
        return None
    if ov is not None:
        return ov
    from .resnet import WgradOverlap
    return WgradOverlap.instance


class use_fusion:
    def __init__(self, fc):
        self.fc = fc if fc is not None else _DEFAULT_FUSION

    def __enter__(self):
        _stack().append(self.fc)
        return self.fc

    def __exit__(self, *exc):
        _stack().pop()
        return False


def _with_ctx_fusion(backward):
    """Decorator for autograd backward()s: run under the FusionContext the node's forward ran in."""
    import functools

    @functools.wraps(backward)
    def wrapped(ctx, *grads):
        with use_fusion(getattr(ctx, 'fc', None)):
            return backward(ctx, *grads)
    return wrapped


# A step that pushes its CNN rows through the backbone as several CONCURRENT passes (engine.StepEngine, one HIP stream per pass) gives each
# forward `pass_order = (dict, pass index)` (ResNet.forward -> FusionContext.pass_order; the module-level PASS_ORDER below is the default
# context's, for ops called outside a model): the running statistics of a BatchNorm layer are then updated in pass order -- pass p's launch
# waits for pass p - 1's launch of the same layer (an event per layer; pass p - 1 is always enqueued first and runs ahead) -- so
# that they hold EMA(EMA(r, batch of pass 0), batch of pass 1), the reference's sequence of forwards, not a race between streams.
PASS_ORDER = None

FOLD_BN_BWD = _os.environ.get('LEC_FOLD_BN_BWD', '1') != '0'
# fp32 counterparts (round 3): pass 1 of a BatchNorm backward in the epilogue of the fp32 data gradient that produces its gradient
# (lec_conv_f32_dgrad_fused), pass 2 on the operand load of the 1x1 convolution behind the BatchNorm (data and weight gradient)
FOLD_BN_BWD_F32 = _os.environ.get('LEC_FOLD_BN_BWD_F32', '1') != '0'
FOLD_BN_BWD_BF16 = _os.environ.get('LEC_FOLD_BN_BWD_BF16', '1') != '0'      # ... and of the bf16 family's data gradients (csrc/conv_bf16.hip)
LAZY_BN_PASS2_F32 = _os.environ.get('LEC_LAZY_BN_PASS2_F32', '1') != '0'
# Where the two forms pay (tools/bench_conv_f32_fused.py, one MI355X, 512 rows; the step now runs as two concurrent half-batch passes, so a
# BatchNorm pass that stays a kernel is about half hidden under the other pass's convolutions, while every microsecond added to a
# convolution is matrix-pipe time the step is bound by):
#   * the fold loses on 64-channel destinations (the narrow-tile kernel's epilogue: +350 / +420 us against passes of 300 us on layer1's
#     conv2 / conv3 data gradients) and wins from 128 channels on (+5 .. +250 us against 155 .. 590 us);
#   * the on-load form adds ~70 us to the data gradient and 50 - 170 us to the weight gradient (its extra operand costs the weight gradient a
#     workgroup per CU), which only a large pass 2 repays: bn3 of layer1 / layer2 (950 / 470 us), not bn1 anywhere (230 .. 30 us) nor
#     bn3 of layer3 / layer4 (220 / 105 us).
FOLD_F32_MIN_CHANNELS = int(_os.environ.get('LEC_FOLD_F32_MIN_CHANNELS', '128'))        # fold into data gradients whose destination has >= this many channels
LAZY_F32_MIN_ELEMS = int(_os.environ.get('LEC_LAZY_F32_MIN_ELEMS', str(80 * 1000 * 1000)))   # on-load pass 2 for BatchNorms of >= this many elements per pass
DEFER_BN_APPLY = _os.environ.get('LEC_DEFER_BN_APPLY', '1') != '0'
LAZY_BN_PASS2 = _os.environ.get('LEC_LAZY_BN_PASS2', '1') != '0'
# the default context's records under their historical names (ops called outside a model: tests, tools)
_FORKS, _FOLDED, _DEFERRED, _LAZY_OK, _LAZY_DX = (_DEFAULT_FUSION.forks, _DEFAULT_FUSION.folded, _DEFAULT_FUSION.deferred,
                                                  _DEFAULT_FUSION.lazy_ok, _DEFAULT_FUSION.lazy_dx)
_BN_WS_OWNER = _DEFAULT_FUSION.ws_owner


# ------------------------------------------------------------------------------------------------ fused BN (+add) (+ReLU)
def _nhwc_rows(t, name):
    """[N, C, H, W] channels_last bf16 / fp32 tensor -> (M, C); raises unless the memory really is [M, C] with C innermost."""
    if t.dtype not in (torch.bfloat16, torch.float32):
        raise TypeError('%s must be bf16 or fp32' % name)
    if t.dim() != 4 or not t.is_contiguous(memory_format=torch.channels_last):
        raise ValueError('%s must be a 4-D channels_last tensor' % name)
    N, Cc, H, W = t.shape
    return N * H * W, Cc


def _dt(name, dtype):
    """The entry point of `name` for activations of `dtype`: bf16 (MI355X-native storage) or fp32 (the reference's precision)."""
    return getattr(lib, name + '_f32') if dtype == torch.float32 else getattr(lib, name)


BN_TIMER = None     # set to a list to record (start_event, end_event, algorithmic_bytes) per fused-BN launch group (bench.py)


def _bn_timed(call, nbytes):
    timer = fusion().bn_timer if fusion().bn_timer is not None else BN_TIMER
    if timer is None:
        return call()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); call(); b.record()
    timer.append((a, b, nbytes))


class BNActFn(torch.autograd.Function):
    """y = [relu](batch_norm(x) [+ residual]) on NHWC bf16, two streaming passes forward, two backward."""

    @staticmethod
    def forward(ctx, x, residual, weight, bias, running_mean, running_var, training, momentum, eps, relu, fork=False, sink=None):
        """fork=True returns (y, y_alias): two handles on the SAME memory for an activation that feeds two branches, so
        that backward receives the two branch gradients separately and the kernels add them on the fly."""
        M, Cc = _nhwc_rows(x, 'x')
        if residual is not None and (_nhwc_rows(residual, 'residual') != (M, Cc) or residual.dtype != x.dtype):
            raise ValueError('residual shape / dtype mismatch')
        ctx.fc = fusion()
        order = fusion().pass_order if fusion().pass_order is not None else PASS_ORDER
        order = order if (training and order is not None and x.is_cuda) else None
        if order is not None and order[1] > 0:
            prev_ev = order[0].get((running_mean.data_ptr(), order[1] - 1))
            if prev_ev is not None:
                torch.cuda.current_stream().wait_event(prev_ev)
        es = x.element_size()                         # 2 (bf16) or 4 (fp32) bytes per activation element
        y = torch.empty_like(x)                       # preserves channels_last
        save_mean = torch.empty(Cc, dtype=torch.float32, device=x.device); save_invstd = torch.empty_like(save_mean)
        ws = _bn_workspace(x.device)
        # ReLU bitmask for backward: 1/16 of the bytes of y (y itself stays alive only as the next conv's input)
        mask = torch.empty(M * (Cc // 8), dtype=torch.uint8, device=x.device) if (relu and training) else None
        el = M * Cc                                   # algorithmic bytes: x (stats) + x + y [+ residual] [+ mask]
        # the convolution that produced x may have left its statistics partials in the workspace (conv1x1_rows)
        prestat = fusion().ws_owner[1] if (training and fusion().ws_owner[0] == x.data_ptr()) else 0
        fusion().ws_owner[0] = 0
        dfr = fusion().deferred.pop(x.data_ptr(), None)
        if dfr is not None:
            x_in, w2 = dfr
            cin = x_in.shape[1]
            fused = (prestat and residual is not None and mask is not None and x.dtype == torch.bfloat16 and residual.dtype == torch.bfloat16
                     and residual.is_contiguous(memory_format=torch.channels_last))
            if fused:
                # statistics are in the workspace: finalize, then the convolution again with scale / shift / residual / ReLU in its
                # epilogue -- it writes x (this layer's input, kept for backward), y and the bitmask; no apply pass
                check(lib.lec_bn_fwd_finalize(M, Cc, dptr(weight), dptr(bias), float(eps), float(momentum), dptr(running_mean), dptr(running_var),
                                              prestat, dptr(save_mean), dptr(save_invstd), dptr(ws), ws.numel(), stream_ptr()))
                off = lib.lec_bn_workspace_coeff_offset(Cc)
                check(lib.lec_conv1x1_fwd_bnapply(dptr(x_in), dptr(w2), M, cin, Cc, C.c_void_p(ws.data_ptr() + off), C.c_void_p(ws.data_ptr() + off + 4 * Cc),
                                                  dptr(residual), dptr(x), dptr(y), dptr(mask), stream_ptr()))
            else:                                 # not the layer the deferral was meant for: materialise the product first
                check(lib.lec_conv1x1_fwd(dptr(x_in), dptr(w2), 0, M, cin, Cc, dptr(x), None, 0, None, stream_ptr()))
                dfr = None
        nbytes = el * es * ((2 if (training and not prestat) else 1) + 1 + (1 if residual is not None else 0)) + (el // 8 if mask is not None else 0)
        if dfr is not None:
            pass                                  # done above, inside the convolution
        elif prestat:
            _bn_timed(lambda: check(_dt('lec_bn_fwd_prestat', x.dtype)(dptr(x), dptr(residual), M, Cc, dptr(weight), dptr(bias), float(eps), float(momentum),
                                                           dptr(running_mean), dptr(running_var), prestat, dptr(save_mean),
                                                           dptr(save_invstd), dptr(y), int(bool(relu)), dptr(mask), dptr(ws), ws.numel(),
                                                           stream_ptr())), nbytes)
        else:
            _bn_timed(lambda: check(_dt('lec_bn_fwd', x.dtype)(dptr(x), dptr(residual), M, Cc, dptr(weight), dptr(bias), float(eps), float(momentum),
                                                   dptr(running_mean), dptr(running_var), int(bool(training)), dptr(save_mean),
                                                   dptr(save_invstd), dptr(y), int(bool(relu)), dptr(mask), dptr(ws), ws.numel(),
                                                   stream_ptr())), nbytes)
        if order is not None:
            done_ev = torch.cuda.Event(); done_ev.record()
            order[0][(running_mean.data_ptr(), order[1])] = done_ev
        if training:
            ctx.save_for_backward(x, mask, weight, save_mean, save_invstd)
            ctx.meta = (M, Cc, bool(relu), residual is not None)
            ctx.sink = sink                       # (weight Parameter, bias Parameter, reducer or None): write d gamma / d beta in place
            ctx.res_ptr = residual.data_ptr() if residual is not None else 0
            ctx.out_ptr = y.data_ptr()
            if fork and FOLD_BN_BWD and residual is not None and ctx.needs_input_grad[0]:
                if len(fusion().forks) > 64:              # forward passes that never ran backward
                    fusion().forks.clear()
                # what the consumer convolution's data gradient needs to run pass 1 of THIS layer's backward in its epilogue
                fusion().forks[y.data_ptr()] = {'x': x, 'mask': mask, 'mean': save_mean, 'invstd': save_invstd, 'dres': None}
            elif (not fork and ctx.needs_input_grad[0] and ((FOLD_BN_BWD_F32 and x.dtype == torch.float32) or (FOLD_BN_BWD_BF16 and x.dtype == torch.bfloat16))):
                # an output with ONE consumer (bn1 -> conv2, bn2 -> conv3): that convolution's data gradient can run pass 1 in its epilogue with
                # no second gradient to wait for (fp32: lec_conv_f32_dgrad_fused; bf16: lec_conv_bf16_dgrad's fold -- a consumer that cannot, e.g. a
                # strided layer or a special-case kernel, ignores the record and this layer runs its own pass 1)
                if len(fusion().forks) > 64:
                    fusion().forks.clear()
                fusion().forks[y.data_ptr()] = {'x': x, 'mask': mask, 'mean': save_mean, 'invstd': save_invstd, 'dres': None, 'single': True}
        if fork:
            return y, y.as_strided(y.size(), y.stride())
        return y

    @staticmethod
    @_with_ctx_fusion
    def backward(ctx, dy, dy2=None):
        x, mask, weight, save_mean, save_invstd = ctx.saved_tensors
        M, Cc, relu, has_res = ctx.meta
        if dy is None:
            dy, dy2 = dy2, None
        if dy is None:
            raise RuntimeError('BNActFn.backward: no incoming gradient')
        if not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.contiguous(memory_format=torch.channels_last)
        if dy2 is not None and not dy2.is_contiguous(memory_format=torch.channels_last):
            dy2 = dy2.contiguous(memory_format=torch.channels_last)
        fusion().forks.pop(ctx.out_ptr, None)
        pre = 0                                   # > 0: dy is already g = mask * (dy + dy2) and its partial sums sit in the workspace
        if dy.data_ptr() in fusion().folded:
            tag = fusion().folded.pop(dy.data_ptr())
            if fusion().ws_owner[0] == dy.data_ptr() and fusion().ws_owner[1] == tag:
                pre = tag
            dy2 = None                            # already folded into dy, and so is the ReLU mask
            relu = False; mask = None
        dx = torch.empty_like(x)
        dres = (dy if pre else torch.empty_like(x)) if has_res else None      # folded: the residual branch's gradient IS g
        sink = ctx.sink
        if sink is not None and sink[0].grad is not None and sink[1].grad is not None:
            dgamma, dbeta = sink[0].grad, sink[1].grad           # the flat arena's slots: no AccumulateGrad kernels
        else:
            sink = None                               # (zeros, not empty: with `accumulate` the kernels ADD into these)
            dgamma = torch.zeros(Cc, dtype=torch.float32, device=x.device); dbeta = torch.zeros_like(dgamma)
        ws = _bn_workspace(x.device)
        fusion().ws_owner[0] = 0
        acc = 1 if fusion().accumulate else 0         # several backward passes of a step share the gradient slots (zeroed once per step)
        el = M * Cc                                   # 2 x (dy [+ dy2] + x [+ mask]) + dx [+ d residual]
        es = x.element_size()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        if dy2 is not None and dy2.dtype != x.dtype:
            dy2 = dy2.to(x.dtype)
        if has_res:      # pass 1 reads dy [+ dy2], x, mask and writes g (= d residual); pass 2 reads g, x and writes dx
            nbytes = el * es * ((2 + (1 if dy2 is not None else 0) + 1) + (2 + 1)) + (el // 8 if relu else 0)
        else:            # both passes read dy [+ dy2], x, mask; pass 2 writes dx
            nbytes = el * es * (2 * (2 + (1 if dy2 is not None else 0)) + 1) + (2 * (el // 8) if relu else 0)
        _ov = overlap()
        lazy = (LAZY_BN_PASS2 and has_res and x.data_ptr() in fusion().lazy_ok and x.dtype == torch.bfloat16
                and _ov is not None and _ov.enabled       # the consumer that materialises dx is _OverlapConvFn.backward
                and lib.lec_conv1x1_wgrad_bnapply_supported(fusion().lazy_ok[x.data_ptr()], Cc, M))
        # fp32: the 1x1 / stride-1 convolution that produced x forms dx on its operand load, in its data gradient AND in its weight gradient
        # (lec_conv_f32_dgrad_fused / _wgrad_fused): pass 2 never runs as a kernel
        lazy32 = (LAZY_BN_PASS2_F32 and x.dtype == torch.float32 and x.data_ptr() in fusion().lazy_ok and el >= LAZY_F32_MIN_ELEMS
                  and _ov is not None and _ov.enabled and ctx.needs_input_grad[0])
        fusion().lazy_ok.pop(x.data_ptr(), None)
        if lazy32:
            coef = torch.empty(3 * Cc, dtype=torch.float32, device=x.device)
            if pre:
                g = dy
                _bn_timed(lambda: check(lib.lec_bn_bwd_coeffs_f32(M, Cc, pre, dptr(weight), dptr(save_mean), dptr(save_invstd), dptr(dgamma), dptr(dbeta),
                                                                  dptr(coef), dptr(ws), ws.numel(), acc, stream_ptr())), 0)
            else:
                g = dres if has_res else torch.empty_like(x)
                nb1 = el * es * (2 + (1 if dy2 is not None else 0) + 1) + (el // 8 if relu else 0)
                _bn_timed(lambda: check(lib.lec_bn_bwd_pass1_coeffs_f32(dptr(dy), dptr(dy2), dptr(mask) if relu else None, dptr(x), M, Cc, dptr(weight),
                                                                        dptr(save_mean), dptr(save_invstd), dptr(g), dptr(dgamma), dptr(dbeta), dptr(coef),
                                                                        dptr(ws), ws.numel(), acc, stream_ptr())), nb1)
            fusion().lazy_dx.clear()
            fusion().lazy_dx[dx.data_ptr()] = {'g': g, 'x': x, 'coef': coef, 'gamma': weight, 'mean': save_mean, 'invstd': save_invstd, 'M': M, 'C': Cc}
        elif lazy:
            # the convolution that produced x runs pass 2 inside its weight-gradient kernel (conv1x1_wgrad_bnapply_rows): here only
            # pass 1 (unless a data-gradient epilogue already did it) and the finalize; dx is handed on UNWRITTEN with a record
            if pre:
                _bn_timed(lambda: check(lib.lec_bn_bwd_finalize(M, Cc, pre, dptr(dgamma), dptr(dbeta), dptr(ws), ws.numel(), acc, stream_ptr())), 0)
            else:
                nb1 = el * es * (2 + (1 if dy2 is not None else 0) + 1) + (el // 8 if relu else 0)
                _bn_timed(lambda: check(_dt('lec_bn_bwd_pass1', x.dtype)(dptr(dy), dptr(dy2), dptr(mask) if relu else None, dptr(x), M, Cc, dptr(save_mean),
                                                             dptr(save_invstd), dptr(dres), dptr(dgamma), dptr(dbeta), dptr(ws), ws.numel(),
                                                             acc, stream_ptr())), nb1)
            fusion().lazy_dx.clear()
            fusion().lazy_dx[dx.data_ptr()] = {'g': dres, 'x': x, 'gamma': weight, 'mean': save_mean, 'invstd': save_invstd, 'M': M, 'C': Cc}
        elif pre:
            nbytes = el * es * 3                  # pass 2 only: read g, x; write dx (pass 1 ran in the convolution's epilogue)
            _bn_timed(lambda: check(_dt('lec_bn_bwd_prereduced', x.dtype)(dptr(dy), dptr(x), M, Cc, dptr(weight), dptr(save_mean), dptr(save_invstd), pre,
                                                              dptr(dx), dptr(dgamma), dptr(dbeta), dptr(ws), ws.numel(), acc, stream_ptr())), nbytes)
        else:
            _bn_timed(lambda: check(_dt('lec_bn_bwd', x.dtype)(dptr(dy), dptr(dy2), None, dptr(mask), dptr(x), M, Cc, dptr(weight), dptr(save_mean),
                                                   dptr(save_invstd), dptr(dx), dptr(dres), dptr(dgamma), dptr(dbeta), int(relu),
                                                   dptr(ws), ws.numel(), acc, stream_ptr())), nbytes)
        if has_res and ctx.res_ptr in fusion().forks:     # this layer's residual is a forked block output: its consumer convolution's data
            fusion().forks[ctx.res_ptr]['dres'] = dres    # gradient can fold this gradient into its epilogue (conv1x1_dgrad_bnfold_rows)
        if sink is not None:
            if sink[2] is not None:
                sink[2].mark_ready(sink[0]); sink[2].mark_ready(sink[1])
            return dx, dres, None, None, None, None, None, None, None, None, None, None
        return dx, dres, dgamma, dbeta, None, None, None, None, None, None, None, None


def conv1x1_supported(cin, cout, M):
    return bool(lib.lec_conv1x1_supported(int(cin), int(cout), int(M)))


def conv3x3_c64(x, w_clast, want_stats=False, w_transposed=False):
    """3x3 / stride 1 / pad 1, C -> C channels (C = 64, or 128 through the tap-streaming kernel) on NHWC bf16.
    x: [N, C, H, W] channels_last, w_clast: [C, C, 3, 3] channels_last (memory [co][r][s][ci]).  w_transposed (C = 64): x is
    a gradient and w_clast the layer's FORWARD weight; the kernel flips and transposes it while loading (data gradient)."""
    n, c, h, w = x.shape
    y = torch.empty_like(x)
    ws = _bn_workspace(x.device) if want_stats else None
    k = C.c_int(0)
    args = (dptr(ws), ws.numel(), C.byref(k)) if want_stats else (None, 0, None)
    if c == 64:
        check(lib.lec_conv3x3_c64_fwd(dptr(x), dptr(w_clast), int(bool(w_transposed)), n, h, w, dptr(y), *args, stream_ptr()))
    else:
        if w_transposed:
            w_clast = w_clast.flip(2, 3).permute(1, 0, 2, 3).contiguous(memory_format=torch.channels_last)
        check(lib.lec_conv3x3_c128_fwd(dptr(x), dptr(w_clast), n, h, w, dptr(y), *args, stream_ptr()))
    if want_stats:
        fusion().ws_owner[0], fusion().ws_owner[1] = y.data_ptr(), k.value
    return y


def conv3x3_c64_wgrad(dy, x, dw):
    """dw (fp32, [64, 64, 3, 3] channels_last, i.e. memory [co][ky][kx][ci]) += weight gradient of the 3x3 / stride 1 / pad 1
    convolution from dy, x [N, 64, H, W] channels_last bf16 (MFMA kernel, float atomics)."""
    n, c, h, w = x.shape
    if c != 64 or dy.shape != x.shape or not (dy.is_contiguous(memory_format=torch.channels_last) and x.is_contiguous(memory_format=torch.channels_last)):
        raise ValueError('conv3x3_c64_wgrad: need NHWC [N, 64, H, W] tensors of equal shape')
    if dw.dtype != torch.float32 or tuple(dw.shape) != (64, 64, 3, 3) or not dw.is_contiguous(memory_format=torch.channels_last):
        raise ValueError('dw must be a float32 [64, 64, 3, 3] channels_last buffer')
    check(lib.lec_conv3x3_c64_wgrad(dptr(dy), dptr(x), n, h, w, dptr(dw), stream_ptr()))
    return dw


def conv1x1_wgrad_supported(cin, cout, M):
    return bool(lib.lec_conv1x1_wgrad_supported(int(cin), int(cout), int(M)))


def conv1x1_wgrad_rows(dy_rows, x_rows, dw):
    """dw[Cout, Cin] (fp32, contiguous) += dy[M, Cout]^T @ x[M, Cin] on the MFMA weight-gradient kernel (float atomics)."""
    M, cout = dy_rows.shape
    cin = x_rows.shape[1]
    if dw.dtype != torch.float32 or dw.numel() != cout * cin or not dw.is_contiguous():
        raise ValueError('dw must be a contiguous float32 [Cout, Cin] buffer')
    check(lib.lec_conv1x1_wgrad(dptr(dy_rows), dptr(x_rows), M, cin, cout, dptr(dw), stream_ptr()))
    return dw


def bn_bwd_apply_lazy(rec, dx):
    """Pass 2 of a BatchNorm backward whose dx was handed on unwritten (fusion().lazy_dx) and whose consumer cannot run it itself."""
    if 'coef' in rec:                             # fp32 form: dx = A g + B x + D from the coefficient vectors (the workspace has moved on)
        C_ = rec['C']; cf = rec['coef'].view(3, 1, C_, 1, 1) if dx.dim() == 4 else rec['coef'].view(3, 1, C_)
        torch.add(torch.addcmul(cf[2].expand_as(dx), rec['x'], cf[1]), rec['g'] * cf[0], out=dx)
        return
    ws = _bn_workspace(dx.device)
    check(_dt('lec_bn_bwd_apply', rec['x'].dtype)(dptr(rec['g']), dptr(rec['x']), rec['M'], rec['C'], dptr(rec['gamma']), dptr(rec['mean']), dptr(rec['invstd']),
                               dptr(dx), dptr(ws), ws.numel(), stream_ptr()))


def conv1x1_wgrad_bnapply_rows(rec, x_rows, dx, dw):
    """dx (the UNWRITTEN gradient of a 1x1 layer's output, [N, Cout, H, W] channels_last) := pass 2 of the BatchNorm backward
    described by rec (fusion().lazy_dx), and dw [Cout, Cin] (fp32) += dx^T x_rows, in one kernel (lec_conv1x1_wgrad_bnapply)."""
    M, cin = x_rows.shape
    cout = rec['C']
    if dw.dtype != torch.float32 or dw.numel() != cout * cin or not dw.is_contiguous():
        raise ValueError('dw must be a contiguous float32 [Cout, Cin] buffer')
    ws = _bn_workspace(dx.device)
    off = lib.lec_bn_workspace_coeff_offset(cout)
    check(lib.lec_conv1x1_wgrad_bnapply(dptr(rec['g']), dptr(rec['x']), dptr(x_rows), M, cin, cout, dptr(rec['gamma']), dptr(rec['mean']),
                                        dptr(rec['invstd']), C.c_void_p(ws.data_ptr() + off), C.c_void_p(ws.data_ptr() + off + 4 * cout),
                                        dptr(dx), dptr(dw), stream_ptr()))


def conv1x1_bnapply_supported(cin, cout, M):
    return bool(lib.lec_conv1x1_bnapply_supported(int(cin), int(cout), int(M)))


def conv1x1_stats_rows(x_rows, w2):
    """Statistics-only pass of y = x w^T (lec_conv1x1_stats): returns an UNWRITTEN [M, Cout] tensor whose per-channel partials sit
    in the BatchNorm workspace; BNActFn.forward, handed that tensor with a residual and ReLU, runs the convolution again with the
    BatchNorm apply in its epilogue (and fills the tensor), or materialises the plain product if it is used differently."""
    M, cin = x_rows.shape
    cout = w2.shape[0]
    y = torch.empty((M, cout), dtype=torch.bfloat16, device=x_rows.device)
    ws = _bn_workspace(x_rows.device)
    n = C.c_int(0)
    check(lib.lec_conv1x1_stats(dptr(x_rows), dptr(w2), M, cin, cout, dptr(ws), ws.numel(), C.byref(n), stream_ptr()))
    fusion().ws_owner[0], fusion().ws_owner[1] = y.data_ptr(), n.value
    fusion().deferred.clear()
    fusion().deferred[y.data_ptr()] = (x_rows, w2)
    if len(fusion().lazy_ok) > 64:
        fusion().lazy_ok.clear()
    fusion().lazy_ok[y.data_ptr()] = cin              # this layer's backward can run the following BatchNorm's pass 2 in its weight gradient
    return y


def materialise_deferred(y):
    """A consumer other than the fused BatchNorm got hold of a tensor conv1x1_stats_rows left unwritten (stock-torch fallback of
    BatchNormAct2d, a hook, a debugger): run the plain convolution into it now.  No-op for every other tensor."""
    dfr = fusion().deferred.pop(y.data_ptr(), None)
    if dfr is not None:
        x_in, w2 = dfr
        if fusion().ws_owner[0] == y.data_ptr():
            fusion().ws_owner[0] = 0
        check(lib.lec_conv1x1_fwd(dptr(x_in), dptr(w2), 0, x_in.shape[0], x_in.shape[1], w2.shape[0], dptr(y), None, 0, None, stream_ptr()))
    return y


def conv1x1_dgrad_bnfold_supported(cin, cout, M):
    return bool(lib.lec_conv1x1_dgrad_bnfold_supported(int(cin), int(cout), int(M)))


def conv1x1_dgrad_bnfold_rows(gy_rows, w2_fwd, entry):
    """Data gradient of a 1x1 layer whose input is a forked block output, with pass 1 of that output's BatchNorm backward in
    the epilogue (lec_conv1x1_dgrad_bnfold).  gy_rows [M, Cconv_out], w2_fwd the layer's forward weight [Cconv_out, C],
    entry: the fusion().forks record of the block output.  Returns g [M, C] = mask * (gy w + d identity) and tags it for BNActFn."""
    M, cin = gy_rows.shape
    cout = w2_fwd.shape[1]
    if w2_fwd.shape[0] != cin:
        raise ValueError('weight shape %s does not match %d gradient channels' % (tuple(w2_fwd.shape), cin))
    xb, dres, mask = entry['x'], entry['dres'], entry['mask']
    for t in (xb, dres):
        if t.dtype != torch.bfloat16 or t.numel() != M * cout or not t.is_contiguous(memory_format=torch.channels_last):
            raise ValueError('fork record does not match the gradient: expected NHWC bf16 with %d x %d elements' % (M, cout))
    g = torch.empty((M, cout), dtype=torch.bfloat16, device=gy_rows.device)
    ws = _bn_workspace(gy_rows.device)
    n = C.c_int(0)
    check(lib.lec_conv1x1_dgrad_bnfold(dptr(gy_rows), dptr(w2_fwd), 1, M, cin, cout, dptr(dres), dptr(xb), dptr(mask), dptr(entry['mean']),
                                       dptr(entry['invstd']), dptr(g), dptr(ws), ws.numel(), C.byref(n), stream_ptr()))
    fusion().ws_owner[0], fusion().ws_owner[1] = g.data_ptr(), n.value
    fusion().folded.clear()
    fusion().folded[g.data_ptr()] = n.value
    return g


def conv1x1_rows(x_rows, w2, want_stats=False, w_transposed=False):
    """y[M, Cout] = x[M, Cin] @ w[Cout, Cin]^T on the hand-written MFMA kernel (lec_conv1x1_fwd); with want_stats the
    per-channel sum / sum-of-squares partials of y are left in the BatchNorm workspace for the BN that follows
    (BNActFn checks the ownership tag before trusting them).  w_transposed: w2 is [Cin, Cout] (a layer's forward weight
    used for its data gradient) and the kernel transposes it while loading."""
    M, cin = x_rows.shape
    cout = w2.shape[1] if w_transposed else w2.shape[0]
    if (w2.shape[0] if w_transposed else w2.shape[1]) != cin:
        raise ValueError('weight shape %s does not match %d input channels' % (tuple(w2.shape), cin))
    y = torch.empty((M, cout), dtype=torch.bfloat16, device=x_rows.device)
    if want_stats:
        ws = _bn_workspace(x_rows.device)
        n = C.c_int(0)
        check(lib.lec_conv1x1_fwd(dptr(x_rows), dptr(w2), int(bool(w_transposed)), M, cin, cout, dptr(y), dptr(ws), ws.numel(), C.byref(n), stream_ptr()))
        fusion().ws_owner[0], fusion().ws_owner[1] = y.data_ptr(), n.value
    else:
        check(lib.lec_conv1x1_fwd(dptr(x_rows), dptr(w2), int(bool(w_transposed)), M, cin, cout, dptr(y), None, 0, None, stream_ptr()))
    return y


# ------------------------------------------------------------------------------------------------ fp32 convolutions
def _nhwc_f32(t, name):
    if t.dtype != torch.float32 or t.dim() != 4 or not t.is_contiguous(memory_format=torch.channels_last):
        raise ValueError('%s must be a 4-D channels_last float32 tensor' % name)
    return t


def conv_f32_supported(conv_or_cin, cout=None):
    """Shapes lec_conv_f32_* serve: channel counts that are powers of two >= 4, stride 1 or 2, square dilation-1 group-1 filters."""
    if cout is None:
        c = conv_or_cin
        return (c.groups == 1 and c.dilation == (1, 1) and c.stride[0] == c.stride[1] and c.stride[0] in (1, 2) and c.bias is None
                and c.padding[0] == c.padding[1] and c.padding[0] < c.kernel_size[0] and c.kernel_size[0] == c.kernel_size[1]
                and conv_f32_supported(c.in_channels, c.out_channels))
    pw2 = lambda v: v >= 4 and (v & (v - 1)) == 0
    return pw2(conv_or_cin) and pw2(cout)


CONV_TIMER = None    # set to a list to record (start_event, end_event, flops) per lec_conv_f32_* launch (bench.py); events are recorded on
                     # the stream the kernel is launched on (the weight gradients run on WgradOverlap's side stream)


def _conv_timed(call, flops):
    timer = fusion().conv_timer if fusion().conv_timer is not None else CONV_TIMER
    if timer is None:
        return call()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); call(); b.record()
    timer.append((a, b, flops))


_conv_scratch_bufs = {}
_conv_scratch_lock = _threading.Lock()


def _conv_scratch():
    """Register the balanced convolution kernel's scratch for the current stream (lec_conv_f32_scratch) the first time the stream launches
    a convolution.  Never inside a graph capture (an allocation there belongs to the graph's pool): the engine's eager warm-up steps run
    first, on the same streams."""
    s = torch.cuda.current_stream()
    key = (s.device.index, s.cuda_stream)
    if key in _conv_scratch_bufs or torch.cuda.is_current_stream_capturing():
        return
    with _conv_scratch_lock:                # two host threads launching on one stream register ONE buffer
        if key in _conv_scratch_bufs:
            return
        buf = torch.zeros(int(lib.lec_conv_f32_scratch_bytes()), dtype=torch.uint8, device=s.device)
        check(lib.lec_conv_f32_scratch(stream_ptr(), dptr(buf), buf.numel()))
        _conv_scratch_bufs[key] = buf


def conv_f32_stem_supported(x):
    """The fp32 stem's own forward kernel serves this input ([N, 4, H, W] channels_last fp32, 3 real channels): even H, W in {64, 128, 224}, tensors < 2 GiB."""
    return x.dim() == 4 and x.shape[1] == 4 and bool(lib.lec_conv_f32_stem_supported(int(x.shape[0]), int(x.shape[2]), int(x.shape[3])))


def conv_f32_stem_fwd(x, w, want_stats=False):
    """The ResNet stem (7x7 / stride 2 / pad 3, -> 64 channels) of a 3-channel image stored with 4 channels per pixel, exact fp32 (lec_conv_f32_stem_fwd):
    channels 0..2 of x [N, 4, H, W] and w [64, 4, 7, 7] (channels_last fp32) enter the product; channel 3 is never multiplied."""
    _nhwc_f32(x, 'x'); _nhwc_f32(w, 'w')
    n, cin, h, wd = x.shape
    if cin != 4 or tuple(w.shape) != (64, 4, 7, 7):
        raise ValueError('stem kernel: x [N, 4, H, W], w [64, 4, 7, 7]')
    y = torch.empty((n, 64, h // 2, wd // 2), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    flops = 2.0 * n * (h // 2) * (wd // 2) * 64 * 3 * 49
    if want_stats:
        ws = _bn_workspace(x.device); k = C.c_int(0)
        _conv_timed(lambda: check(lib.lec_conv_f32_stem_fwd(dptr(x), dptr(w), n, h, wd, dptr(y), dptr(ws), ws.numel(), C.byref(k), stream_ptr())), flops)
        fusion().ws_owner[0], fusion().ws_owner[1] = y.data_ptr(), k.value
    else:
        _conv_timed(lambda: check(lib.lec_conv_f32_stem_fwd(dptr(x), dptr(w), n, h, wd, dptr(y), None, 0, None, stream_ptr())), flops)
    return y


def conv_f32_fwd(x, w, stride, pad, want_stats=False):
    """y = conv2d(x, w) in exact fp32 on the f32 MFMA (lec_conv_f32_fwd).  x [N, Cin, H, W], w [Cout, Cin, R, S], both
    channels_last fp32.  want_stats: the BatchNorm statistics partials of y are left in the BatchNorm workspace."""
    _nhwc_f32(x, 'x'); _nhwc_f32(w, 'w')
    n, cin, h, wd = x.shape; cout, _, r, s_ = w.shape
    ho, wo = (h + 2 * pad - r) // stride + 1, (wd + 2 * pad - s_) // stride + 1
    _conv_scratch()
    y = torch.empty((n, cout, ho, wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    flops = 2.0 * n * ho * wo * cout * cin * r * s_
    if want_stats:
        ws = _bn_workspace(x.device); k = C.c_int(0)
        _conv_timed(lambda: check(lib.lec_conv_f32_fwd(dptr(x), dptr(w), n, h, wd, cin, cout, r, s_, stride, pad, dptr(y), dptr(ws), ws.numel(),
                                                       C.byref(k), fusion().schedule, stream_ptr())), flops)
        fusion().ws_owner[0], fusion().ws_owner[1] = y.data_ptr(), k.value
    else:
        _conv_timed(lambda: check(lib.lec_conv_f32_fwd(dptr(x), dptr(w), n, h, wd, cin, cout, r, s_, stride, pad, dptr(y), None, 0, None, fusion().schedule, stream_ptr())), flops)
    return y


def conv_f32_fwd_affine(x, w, stride, pad, scale, shift, residual=None, relu=False):
    """lec_conv_f32_fwd_affine: y = [relu](conv2d(x, w) * scale[c] + shift[c] [+ residual]) -- an eval-mode BatchNorm (+ the block's
    residual add and ReLU) in the convolution's epilogue.  x, w, residual channels_last fp32; scale / shift fp32 [Cout]."""
    _nhwc_f32(x, 'x'); _nhwc_f32(w, 'w')
    n, cin, h, wd = x.shape; cout, _, r, s_ = w.shape
    ho, wo = (h + 2 * pad - r) // stride + 1, (wd + 2 * pad - s_) // stride + 1
    if scale.numel() != cout or shift.numel() != cout or scale.dtype != torch.float32 or shift.dtype != torch.float32:
        raise ValueError('conv_f32_fwd_affine: scale / shift must be fp32 vectors of Cout entries')
    if residual is not None and tuple(_nhwc_f32(residual, 'residual').shape) != (n, cout, ho, wo):
        raise ValueError('conv_f32_fwd_affine: residual must have the output\'s shape')
    y = torch.empty((n, cout, ho, wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    _conv_scratch()
    _conv_timed(lambda: check(lib.lec_conv_f32_fwd_affine(dptr(x), dptr(w), n, h, wd, cin, cout, r, s_, stride, pad, dptr(y), dptr(scale), dptr(shift),
                                                          dptr(residual), 1 if relu else 0, fusion().schedule, stream_ptr())), 2.0 * n * ho * wo * cout * cin * r * s_)
    return y


def conv_f32_dgrad(dy, w, x_shape, stride, pad):
    """dx of the same convolution (lec_conv_f32_dgrad): dy [N, Cout, Ho, Wo], w the FORWARD weight, x_shape = (N, Cin, H, W)."""
    _nhwc_f32(dy, 'dy'); _nhwc_f32(w, 'w')
    n, cin, h, wd = x_shape; cout, _, r, s_ = w.shape
    dx = torch.empty((n, cin, h, wd), dtype=torch.float32, device=dy.device, memory_format=torch.channels_last)
    _conv_scratch()
    _conv_timed(lambda: check(lib.lec_conv_f32_dgrad(dptr(dy), dptr(w), n, h, wd, cin, cout, r, s_, stride, pad, dptr(dx), fusion().schedule, stream_ptr())),
                2.0 * dy.shape[0] * dy.shape[2] * dy.shape[3] * cout * cin * r * s_)
    return dx


def conv_f32_dgrad_fused(dy, w, x_shape, stride, pad, xf=None, fold=None):
    """lec_conv_f32_dgrad_fused.  xf = (xsrc, coef): `dy` is g and the kernel forms the gradient A g + B xsrc + D on load (1x1 layers);
    fold = a FusionContext.forks record {'x', 'mask', 'mean', 'invstd', 'dres'}: the result is g = mask * (dx + dres) of the BatchNorm whose
    output this layer consumed, tagged for BNActFn.backward, its partial sums in the BatchNorm workspace."""
    _nhwc_f32(dy, 'dy'); _nhwc_f32(w, 'w')
    n, cin, h, wd = x_shape; cout, _, r, s_ = w.shape
    dx = torch.empty((n, cin, h, wd), dtype=torch.float32, device=dy.device, memory_format=torch.channels_last)
    _conv_scratch()
    xs = cf = None
    if xf is not None:
        xs, cf = xf; _nhwc_f32(xs, 'xsrc')
        if xs.shape != dy.shape or cf.numel() != 3 * cout:
            raise ValueError('on-load form: xsrc must have dy\'s shape and coef 3 * Cout entries')
    a = [None] * 5; ws = None; k = C.c_int(0)
    if fold is not None:
        xb = _nhwc_f32(fold['x'], 'fold x')
        if tuple(xb.shape) != (n, cin, h, wd) or (fold['dres'] is not None and tuple(_nhwc_f32(fold['dres'], 'dres').shape) != (n, cin, h, wd)):
            raise ValueError('fold record does not match the layer input')
        ws = _bn_workspace(dy.device)
        a = [dptr(fold['dres']), dptr(xb), dptr(fold['mask']), dptr(fold['mean']), dptr(fold['invstd'])]
    _conv_timed(lambda: check(lib.lec_conv_f32_dgrad_fused(dptr(dy), dptr(w), n, h, wd, cin, cout, r, s_, stride, pad, dptr(dx), dptr(xs), dptr(cf),
                                                           a[0], a[1], a[2], a[3], a[4], dptr(ws), ws.numel() if ws is not None else 0,
                                                           C.byref(k) if ws is not None else None, fusion().schedule, stream_ptr())),
                2.0 * dy.shape[0] * dy.shape[2] * dy.shape[3] * cout * cin * r * s_)
    if fold is not None:
        fc = fusion()
        fc.ws_owner[0], fc.ws_owner[1] = dx.data_ptr(), k.value
        fc.folded.clear(); fc.folded[dx.data_ptr()] = k.value
    return dx


def conv_f32_wgrad_c3(dy, x4, dw3, stride, pad):
    """The stem's weight gradient: x4 [N, 4, H, W] (zero 4th channel), dw3 [Cout, 3, R, S] channels_last fp32, dw3 += (float atomics)."""
    _nhwc_f32(dy, 'dy'); _nhwc_f32(x4, 'x4'); _nhwc_f32(dw3, 'dw3')
    n, c4, h, wd = x4.shape; cout, c3, r, s_ = dw3.shape
    if c4 != 4 or c3 != 3:
        raise ValueError('conv_f32_wgrad_c3: x4 must have 4 channels and dw3 3')
    _conv_timed(lambda: check(lib.lec_conv_f32_wgrad_c3(dptr(dy), dptr(x4), n, h, wd, cout, r, s_, stride, pad, dptr(dw3), stream_ptr())),
                2.0 * dy.shape[0] * dy.shape[2] * dy.shape[3] * cout * 4 * r * s_)
    return dw3


def conv_f32_wgrad(dy, x, dw, stride, pad, xf=None):
    """dw += weight gradient (lec_conv_f32_wgrad, float atomics).  dw [Cout, Cin, R, S] channels_last fp32.  xf = (xsrc, coef): `dy`
    is g and the gradient is formed on load (lec_conv_f32_wgrad_fused, 1x1 / stride 1 layers)."""
    _nhwc_f32(dy, 'dy'); _nhwc_f32(x, 'x'); _nhwc_f32(dw, 'dw')
    n, cin, h, wd = x.shape; cout, _, r, s_ = dw.shape
    xs, cf = xf if xf is not None else (None, None)
    _conv_timed(lambda: check(lib.lec_conv_f32_wgrad_fused(dptr(dy), dptr(x), n, h, wd, cin, cout, r, s_, stride, pad, dptr(dw), dptr(xs), dptr(cf),
                                                           stream_ptr())),
                2.0 * dy.shape[0] * dy.shape[2] * dy.shape[3] * cout * cin * r * s_)
    return dw


# ------------------------------------------------------------------------------------------------ bf16 convolutions (csrc/conv_bf16.hip)
def _nhwc_bf16(t, name):
    if t.dtype != torch.bfloat16 or t.dim() != 4 or not t.is_contiguous(memory_format=torch.channels_last):
        raise ValueError('%s must be a 4-D channels_last bfloat16 tensor' % name)
    return t


def conv_bf16_supported(conv):
    """Layers lec_conv_bf16_* serve: every layer of ResNet-18 / -50 (the 3-channel stem through zero channels 3..7)."""
    cin = 8 if conv.in_channels == 3 else conv.in_channels
    return (conv.groups == 1 and conv.dilation == (1, 1) and conv.stride[0] == conv.stride[1] and conv.bias is None
            and conv.padding[0] == conv.padding[1] and conv.kernel_size[0] == conv.kernel_size[1]
            and bool(lib.lec_conv_bf16_supported(cin, conv.out_channels, conv.kernel_size[0], conv.kernel_size[1], conv.stride[0], conv.padding[0])))


def conv_bf16_wt(w):
    """w [Cout, Cin, R, S] channels_last bf16 (memory [Cout][RS][Cin]) -> the data gradient's operand, memory [Cin][RS][Cout], returned as a
    channels_last [Cin, Cout, R, S] tensor (lec_conv_bf16_wt_transpose)."""
    _nhwc_bf16(w, 'w')
    cout, cin, r, s_ = w.shape
    wt = torch.empty((cin, cout, r, s_), dtype=torch.bfloat16, device=w.device, memory_format=torch.channels_last)
    check(lib.lec_conv_bf16_wt_transpose(dptr(w), dptr(wt), cout, r * s_, cin, stream_ptr()))
    return wt


def conv_bf16_stem_supported(x):
    """The stem's own forward kernel serves this input ([N, 8, H, W] channels_last bf16, 3 real channels): even H, W in {64, 128, 224}."""
    return x.dim() == 4 and x.shape[1] == 8 and bool(lib.lec_conv_bf16_stem_supported(int(x.shape[2]), int(x.shape[3])))


def conv_bf16_stem_fwd(x, w, want_stats=False):
    """The ResNet stem (7x7 / stride 2 / pad 3, -> 64 channels) of a 3-channel image stored with 8 channels per pixel (lec_conv_bf16_stem_fwd): channels 0..3 of
    x [N, 8, H, W] and w [64, 8, 7, 7] (channels_last bf16) enter the product -- channel 3 must be zero in one of them -- channels 4..7 are never read."""
    _nhwc_bf16(x, 'x'); _nhwc_bf16(w, 'w')
    n, cin, h, wd = x.shape
    if cin != 8 or tuple(w.shape) != (64, 8, 7, 7):
        raise ValueError('stem kernel: x [N, 8, H, W], w [64, 8, 7, 7]')
    y = torch.empty((n, 64, h // 2, wd // 2), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    flops = 2.0 * n * (h // 2) * (wd // 2) * 64 * 3 * 49
    if want_stats:
        ws = _bn_workspace(x.device); k = C.c_int(0)
        _conv_timed(lambda: check(lib.lec_conv_bf16_stem_fwd(dptr(x), dptr(w), n, h, wd, dptr(y), dptr(ws), ws.numel(), C.byref(k), stream_ptr())), flops)
        fusion().ws_owner[0], fusion().ws_owner[1] = y.data_ptr(), k.value
    else:
        _conv_timed(lambda: check(lib.lec_conv_bf16_stem_fwd(dptr(x), dptr(w), n, h, wd, dptr(y), None, 0, None, stream_ptr())), flops)
    return y


def conv_bf16_fwd(x, w, stride, pad, want_stats=False):
    """y = conv2d(x, w) on bf16 NHWC tensors, fp32 accumulation (lec_conv_bf16_fwd).  want_stats: the BatchNorm statistics partials of the
    (rounded) output are left in the BatchNorm workspace."""
    _nhwc_bf16(x, 'x'); _nhwc_bf16(w, 'w')
    n, cin, h, wd = x.shape; cout, _, r, s_ = w.shape
    ho, wo = (h + 2 * pad - r) // stride + 1, (wd + 2 * pad - s_) // stride + 1
    y = torch.empty((n, cout, ho, wo), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    flops = 2.0 * n * ho * wo * cout * cin * r * s_
    if want_stats:
        ws = _bn_workspace(x.device); k = C.c_int(0)
        _conv_timed(lambda: check(lib.lec_conv_bf16_fwd(dptr(x), dptr(w), n, h, wd, cin, cout, r, s_, stride, pad, dptr(y), dptr(ws), ws.numel(),
                                                        C.byref(k), stream_ptr())), flops)
        fusion().ws_owner[0], fusion().ws_owner[1] = y.data_ptr(), k.value
    else:
        _conv_timed(lambda: check(lib.lec_conv_bf16_fwd(dptr(x), dptr(w), n, h, wd, cin, cout, r, s_, stride, pad, dptr(y), None, 0, None, stream_ptr())), flops)
    return y


def conv_bf16_dgrad(dy, wt, x_shape, stride, pad, fold=None):
    """dx of the same convolution from the TRANSPOSED weights wt (conv_bf16_wt).  fold = a FusionContext.forks record {'x', 'mask', 'mean',
    'invstd', 'dres'} (stride-1 layers): the result is g = mask * (dx + dres) of the BatchNorm whose output this layer consumed, tagged for
    BNActFn.backward, its partial sums in the BatchNorm workspace."""
    _nhwc_bf16(dy, 'dy'); _nhwc_bf16(wt, 'wt')
    n, cin, h, wd = x_shape; _, cout, r, s_ = wt.shape
    if wt.shape[0] != cin or dy.shape[1] != cout:
        raise ValueError('conv_bf16_dgrad: wt %s does not match Cin %d / Cout %d' % (tuple(wt.shape), cin, dy.shape[1]))
    dx = torch.empty((n, cin, h, wd), dtype=torch.bfloat16, device=dy.device, memory_format=torch.channels_last)
    a = [None] * 5; ws = None; k = C.c_int(0)
    if fold is not None:
        xb = _nhwc_bf16(fold['x'], 'fold x')
        if tuple(xb.shape) != (n, cin, h, wd) or (fold['dres'] is not None and tuple(_nhwc_bf16(fold['dres'], 'dres').shape) != (n, cin, h, wd)):
            raise ValueError('fold record does not match the layer input')
        ws = _bn_workspace(dy.device)
        a = [dptr(fold['dres']), dptr(xb), dptr(fold['mask']), dptr(fold['mean']), dptr(fold['invstd'])]
    _conv_timed(lambda: check(lib.lec_conv_bf16_dgrad(dptr(dy), dptr(wt), n, h, wd, cin, cout, r, s_, stride, pad, dptr(dx), a[0], a[1], a[2], a[3], a[4],
                                                      dptr(ws), ws.numel() if ws is not None else 0, C.byref(k) if ws is not None else None, stream_ptr())),
                2.0 * dy.shape[0] * dy.shape[2] * dy.shape[3] * cout * cin * r * s_)
    if fold is not None:
        fc = fusion()
        fc.ws_owner[0], fc.ws_owner[1] = dx.data_ptr(), k.value
        fc.folded.clear(); fc.folded[dx.data_ptr()] = k.value
    return dx


def conv_bf16_wgrad(dy, x, dw, stride, pad):
    """dw += weight gradient (lec_conv_bf16_wgrad, float atomics).  dw [Cout, Cin', R, S] channels_last fp32 with Cin' = x's channels, or fewer
    for a zero-padded stem input (x [N, 8, H, W], dw [Cout, 3, R, S])."""
    _nhwc_bf16(dy, 'dy'); _nhwc_bf16(x, 'x')
    if dw.dtype != torch.float32 or dw.dim() != 4 or not dw.is_contiguous(memory_format=torch.channels_last):
        raise ValueError('dw must be a 4-D channels_last float32 tensor')
    n, cin, h, wd = x.shape; cout, dcin, r, s_ = dw.shape
    _conv_timed(lambda: check(lib.lec_conv_bf16_wgrad(dptr(dy), dptr(x), n, h, wd, cin, cout, r, s_, stride, pad, dptr(dw), dcin, stream_ptr())),
                2.0 * dy.shape[0] * dy.shape[2] * dy.shape[3] * cout * cin * r * s_)
    return dw


class X3Planes:
    """Pre-split bf16 planes of one conv weight (lec_conv_f32x3_split_weights): `fwd` feeds conv_f32x3_fwd, `t` conv_f32x3_dgrad."""
    __slots__ = ('shape', 'fwd', 't')

    def __init__(self, shape, device, want_t=True):
        cout, cin, r, s_ = shape
        self.shape = tuple(shape)
        self.fwd = torch.empty(int(lib.lec_conv_f32x3_planes_elems(cout, r * s_, cin, 0)), dtype=torch.int16, device=device)
        self.t = torch.empty(int(lib.lec_conv_f32x3_planes_elems(cout, r * s_, cin, 1)), dtype=torch.int16, device=device) if want_t else None

    def update(self, w):
        _nhwc_f32(w, 'w')
        cout, cin, r, s_ = self.shape
        if tuple(w.shape) != self.shape:
            raise ValueError('weight shape %s does not match the planes %s' % (tuple(w.shape), self.shape))
        check(lib.lec_conv_f32x3_split_weights(dptr(w), cout, r * s_, cin, dptr(self.fwd), dptr(self.t) if self.t is not None else None, stream_ptr()))
        return self


def conv_f32x3_split_weights(w, want_t=True):
    """The bf16 planes of a channels_last fp32 conv weight [Cout, Cin, R, S] (run again after every change of w)."""
    return X3Planes(w.shape, w.device, want_t).update(w)


def conv_f32x3_fwd(x, planes, stride, pad, want_stats=False):
    """y = conv2d(x, w) with fp32 products on the bf16 matrix cores (lec_conv_f32x3_fwd); planes = conv_f32x3_split_weights(w)."""
    _nhwc_f32(x, 'x')
    n, cin, h, wd = x.shape; cout, cin_w, r, s_ = planes.shape
    if cin_w != cin:
        raise ValueError('weight planes have %d input channels, x has %d' % (cin_w, cin))
    ho, wo = (h + 2 * pad - r) // stride + 1, (wd + 2 * pad - s_) // stride + 1
    y = torch.empty((n, cout, ho, wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    flops = 2.0 * n * ho * wo * cout * cin * r * s_
    if want_stats:
        ws = _bn_workspace(x.device); k = C.c_int(0)
        _conv_timed(lambda: check(lib.lec_conv_f32x3_fwd(dptr(x), dptr(planes.fwd), n, h, wd, cin, cout, r, s_, stride, pad, dptr(y), dptr(ws), ws.numel(),
                                                         C.byref(k), stream_ptr())), flops)
        fusion().ws_owner[0], fusion().ws_owner[1] = y.data_ptr(), k.value
    else:
        _conv_timed(lambda: check(lib.lec_conv_f32x3_fwd(dptr(x), dptr(planes.fwd), n, h, wd, cin, cout, r, s_, stride, pad, dptr(y), None, 0, None, stream_ptr())), flops)
    return y


def conv_f32x3_dgrad(dy, planes, x_shape, stride, pad):
    """dx of the same convolution (lec_conv_f32x3_dgrad)."""
    _nhwc_f32(dy, 'dy')
    n, cin, h, wd = x_shape; cout, _, r, s_ = planes.shape
    if planes.t is None:
        raise ValueError('these planes were split without the data-gradient layout')
    dx = torch.empty((n, cin, h, wd), dtype=torch.float32, device=dy.device, memory_format=torch.channels_last)
    _conv_timed(lambda: check(lib.lec_conv_f32x3_dgrad(dptr(dy), dptr(planes.t), n, h, wd, cin, cout, r, s_, stride, pad, dptr(dx), stream_ptr())),
                2.0 * dy.shape[0] * dy.shape[2] * dy.shape[3] * cout * cin * r * s_)
    return dx


def conv_f32x3_wgrad_supported(cin, cout, r=1, s_=1):
    return bool(lib.lec_conv_f32x3_wgrad_supported(cin, cout, r, s_))


def conv_f32x3_wgrad_preferred(cin, cout, r=1, s_=1):
    """Where the split weight gradient beats the f32-MFMA one: from 128 channels on both sides (1.5 - 1.9x).  64-channel layers run
    half-empty 128 x 128 tiles and only reach parity (measured: 64 -> 64 3x3 @56: 1634 vs 1684 us; 128 -> 64: 2920 vs 2809)."""
    return cin >= 128 and cout >= 128 and conv_f32x3_wgrad_supported(cin, cout, r, s_)


def conv_f32x3_wgrad(dy, x, dw, stride, pad):
    """dw += weight gradient with fp32 products on the bf16 matrix cores (lec_conv_f32x3_wgrad, float atomics)."""
    _nhwc_f32(dy, 'dy'); _nhwc_f32(x, 'x'); _nhwc_f32(dw, 'dw')
    n, cin, h, wd = x.shape; cout, _, r, s_ = dw.shape
    _conv_timed(lambda: check(lib.lec_conv_f32x3_wgrad(dptr(dy), dptr(x), n, h, wd, cin, cout, r, s_, stride, pad, dptr(dw), stream_ptr())),
                2.0 * dy.shape[0] * dy.shape[2] * dy.shape[3] * cout * cin * r * s_)
    return dw


def _bn_workspace(device):
    """The current FusionContext's BatchNorm workspace on `device` (statistics partials + coefficient table)."""
    return fusion().workspace(device)


# ------------------------------------------------------------------------------------------------ the stem's tail as one op
FUSE_STEM_POOL = _os.environ.get('LEC_FUSE_STEM_POOL', '1') != '0'


def stem_pool_supported(x, bn):
    """maxpool(relu(bn(x))) of the stem as BNReluPoolFn: fp32 NHWC output of an fp32 convolution that left its statistics partials in the workspace."""
    if not (FUSE_STEM_POOL and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)):
        return False
    N, Cc, H, W = x.shape
    return (Cc % 8 == 0 and Cc <= 512 and 256 % (Cc // 8) == 0 and H % 2 == 0 and W % 2 == 0 and bn.training and bn.fuse_relu
            and bn.weight.dtype == torch.float32 and fusion().ws_owner[0] == x.data_ptr() and fusion().ws_owner[1] > 0)


class BNReluPoolFn(torch.autograd.Function):
    """p = maxpool3x3s2(relu(batch_norm(x))) for the fp32 stem (torchvision ResNet `maxpool(relu(bn1(conv1(x))))`, oe_h.py:311,317), train mode, the
    statistics partials of x already in the workspace (the convolution's epilogue): finalize + ONE pooling launch that normalises on load; backward = two
    launches that rebuild the pooling's input gradient on the fly (lec_bn_relu_maxpool_fwd_f32 / _bwd_f32).  Same p, argmax, running statistics as
    BNActFn + MaxPool3x3s2Fn, bit for bit; the gradients agree to summation order."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, sink=None, fork=False):
        """fork=True returns (p, p_alias), two handles on the same memory for the two branches of the first block: backward then receives the two branch
        gradients separately and the kernels add them on load (no accumulation pass over the pooled gradient)."""
        N, Cc, H, W = x.shape
        M = N * H * W
        ctx.fc = fusion()
        order = fusion().pass_order if fusion().pass_order is not None else PASS_ORDER
        if order is not None and order[1] > 0:                # running statistics in pass order (see PASS_ORDER)
            prev_ev = order[0].get((running_mean.data_ptr(), order[1] - 1))
            if prev_ev is not None:
                torch.cuda.current_stream().wait_event(prev_ev)
        prestat = fusion().ws_owner[1]
        fusion().ws_owner[0] = 0
        save_mean = torch.empty(Cc, dtype=torch.float32, device=x.device); save_invstd = torch.empty_like(save_mean)
        ws = _bn_workspace(x.device)
        p = torch.empty((N, Cc, H // 2, W // 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        arg = torch.empty(N * (H // 2) * (W // 2) * Cc, dtype=torch.uint8, device=x.device)
        off = lib.lec_bn_workspace_coeff_offset(Cc)

        def run():
            check(lib.lec_bn_fwd_finalize(M, Cc, dptr(weight), dptr(bias), float(eps), float(momentum), dptr(running_mean), dptr(running_var),
                                          prestat, dptr(save_mean), dptr(save_invstd), dptr(ws), ws.numel(), stream_ptr()))
            check(lib.lec_bn_relu_maxpool_fwd_f32(dptr(x), N, H, W, Cc, C.c_void_p(ws.data_ptr() + off), C.c_void_p(ws.data_ptr() + off + 4 * Cc),
                                                  dptr(p), dptr(arg), stream_ptr()))
        _bn_timed(run, M * Cc * 4 + (M * Cc // 4) * 5)          # read x; write p and its argmax bytes
        if order is not None:
            done_ev = torch.cuda.Event(); done_ev.record()
            order[0][(running_mean.data_ptr(), order[1])] = done_ev
        ctx.save_for_backward(x, arg, weight, bias, save_mean, save_invstd)
        ctx.sink = sink
        if fork:
            return p, p.as_strided(p.size(), p.stride())
        return p

    @staticmethod
    @_with_ctx_fusion
    def backward(ctx, dp, dp2=None):
        x, arg, weight, bias, save_mean, save_invstd = ctx.saved_tensors
        N, Cc, H, W = x.shape
        if dp is None:
            dp, dp2 = dp2, None
        if dp is None:
            raise RuntimeError('BNReluPoolFn.backward: no incoming gradient')

        def nhwc(t):
            if t is None:
                return None
            t = t.to(x.dtype) if t.dtype != x.dtype else t
            return t if t.is_contiguous(memory_format=torch.channels_last) else t.contiguous(memory_format=torch.channels_last)
        dp, dp2 = nhwc(dp), nhwc(dp2)
        dx = torch.empty_like(x)
        sink = ctx.sink
        if sink is not None and sink[0].grad is not None and sink[1].grad is not None:
            dgamma, dbeta = sink[0].grad, sink[1].grad           # the flat arena's slots: no AccumulateGrad kernels
        else:
            sink = None                                          # (zeros, not empty: with `accumulate` the kernels ADD into these)
            dgamma = torch.zeros(Cc, dtype=torch.float32, device=x.device); dbeta = torch.zeros_like(dgamma)
        ws = _bn_workspace(x.device)
        fusion().ws_owner[0] = 0
        acc = 1 if fusion().accumulate else 0
        el = N * H * W * Cc
        _bn_timed(lambda: check(lib.lec_bn_relu_maxpool_bwd_f32(dptr(dp), dptr(dp2), dptr(arg), dptr(x), N, H, W, Cc, dptr(weight), dptr(bias), dptr(save_mean),
                                                                dptr(save_invstd), dptr(dx), dptr(dgamma), dptr(dbeta), dptr(ws), ws.numel(), acc,
                                                                stream_ptr())), el * 4 * 3 + (el // 4) * 5 * 2)   # x twice, dx; dp + argmax twice
        if sink is not None:
            if sink[2] is not None:
                sink[2].mark_ready(sink[0]); sink[2].mark_ready(sink[1])
            return dx, None, None, None, None, None, None, None, None
        return dx, dgamma, dbeta, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------------ stem max pooling
class MaxPool3x3s2Fn(torch.autograd.Function):
    """3x3 / stride 2 / pad 1 max pooling on NHWC bf16 with a one-byte argmax and a gather backward."""

    @staticmethod
    def forward(ctx, x):
        N, Cc, H, W = x.shape
        y = torch.empty((N, Cc, H // 2, W // 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        arg = torch.empty(N * (H // 2) * (W // 2) * Cc, dtype=torch.uint8, device=x.device)
        check(_dt('lec_maxpool3x3s2_fwd', x.dtype)(dptr(x), N, H, W, Cc, dptr(y), dptr(arg), stream_ptr()))
        ctx.save_for_backward(arg); ctx.shape = (N, Cc, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        N, Cc, H, W = ctx.shape
        if not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty((N, Cc, H, W), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
        check(_dt('lec_maxpool3x3s2_bwd', dy.dtype)(dptr(dy), dptr(arg), N, H, W, Cc, dptr(dx), stream_ptr()))
        return dx
