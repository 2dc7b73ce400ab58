"""ResNet-18 / ResNet-50 backbones owned by this package (torchvision is not in the image, and the conv stack is the
MFMA half of the hot path: SURVEY.md 8a row a3).  Parameter / buffer names follow torchvision's so that the reference's
checkpoints (`models.resnet18(pretrained=True)` state dicts, oe_h.py:311, finetuner.py:121-122) load unchanged.

Layout / precision are MI355X choices, not the reference's: activations NHWC (channels_last) so MIOpen / our kernels see
the GEMM-friendly layout, bf16 compute under autocast with fp32 master weights held in one flat arena
(parallel.FlatArena) so the data-parallel all-reduce is a single RCCL collective and Adam a single launch.
"""
import os
import weakref
import torch
import torch.nn as nn
import torch.nn.functional as F


class BatchNormAct2d(nn.BatchNorm2d):
    """BatchNorm2d [+ residual add] [+ ReLU] as ONE op.  Same parameters / buffers / state-dict keys as nn.BatchNorm2d.
    On the MI355X with NHWC bf16 activations it runs the fused HIP kernels of liblecone.so (csrc/bn.hip: two streaming
    passes forward, two backward, instead of BN + add + ReLU framework kernels); fp32 activations (``--dtype fp32``) and
    the CPU baseline leg of bench.py take the stock torch ops.  `num_batches_tracked` is not advanced (it only matters
    for momentum=None, which the reference never uses)."""

    fused_enabled = True            # class-wide switch (tests / A-B profiling): False -> stock torch ops everywhere

    def __init__(self, num_features, relu=False):
        super().__init__(num_features)
        self.fuse_relu = relu

    def eval_affine(self):
        """(scale, shift) of the eval-mode map y = x * scale + shift: gamma / sqrt(running_var + eps), beta - running_mean * scale, computed by a
        small launch ON THE STREAM into two vectors this layer owns (lec_bn_eval_coeffs_f32) -- every call: the parameters live in the flat
        arena and the running statistics are written by liblecone's kernels through raw pointers, so no tensor version counter would tell the
        host that they moved; in stream order the launch reads what the optimizer step / the training forwards before it wrote, and a captured
        inference graph recomputes the vectors on every replay (their addresses never change)."""
        buf = getattr(self, '_affine', None)
        if buf is None or buf.device != self.weight.device:
            buf = self._affine = torch.empty((2, self.num_features), dtype=torch.float32, device=self.weight.device)
        from . import ops
        ops.check(ops.lib.lec_bn_eval_coeffs_f32(self.num_features, ops.dptr(self.weight), ops.dptr(self.bias), float(self.eps), ops.dptr(self.running_mean),
                                                 ops.dptr(self.running_var), ops.dptr(buf[0]), ops.dptr(buf[1]), ops.stream_ptr()))
        return buf[0], buf[1]

    def forward(self, x, residual=None, fork=False):
        """fork=True: return TWO handles (y, y_alias) on the output, one per consuming branch of the next block (its conv
        path and its identity / downsample path): the fused backward then adds the two branch gradients on the fly."""
        if (BatchNormAct2d.fused_enabled and x.is_cuda and x.dtype in (torch.bfloat16, torch.float32) and x.dim() == 4 and self.num_features % 8 == 0
                and self.num_features <= 2048 and x.is_contiguous(memory_format=torch.channels_last)
                and (residual is None or (residual.dtype == x.dtype and residual.shape == x.shape
                                          and residual.is_contiguous(memory_format=torch.channels_last)))):
            from . import ops
            ov = ops.overlap()
            sink = (self.weight, self.bias, ov.reducer) if (ov is not None and ov.enabled and ov.arena is not None and self.training and not ov.accumulate) else None
            return ops.BNActFn.apply(x, residual, self.weight, self.bias, self.running_mean, self.running_var,
                                     self.training, self.momentum, self.eps, self.fuse_relu, fork and self.training, sink)
        if x.is_cuda:
            _ops().materialise_deferred(x)          # conv3 may have handed its output on unwritten, expecting the fused layer above
        y = F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, self.training, self.momentum, self.eps)
        if residual is not None:
            y = y + residual
        y = F.relu(y) if self.fuse_relu else y
        return (y, y) if fork else y


class WgradOverlap:
    """Runs every convolution's weight-gradient kernel on a second HIP stream, concurrently with the rest of backward.

    Backward of the backbone is a chain  ... -> BN backward (HBM-bound) -> conv data-gradient (MFMA-bound) -> ...; the
    weight gradients hang off that chain and nothing downstream needs them until the all-reduce / optimizer step.
    Issuing them on a side stream lets MFMA-bound wgrad kernels share the GPU with the HBM-bound BatchNorm kernels of the
    following layers instead of lengthening the chain.  The side stream adds the result into the parameter's `.grad`
    (the flat arena view) and tells the gradient reducer the parameter is ready; `join()` makes the main stream wait."""

    instance = None     # the PROCESS DEFAULT, for ops called outside a model (tests, tools) and models that name no overlap of their own.  A
                        # trainer / engine gives its backbone its own (`ResNet.wgrad_overlap`): nothing on the step path writes this attribute.

    def __init__(self, reducer=None, arena=None, side_stream=True):
        """`arena` (a FlatArena with a bf16 shadow) lets convolutions read their weights in low precision without a cast
        kernel and write their weight gradient straight into the arena; `side_stream=False` keeps the wgrad kernels on
        the main stream (only the cast/accumulate kernels are saved)."""
        self.side = torch.cuda.Stream() if side_stream else None
        self.reducer = reducer
        self.arena = arena
        self.enabled = True
        # fp32, ONE pass, weight gradients on the side stream: data gradient and weight gradient are both bound by the matrix pipe -- side by side they only
        # stretch each other (config 4, round 5: 137.8 ms with the side stream, 137.2 in line).  `antiphase`: the side stream's weight gradient of layer L runs
        # beside the main stream's HBM-bound BatchNorm backward of layer L - 1, and the NEXT data gradient waits for it (wait_matrix) -- matrix work never
        # overlaps matrix work, the BatchNorm backward passes disappear under the weight gradients.
        self.antiphase = False
        self._wgrad_done = None
        self.accumulate = False         # True: several backward passes per step (chunked CNN rows): BatchNorm parameter gradients go
                                        # through autograd's AccumulateGrad (which adds) instead of being written in place
        if reducer is not None and self.side is not None and hasattr(reducer, 'side_streams'):
            reducer.side_streams.append(self.side)

    def weight_lp(self, conv, dtype):
        w16 = self.arena.lowp_view(conv.weight) if (self.arena is not None and dtype == torch.bfloat16) else None
        return w16 if w16 is not None else conv.weight.to(dtype)

    def weight_lp_t(self, conv, w16):
        """The bf16 weights of `conv` in the data gradient's layout [Cin][RS][Cout]: the arena's transposed twin where it keeps one
        (FlatArena.enable_lowp_transposed: refreshed once per optimizer step, by one launch behind the Adam kernel), else transposed per call."""
        if self.arena is not None:
            lp = self.arena.lowp_view(conv.weight)
            wt = self.arena.lowp_t_view(conv.weight) if lp is not None and lp.data_ptr() == w16.data_ptr() else None
            if wt is not None:
                return wt
        return _ops().conv_bf16_wt(w16)

    def _finish_wgrad(self, gw, conv):
        w = conv.weight
        if w.grad is None:
            w.grad = gw.to(w.dtype)
        else:
            w.grad.add_(gw)                                 # ADD, like liblecone's own weight-gradient kernels and autograd's AccumulateGrad:
                                                            # the arena is zeroed at the start of a step, and a step may run several backward
                                                            # passes (CNN rows processed in chunks)
        if self.reducer is not None:
            self.reducer.mark_ready(w)

    @staticmethod
    def _materialise(g, xf):
        """The gradient a fused kernel would have formed on load, as a tensor (a weight gradient that falls back to the library)."""
        dy = torch.empty_like(g)
        _ops().bn_bwd_apply_lazy({'g': g, 'x': xf[0], 'coef': xf[1], 'C': g.shape[1]}, dy)
        return dy

    def _own_wgrad(self, gy, x, conv, xf=None):
        """The wide 1x1 layers: liblecone's MFMA weight-gradient kernel adds dY^T X straight into the arena's fp32 gradient
        slot (zeroed at the start of the step) -- no library kernel, zero-fill, cast or copy.  False when not applicable."""
        w = conv.weight
        if (MFMA_F32 and gy.dtype == torch.float32 and x.dtype == torch.float32 and w.grad is not None and w.grad.dtype == torch.float32
                and w.grad.shape == w.shape and _f32_conv_ok(conv) and _f32_conv_fits(conv, x) and w.grad.is_contiguous(memory_format=torch.channels_last)
                and gy.is_contiguous(memory_format=torch.channels_last) and x.is_contiguous(memory_format=torch.channels_last)):
            if conv.in_channels == 3:             # the stem: x carries the zero 4th channel (_pad_c4), which has no slot in the gradient;
                                                  # atomics straight into the 3-channel slot (a separate add would race between
                                                  # concurrent backward passes)
                _ops().conv_f32_wgrad_c3(gy, x, w.grad, conv.stride[0], conv.padding[0])
            elif F32_MODE == 'x3' and _ops().conv_f32x3_wgrad_preferred(conv.in_channels, conv.out_channels, *conv.kernel_size):
                _ops().conv_f32x3_wgrad(gy, x, w.grad, conv.stride[0], conv.padding[0])  # bf16 matrix cores (layers of >= 128 channels)
            else:
                _ops().conv_f32_wgrad(gy, x, w.grad, conv.stride[0], conv.padding[0], xf=xf)   # f32 MFMA, atomics straight into the gradient slot
            if self.reducer is not None:
                self.reducer.mark_ready(w)
            return True
        if self._bf16_family_wgrad(gy, x, conv, first=True):
            return True
        if (MFMA_WGRAD_3X3 and self.arena is not None and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1)
                and conv.dilation == (1, 1) and conv.groups == 1 and conv.in_channels == 64 and conv.out_channels == 64
                and w.grad is not None and w.grad.dtype == torch.float32 and w.grad.is_contiguous(memory_format=torch.channels_last)
                and gy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and gy.shape == x.shape
                and gy.is_contiguous(memory_format=torch.channels_last) and x.is_contiguous(memory_format=torch.channels_last)
                and x.shape[2] % 8 == 0 and x.shape[3] % 8 == 0):
            _ops().conv3x3_c64_wgrad(gy, x, w.grad)                # layer1's 3x3: MFMA kernel, fp32 atomics into the arena slot
            if self.reducer is not None:
                self.reducer.mark_ready(w)
            return True
        if not (MFMA_WGRAD and self.arena is not None and _is_pointwise(conv) and w.grad is not None and w.grad.dtype == torch.float32
                and gy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and CONV_BF16 != '2'
                and gy.is_contiguous(memory_format=torch.channels_last) and x.is_contiguous(memory_format=torch.channels_last)):
            return self._bf16_family_wgrad(gy, x, conv)
        ops = _ops()
        m = gy.shape[0] * gy.shape[2] * gy.shape[3]
        if not ops.conv1x1_wgrad_supported(conv.in_channels, conv.out_channels, m) or (max(conv.in_channels, conv.out_channels) > 512 and not MFMA_WGRAD_L3):
            return self._bf16_family_wgrad(gy, x, conv)
        gw = w.grad.view(conv.out_channels, conv.in_channels)      # a 1x1 weight is [Cout][Cin] in memory in either layout
        ops.conv1x1_wgrad_rows(_rows(gy), _rows(x), gw)
        if self.reducer is not None:
            self.reducer.mark_ready(w)
        return True

    def _bf16_family_wgrad(self, gy, x, conv, first=False):
        """The bf16 implicit-GEMM family's weight gradient (lec_conv_bf16_wgrad: float atomics into the arena's fp32 slot, or into a fresh fp32
        buffer that is then added like a library result).  first=True: asked BEFORE the special-case kernels (LEC_CONV_BF16=2 only)."""
        if CONV_BF16 == '0' or (first and CONV_BF16 != '2'):
            return False
        w = conv.weight
        if not (gy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and gy.dim() == 4 and _ops().conv_bf16_supported(conv) and _bf16_conv_fits(conv, x)
                and gy.is_contiguous(memory_format=torch.channels_last) and x.is_contiguous(memory_format=torch.channels_last)):
            return False
        slot = w.grad if (w.grad is not None and w.grad.dtype == torch.float32 and w.grad.shape == w.shape
                          and (w.grad.is_contiguous(memory_format=torch.channels_last) or _is_pointwise(conv))) else None
        if slot is not None:
            tgt = slot if slot.is_contiguous(memory_format=torch.channels_last) else slot.view(w.shape).contiguous(memory_format=torch.channels_last)
            if tgt.data_ptr() != slot.data_ptr():
                slot = None
        if slot is not None:
            _ops().conv_bf16_wgrad(gy, x, tgt, conv.stride[0], conv.padding[0])
            if self.reducer is not None:
                self.reducer.mark_ready(w)
            return True
        gw = torch.zeros(w.shape, dtype=torch.float32, device=w.device).contiguous(memory_format=torch.channels_last)
        _ops().conv_bf16_wgrad(gy, x, gw, conv.stride[0], conv.padding[0])
        self._finish_wgrad(gw, conv)
        return True

    def submit(self, gy, x, w16, conv, xf=None):
        """xf = (xsrc, coef): gy holds g and the kernel forms the gradient on load (fp32 1x1 layers, lec_conv_f32_wgrad_fused)."""
        if self.side is None:
            if self._own_wgrad(gy, x, conv, xf):
                return
            if xf is not None:
                gy = self._materialise(gy, xf)
            if x.shape[1] != conv.in_channels:                  # (a zero-padded stem input: the library sees the layer's own 3 channels)
                x = x[:, :conv.in_channels]; w16 = w16[:, :conv.in_channels]
            _lib_launch('wgrad')
            gw = torch.ops.aten.convolution_backward(gy, x, w16, None, conv.stride, conv.padding, conv.dilation, False,
                                                     [0, 0], conv.groups, [False, True, False])[1]
            self._finish_wgrad(gw, conv)
            return
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event(); ev.record(main)
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            own = self._own_wgrad(gy, x, conv, xf)
            if xf is not None:
                for t in xf:
                    t.record_stream(self.side)
            if not own and xf is not None:
                gy = self._materialise(gy, xf)
            if not own:
                _lib_launch('wgrad')
                pad_ = x.shape[1] != conv.in_channels
                gw = torch.ops.aten.convolution_backward(gy, x[:, :conv.in_channels] if pad_ else x, w16[:, :conv.in_channels] if pad_ else w16, None, conv.stride, conv.padding,
                                                         conv.dilation, False, [0, 0], conv.groups, [False, True, False])[1]
            for t in (gy, x, w16):
                t.record_stream(self.side)                 # the caching allocator must not recycle them under the side stream
                                                           # (under hipGraph capture such blocks are held until the capture ends)
            if not own:
                self._finish_wgrad(gw, conv)
            if self.antiphase:
                self._wgrad_done = torch.cuda.Event(); self._wgrad_done.record(self.side)

    def wait_matrix(self):
        """(antiphase) the main stream's next matrix-bound kernel starts after the side stream's last weight gradient has finished."""
        if self.antiphase and self._wgrad_done is not None:
            torch.cuda.current_stream().wait_event(self._wgrad_done)
            self._wgrad_done = None

    def join(self):
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)


# A 1x1 stride-1 convolution on NHWC activations IS the GEMM [N*H*W, Cin] x [Cin, Cout].  Per shape and direction the step
# calls whichever library kernel measured faster on the MI355X at the bench batch (tools/bench_conv1x1.py,
# profiles/r01_conv1x1_gemm.md): hipBLASLt wins the data gradient from Cin >= 256 (e.g. 256->64 @56x56: 396 -> 267 us,
# 256->1024 @14x14: 107 -> 60 us) and the forward from Cin >= 1024; MIOpen keeps the narrow layers and every weight
# gradient (a transposed-A GEMM with K = N*H*W is 2-30x slower in hipBLASLt).
GEMM_1X1 = os.environ.get('LEC_CONV1X1_GEMM', '1') != '0'
# ... and the wide HBM-bound 1x1 layers (Cin 64 / 128 / 256 at 56x56 and 28x28) run liblecone's own MFMA kernel
# (csrc/conv_mfma.hip): forward with the BatchNorm statistics of the output in its epilogue, and the data gradient as the
# same kernel on the transposed weights.  tools/bench_conv1x1_fused.py: 64->256 @56x56 169 us against MIOpen's 227 us,
# and the BatchNorm that follows drops its statistics pass (447 -> 302 us).
MFMA_1X1 = os.environ.get('LEC_CONV1X1_MFMA', '1') != '0'
# layer1's 3x3 convolution (64 -> 64 @56x56) runs liblecone's LDS-halo MFMA kernel (lec_conv3x3_c64_fwd: 4x8-pixel tiles, the
# 6x10 input halo loaded once into LDS, nine taps as the K loop, BatchNorm statistics in the epilogue; the data gradient is
# the same kernel on the flipped, transposed weights): 185 us forward / 168 us data gradient against MIOpen's 298 / 350 us.
MFMA_3X3 = os.environ.get('LEC_CONV3X3_MFMA', '1') != '0'
# the 128 -> 128 @28x28 instance (weights streamed through LDS tap by tap) trails MIOpen forward (214 vs 170 us) and only
# leads it in the data gradient (207 vs 245 us); no measurable gain inside the step, so it stays opt-in:
# '0' (default) nowhere, 'dgrad' data gradient only, '1' everywhere
MFMA_3X3_C128 = os.environ.get('LEC_CONV3X3_C128', '0')
# the weight gradients of the same wide 1x1 layers and of layer1's 3x3 convolution: liblecone's MFMA kernels (lec_conv1x1_wgrad,
# lec_conv3x3_c64_wgrad) accumulate straight into the arena's fp32 gradient slot with float atomics -- alone 179 us against the
# library's 204 us (+ cast and copy kernels) at 64 -> 256 @56x56, 222 against 385 us for the 3x3.  While the second stream still
# had slack they changed nothing in the step (46.4 ms either way); since the BatchNorm passes moved into the convolutions the step
# is the sum of its kernels and they are worth 0.4 ms (44.5 -> 44.1 ms).  Float atomics make the summation ORDER (not the value to
# fp32 tolerance) vary from run to run, as the library's own split-K weight gradients do.
MFMA_WGRAD = os.environ.get('LEC_CONV1X1_WGRAD', '1') != '0'
MFMA_WGRAD_3X3 = os.environ.get('LEC_CONV3X3_WGRAD', '1') != '0'
# layer3's 256 <-> 1024 pairs (14 x 14): 104 against the library's 115 us alone, but 44.0 against 43.5 ms in the step (its 131 KB,
# eight-wave workgroups sit on every CU beside layer3's short main-stream kernels): off
MFMA_WGRAD_L3 = os.environ.get('LEC_CONV1X1_WGRAD_L3', '0') != '0'
# fp32 activations (the reference's precision): every convolution runs liblecone's f32-MFMA implicit-GEMM family (csrc/conv_f32.hip)
MFMA_F32 = os.environ.get('LEC_CONV_F32', '1') != '0'
# 'native': the f32-input MFMA (exact fp32 fmaf chains, 157 TFLOP/s peak).  'x3': the same fp32 products on the bf16 matrix cores
# (csrc/conv_f32x3.hip: three bf16 pieces per operand, six exact products per fp32 product, fp32 accumulation -- an fp32 dot product's
# error, measured against fp64 in tests/test_fp32_gpu.py) for the forward and the data gradient; the weight gradient stays native.
F32_MODE = os.environ.get('LEC_CONV_F32_MODE', 'native')
GEMM_FWD_MIN_CIN = 1024
GEMM_DGRAD_MIN_CIN = 256
# bf16 activations (config 5's 16-bit conv stack): every layer that has no special-case kernel above runs liblecone's bf16 implicit-GEMM family
# (csrc/conv_bf16.hip: forward with BatchNorm statistics, data gradient with the BatchNorm-backward fold, weight gradient) -- no library convolution is
# left on the 16-bit path.  '1' (default): special-case kernels where they exist, the family elsewhere; '2': the family everywhere (A/B runs);
# '0': the library (MIOpen / CK / hipBLASLt) for everything the special cases do not serve, as until round 5 (A/B runs only).
CONV_BF16 = os.environ.get('LEC_CONV_BF16', '1')
# 1x1 data gradients on maps of <= 28 x 28 pixels with >= 128 channels on both sides: the family's whole-line LDS-DMA kernel instead of the round-1 1x1 kernel
# (same-box table at 256 rows, profiles/EXPERIMENTS.md round 6 (3): 48.8 vs 71.0 us at 512 -> 128, 57.6 vs 82.1 at 128 -> 512; the larger maps stay: 142 vs 113 at 56 x 56)
FAMILY_DGRAD_SMALL_MAPS = os.environ.get('LEC_FAMILY_DGRAD_SMALL_MAPS', '1') == '1'
# Convolutions / GEMMs handed to a LIBRARY (aten.convolution, aten.convolution_backward, torch.mm) by this module since the last reset, by direction.
# A step captured into a hipGraph counts once, at capture.  bench.py prints it per step; the fp32 and bf16 config tests assert zero.
LIBRARY_LAUNCHES = {'fwd': 0, 'dgrad': 0, 'wgrad': 0, 'module': 0}


def library_launches(reset=False):
    n = dict(LIBRARY_LAUNCHES)
    if reset:
        for k in LIBRARY_LAUNCHES:
            LIBRARY_LAUNCHES[k] = 0
    return n


LIBRARY_LAUNCHES_TOTAL = [0]   # never reset: every library convolution / GEMM of the process, inside steps or not (bench.py prints it)


def _lib_launch(kind):
    LIBRARY_LAUNCHES[kind] += 1
    LIBRARY_LAUNCHES_TOTAL[0] += 1


def _ops():
    from . import ops
    return ops


def _f32_conv_ok(conv):
    """Layers lec_conv_f32_* serve: everything in ResNet-18 / -50 (the 3-channel stem through a zero 4th channel)."""
    ops = _ops()
    if conv.in_channels == 3:
        return (conv.groups == 1 and conv.dilation == (1, 1) and conv.bias is None and conv.stride[0] == conv.stride[1] and conv.stride[0] in (1, 2)
                and conv.padding[0] == conv.padding[1] and conv.kernel_size[0] == conv.kernel_size[1] and conv.padding[0] < conv.kernel_size[0]
                and ops.conv_f32_supported(4, conv.out_channels))
    return ops.conv_f32_supported(conv)


def _f32_conv_fits(conv, x):
    """lec_conv_f32_* address a tensor with 32-bit byte offsets: one launch serves the images whose input and output stay below 2 GiB (ResNet at 224 x 224 in
    fp32: 668 rows), and the entry points split a larger batch into groups of images themselves (csrc/conv_f32.hip, LEC_CONV_GROUPS) -- up to four groups:
    the statistics / fold partials of all groups share the BatchNorm workspace's 2 048 rows.  The split (x3) mode has no group form: one launch only."""
    n, c, h, w = x.shape
    c = max(c, 4)
    k, st, pd = conv.kernel_size[0], conv.stride[0], conv.padding[0]
    ho, wo = (h + 2 * pd - k) // st + 1, (w + 2 * pd - k) // st + 1
    per_img = max(h * w * c * 4, ho * wo * conv.out_channels * 4)
    g = ((1 << 31) - 1) // max(per_img, 1)
    if g < 1:
        return False
    return -(-n // g) <= (1 if F32_MODE == 'x3' else 4)


def _pad_c4(t):
    """[N, 3, H, W] -> [N, 4, H, W] channels_last with a zero 4th channel (already 4 channels: returned as is)."""
    if t.shape[1] == 4:
        return t
    out = torch.empty((t.shape[0], 4, t.shape[2], t.shape[3]), dtype=t.dtype, device=t.device, memory_format=torch.channels_last)
    out[:, 3:].zero_(); out[:, :3] = t
    return out


def _pad_c8(t):
    """[N, 3, H, W] -> [N, 8, H, W] channels_last with zero channels 3..7 (the bf16 family's 16-byte piece is 8 channels)."""
    if t.shape[1] == 8:
        return t
    out = torch.empty((t.shape[0], 8, t.shape[2], t.shape[3]), dtype=t.dtype, device=t.device, memory_format=torch.channels_last)
    out[:, t.shape[1]:].zero_(); out[:, :t.shape[1]] = t
    return out


def _bf16_conv_fits(conv, x):
    """lec_conv_bf16_* address a tensor with 32-bit byte offsets: input and output must stay below 2 GiB (ResNet-50 at 224 x 224 in bf16: 1 337 rows)."""
    n, c, h, w = x.shape
    c = max(c, 8)
    k, st, pd = conv.kernel_size[0], conv.stride[0], conv.padding[0]
    ho, wo = (h + 2 * pd - k) // st + 1, (w + 2 * pd - k) // st + 1
    return n * h * w * c * 2 < (1 << 31) and n * ho * wo * conv.out_channels * 2 < (1 << 31)


def _is_pointwise(conv):
    return (conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.dilation == (1, 1)
            and conv.groups == 1)


def _rows(t):
    """[N, C, H, W] channels_last -> the [N*H*W, C] matrix it is in memory (a view)."""
    n, c, h, w = t.shape
    return t.permute(0, 2, 3, 1).reshape(n * h * w, c)


def _from_rows(m, n, h, w):
    return m.view(n, h, w, m.shape[1]).permute(0, 3, 1, 2)


_WALK_MODULES = False       # conv_macs(): every convolution through its module's forward (hooks), none through the fused conv + BatchNorm call


def _inference_f32(conv, x):
    """An fp32 channels_last CUDA forward that nobody will differentiate, of a layer lec_conv_f32_* serve."""
    return (MFMA_F32 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and conv.weight.dtype == torch.float32
            and not (torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad))
            and x.is_contiguous(memory_format=torch.channels_last) and _f32_conv_ok(conv) and _f32_conv_fits(conv, x))


def conv_bn(conv, bn, x, residual=None, fork=False):
    """bn(conv(x) [, residual]) -- in an eval-mode fp32 inference forward as ONE kernel: the BatchNorm is a per-channel affine map of its
    running statistics and runs, with the residual add and the ReLU, in the convolution's epilogue (lec_conv_f32_fwd_affine)."""
    if (not bn.training and not _WALK_MODULES and BatchNormAct2d.fused_enabled and isinstance(conv, Conv2d) and _inference_f32(conv, x) and bn.weight.dtype == torch.float32
            and bn.running_mean is not None and bn.running_mean.dtype == torch.float32
            and (residual is None or (residual.dtype == x.dtype and residual.is_contiguous(memory_format=torch.channels_last)))):
        scale, shift = bn.eval_affine()
        w = conv.weight if conv.weight.is_contiguous(memory_format=torch.channels_last) else conv.weight.contiguous(memory_format=torch.channels_last)
        if conv.in_channels == 3:
            x, w = _pad_c4(x), _pad_c4(w)
        y = _ops().conv_f32_fwd_affine(x, w, conv.stride[0], conv.padding[0], scale, shift, residual, bn.fuse_relu)
        return (y, y) if fork else y
    if residual is None and not fork:
        return bn(conv(x))
    return bn(conv(x), residual, fork)


AVGPOOL_FN = os.environ.get('LEC_AVGPOOL_FN', '1') != '0'        # (0: the framework's backward, for A/B runs)


class _GlobalAvgPoolFn(torch.autograd.Function):
    """flatten(AdaptiveAvgPool2d(1)(x)) with a backward that writes the gradient of a channels_last x IN channels_last: dx[n, :, h, w] = g[n, :] / (H W),
    one coalesced broadcast store.  The framework's backward produces an NCHW-contiguous tensor (34 us) which the BatchNorm backward behind it then
    converts (a 7 x 7-row transpose: 216 us for [256, 2048, 7, 7] at 0.5 TB/s) -- on the step's critical path between forward and backward, with nothing
    of either pass to overlap with.  Forward is the framework's kernel (same bits as before)."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = x.shape
        ctx.cl = x.is_contiguous(memory_format=torch.channels_last)
        # this node is the LAST one of the backbone's graph: it dies when the caller drops the forward's output without a backward -- the fusion context's
        # own records (block outputs) keep every node upstream of it alive, so it is the one witness of "a backward can still come" (FusionContext.busy)
        _ops().fusion().graph_ref = weakref.ref(ctx)
        return torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)

    @staticmethod
    def backward(ctx, g):
        n, c, h, w = ctx.shape
        dx = torch.empty(ctx.shape, dtype=g.dtype, device=g.device, memory_format=torch.channels_last if ctx.cl else torch.contiguous_format)
        dx.copy_((g / float(h * w)).view(n, c, 1, 1).expand(n, c, h, w))
        return dx


def _global_avgpool(pool, x):
    if AVGPOOL_FN and x.is_cuda and x.dim() == 4 and isinstance(pool, nn.AdaptiveAvgPool2d) and pool.output_size in (1, (1, 1)):
        return _GlobalAvgPoolFn.apply(x)
    return torch.flatten(pool(x), 1)


def conv_bn_pool(conv, bn, pool, x):
    """The stem: pool(relu(bn(conv(x)))).  Train mode at fp32 on liblecone's kernels: the BatchNorm apply, the ReLU and the pooling are ONE launch over the
    convolution's output (ops.BNReluPoolFn: the normalised 112 x 112 activation and, in backward, the pooling's input gradient never exist in memory)."""
    if (bn.training and BatchNormAct2d.fused_enabled and isinstance(pool, MaxPool3x3s2) and isinstance(bn, BatchNormAct2d) and bn.fuse_relu
            and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled()):
        y = conv(x)
        ops = _ops()
        if ops.stem_pool_supported(y, bn):
            ov = ops.overlap()
            sink = (bn.weight, bn.bias, ov.reducer) if (ov is not None and ov.enabled and ov.arena is not None and not ov.accumulate) else None
            # (two handles on p: the first block's conv path and its identity / downsample path -- the two gradients meet inside the fused backward)
            return ops.BNReluPoolFn.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, sink, True)
        return pool(bn(y))
    return pool(conv_bn(conv, bn, x))


class _OverlapConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, conv):
        ctx.fc = _ops().fusion()                  # the owning backbone's fusion records (ops.FusionContext); backward restores it
        w16 = _ops().overlap().weight_lp(conv, x.dtype)
        nhwc = x.is_contiguous(memory_format=torch.channels_last)
        if nhwc and w16.dim() == 4 and not w16.is_contiguous(memory_format=torch.channels_last):
            w16 = w16.contiguous(memory_format=torch.channels_last)
        ctx.f32 = (MFMA_F32 and nhwc and x.dtype == torch.float32 and w16.dtype == torch.float32 and _f32_conv_ok(conv)
                   and w16.is_contiguous(memory_format=torch.channels_last) and _f32_conv_fits(conv, x))
        if ctx.f32:
            # the reference's precision: liblecone's f32-MFMA implicit GEMM (every layer shape, stride and direction), with the
            # BatchNorm statistics of the output in its epilogue
            ctx.stem = conv.in_channels == 3
            if ctx.stem:                          # the kernels want >= 4 input channels: a zero 4th channel on both operands
                x, w16 = _pad_c4(x), _pad_c4(w16)
            ctx.planes = None
            if F32_MODE == 'x3':
                planes = getattr(conv, '_x3_planes', None)
                if planes is None or planes.shape != tuple(w16.shape) or planes.fwd.device != w16.device:
                    planes = conv._x3_planes = _ops().X3Planes(w16.shape, w16.device)
                ctx.planes = planes.update(w16)                 # the weights move every optimizer step: split again (tens of KB to a few MB)
                y = _ops().conv_f32x3_fwd(x, planes, conv.stride[0], conv.padding[0], want_stats=True)
            elif (ctx.stem and conv.kernel_size == (7, 7) and conv.stride == (2, 2) and conv.padding == (3, 3) and conv.out_channels == 64
                  and _ops().conv_f32_stem_supported(x)):
                y = _ops().conv_f32_stem_fwd(x, w16, want_stats=True)       # the stem's own kernel: the 7 input rows of an output row staged once, only the 3 real channels multiplied
            else:
                y = _ops().conv_f32_fwd(x, w16, conv.stride[0], conv.padding[0], want_stats=True)
                if _ops().LAZY_BN_PASS2_F32 and getattr(conv, 'bn_exclusive', False) and _is_pointwise(conv) and not ctx.stem and conv.out_channels % 32 == 0:
                    lz_ok = ctx.fc.lazy_ok                      # the BatchNorm behind this layer may leave pass 2 of its backward to
                    if len(lz_ok) > 64:                         # this layer's operand loads (lec_conv_f32_dgrad_fused / _wgrad_fused)
                        lz_ok.clear()
                    lz_ok[y.data_ptr()] = conv.in_channels
            ctx.save_for_backward(x, w16); ctx.conv = conv
            ctx.pointwise = ctx.own = ctx.own3 = ctx.gen = False
            return y
        ctx.pointwise = GEMM_1X1 and nhwc and _is_pointwise(conv)
        ctx.own = ctx.pointwise and MFMA_1X1 and x.dtype == torch.bfloat16 and CONV_BF16 != '2'
        ctx.own3 = (MFMA_3X3 and nhwc and x.dtype == torch.bfloat16 and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
                    and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1 and CONV_BF16 != '2'
                    and conv.in_channels == conv.out_channels and conv.in_channels in (64, 128)
                    and (x.shape[0] * x.shape[2] * x.shape[3]) % 32 == 0)
        if ctx.own3 and conv.in_channels == 128 and MFMA_3X3_C128 == '0':
            ctx.own3 = False
        # the bf16 implicit-GEMM family: every layer shape (the stem through zero channels 3..7)
        ctx.gen = (CONV_BF16 != '0' and nhwc and x.dtype == torch.bfloat16 and w16.dtype == torch.bfloat16 and w16.dim() == 4
                   and w16.is_contiguous(memory_format=torch.channels_last) and _ops().conv_bf16_supported(conv) and _bf16_conv_fits(conv, x))
        ctx.stem = False
        if ctx.own3 and (conv.in_channels == 64 or MFMA_3X3_C128 == '1'):                                            # layer1's conv2: liblecone's MFMA kernel, statistics in the epilogue
            y = _ops().conv3x3_c64(x, w16, want_stats=True)
        elif ctx.own and _ops().conv1x1_supported(conv.in_channels, conv.out_channels, x.shape[0] * x.shape[2] * x.shape[3]):
            n, _, h, wd = x.shape
            ops = _ops()
            # (only when the fused BatchNorm op is the one that will consume the unwritten tensor: its stock-torch fallback would read
            # uninitialised memory)
            if (getattr(conv, 'defer_bn', False) and ops.DEFER_BN_APPLY and BatchNormAct2d.fused_enabled and x.dtype == torch.bfloat16
                    and ops.conv1x1_bnapply_supported(conv.in_channels, conv.out_channels, n * h * wd)):
                # conv3 -> bn3: statistics only for now; the BatchNorm runs the product again with its apply pass in the epilogue
                y = _from_rows(ops.conv1x1_stats_rows(_rows(x), w16.reshape(conv.out_channels, conv.in_channels)), n, h, wd)
            else:
                y = _from_rows(ops.conv1x1_rows(_rows(x), w16.reshape(conv.out_channels, conv.in_channels), want_stats=True), n, h, wd)
        elif ctx.gen:
            ctx.stem = conv.in_channels == 3
            if ctx.stem:
                x, w16 = _pad_c8(x), _pad_c8(w16)
            if (ctx.stem and conv.kernel_size == (7, 7) and conv.stride == (2, 2) and conv.padding == (3, 3) and conv.out_channels == 64
                    and _ops().conv_bf16_stem_supported(x)):
                y = _ops().conv_bf16_stem_fwd(x, w16, want_stats=True)      # the stem's own kernel: the 7 input rows of an output row staged once, 7 K steps
            else:
                y = _ops().conv_bf16_fwd(x, w16, conv.stride[0], conv.padding[0], want_stats=True)
        elif ctx.pointwise and conv.in_channels >= GEMM_FWD_MIN_CIN:
            n, _, h, wd = x.shape
            _lib_launch('fwd')
            y = _from_rows(torch.mm(_rows(x), w16.reshape(conv.out_channels, conv.in_channels).t()), n, h, wd)
        else:
            _lib_launch('fwd')
            y = torch.ops.aten.convolution(x, w16, None, conv.stride, conv.padding, conv.dilation, False, [0, 0], conv.groups)
        ctx.save_for_backward(x, w16); ctx.conv = conv
        return y

    @staticmethod
    def backward(ctx, gy):
        with _ops().use_fusion(ctx.fc):
            return _OverlapConvFn._backward(ctx, gy)

    @staticmethod
    def _backward(ctx, gy):
        x, w16 = ctx.saved_tensors; conv = ctx.conv
        gx = None
        wgrad_done = False
        lz = _ops().fusion().lazy_dx.pop(gy.data_ptr(), None)
        xf = None
        if lz is not None and 'coef' in lz:
            # fp32: gy is UNWRITTEN; the BatchNorm behind this layer left (g, its input, the coefficient vectors): both gradient kernels
            # form dy on their operand load.  Anything that cannot (the split mode, a layer this is not meant for) materialises it first.
            if ctx.f32 and ctx.planes is None and not ctx.stem and _is_pointwise(conv) and lz['g'].shape == gy.shape:
                xf = (lz['x'], lz['coef']); gy = lz['g']
            else:
                _ops().bn_bwd_apply_lazy(lz, gy)
            lz = None
        if lz is not None:
            # gy is UNWRITTEN: the BatchNorm behind this layer left pass 2 of its backward to us.  With the flat arena's fp32 gradient
            # slot at hand the weight-gradient kernel does it on the way (and writes gy for the data gradient below); otherwise
            # pass 2 runs on its own first.
            ov = _ops().overlap()
            w = conv.weight
            if (ov is not None and ov.arena is not None and w.grad is not None and w.grad.dtype == torch.float32 and _is_pointwise(conv)
                    and x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last)
                    and gy.is_contiguous(memory_format=torch.channels_last)):
                _ops().conv1x1_wgrad_bnapply_rows(lz, _rows(x), gy, w.grad.view(conv.out_channels, conv.in_channels))
                if ov.reducer is not None:
                    ov.reducer.mark_ready(w)
                wgrad_done = True
            else:
                _ops().bn_bwd_apply_lazy(lz, gy)
        if ctx.f32 and not gy.is_contiguous(memory_format=torch.channels_last):
            gy = gy.contiguous(memory_format=torch.channels_last)
        if ctx.needs_input_grad[0]:
            nhwc_g = gy.is_contiguous(memory_format=torch.channels_last)
            if ctx.f32:
                ops = _ops()
                if ops.overlap() is not None:
                    ops.overlap().wait_matrix()
                rec = ops.fusion().forks.get(x.data_ptr()) if (ops.FOLD_BN_BWD_F32 and ctx.planes is None and not ctx.stem) else None
                if rec is not None and not (conv.stride[0] == 1 and conv.out_channels % 32 == 0 and x.shape[1] % 8 == 0 and x.shape[1] >= ops.FOLD_F32_MIN_CHANNELS
                                            and rec['x'].dtype == torch.float32
                                            and rec['x'].shape == x.shape and (rec.get('single') or rec['dres'] is not None)
                                            and (rec['dres'] is None or (rec['dres'].dtype == torch.float32 and rec['dres'].shape == x.shape
                                                                         and rec['dres'].is_contiguous(memory_format=torch.channels_last)))):
                    rec = None
                if ctx.planes is not None and not ctx.stem:
                    gx = ops.conv_f32x3_dgrad(gy, ctx.planes, x.shape, conv.stride[0], conv.padding[0])
                elif xf is not None or rec is not None:
                    # pass 2 of the BatchNorm behind this layer on the operand load, pass 1 of the one in front of it in the epilogue
                    gx = ops.conv_f32_dgrad_fused(gy, w16, x.shape, conv.stride[0], conv.padding[0], xf=xf, fold=rec)
                else:
                    gx = ops.conv_f32_dgrad(gy, w16, x.shape, conv.stride[0], conv.padding[0])
                if ctx.stem:
                    gx = gx[:, :3]
            elif ctx.own3 and nhwc_g:                             # dX = conv(dY, W flipped and transposed): the same kernel,
                gx = _ops().conv3x3_c64(gy, w16, w_transposed=True)   # which flips / transposes the weight as it loads it
            elif (ctx.own and nhwc_g and not (FAMILY_DGRAD_SMALL_MAPS and ctx.gen and gy.shape[2] * gy.shape[3] <= 784 and conv.out_channels % 64 == 0 and min(conv.in_channels, conv.out_channels) >= 128)
                  and _ops().conv1x1_supported(conv.out_channels, conv.in_channels, gy.shape[0] * gy.shape[2] * gy.shape[3])):
                n, _, h, wd = gy.shape                          # dX = dY * W: the same kernel, W transposed as it is loaded
                ops = _ops()
                fork = ops.fusion().forks.get(x.data_ptr()) if ops.FOLD_BN_BWD else None
                if (fork is not None and fork['dres'] is not None and fork['x'].shape == x.shape and fork['dres'].shape == x.shape
                        and fork['dres'].dtype == torch.bfloat16 and fork['dres'].is_contiguous(memory_format=torch.channels_last)
                        and ops.conv1x1_dgrad_bnfold_supported(conv.out_channels, conv.in_channels, n * h * wd)):
                    # x is a forked block output and the identity branch's gradient is already there: pass 1 of that
                    # block's BatchNorm backward runs in this kernel's epilogue (BNActFn.backward recognises the result)
                    gx = _from_rows(ops.conv1x1_dgrad_bnfold_rows(_rows(gy), w16.reshape(conv.out_channels, conv.in_channels), fork), n, h, wd)
                else:
                    gx = _from_rows(ops.conv1x1_rows(_rows(gy), w16.reshape(conv.out_channels, conv.in_channels), w_transposed=True), n, h, wd)
            elif ctx.gen and nhwc_g and not ctx.stem and conv.out_channels % 64 == 0 and gy.dtype == torch.bfloat16:
                ops = _ops()
                # pass 1 of the backward of the BatchNorm whose output this layer consumed, in the epilogue (stride-1 layers): a block output with
                # the other consumer's gradient already there, or an output with this one consumer
                rec = ops.fusion().forks.get(x.data_ptr()) if (ops.FOLD_BN_BWD and conv.stride[0] == 1) else None
                if rec is not None and not (rec['x'].dtype == torch.bfloat16 and rec['x'].shape == x.shape and (rec.get('single') or rec['dres'] is not None)
                                            and (rec['dres'] is None or (rec['dres'].dtype == torch.bfloat16 and rec['dres'].shape == x.shape
                                                                         and rec['dres'].is_contiguous(memory_format=torch.channels_last)))):
                    rec = None
                gx = ops.conv_bf16_dgrad(gy, ops.overlap().weight_lp_t(conv, w16), x.shape, conv.stride[0], conv.padding[0], fold=rec)
            elif (ctx.pointwise and conv.in_channels >= GEMM_DGRAD_MIN_CIN and nhwc_g):
                n, _, h, wd = gy.shape
                _lib_launch('dgrad')
                gx = _from_rows(torch.mm(_rows(gy), w16.reshape(conv.out_channels, conv.in_channels)), n, h, wd)
            else:
                _lib_launch('dgrad')
                gx = torch.ops.aten.convolution_backward(gy, x, w16, None, conv.stride, conv.padding, conv.dilation, False,
                                                         [0, 0], conv.groups, [True, False, False])[0]
        fc = _ops().fusion()
        if gx is not None and x.data_ptr() in fc.forks:         # x is a forked block output and this layer one of its two consumers
            rec = fc.forks[x.data_ptr()]
            if rec['dres'] is None and x.data_ptr() not in fc.folded and gx.data_ptr() not in fc.folded:
                rec['dres'] = gx                                # (the other consumer's data gradient may fold it into its epilogue)
        if not wgrad_done:
            _ops().overlap().submit(gy, x, w16, conv, xf=xf)
        if getattr(conv, 'is_stem', False):
            fc.in_flight = False                                # the stem's backward is the last node of the backbone's: the context is free again
        return gx, None, None


class Conv2d(nn.Conv2d):
    """nn.Conv2d whose weight gradient can be computed on the side stream of the pass's WgradOverlap (`ops.overlap()`; GPU, low-precision
    activations, training).  Same parameters and state-dict keys as nn.Conv2d."""

    def forward(self, x):
        ov = _ops().overlap()
        # (also under torch.no_grad(): the forward-only first pass of a chunked step must run the SAME kernels as the pass that is
        # differentiated -- the loss gradient is evaluated at the first pass's outputs)
        if (ov is not None and ov.enabled and x.is_cuda and self.training and self.bias is None
                and x.dtype in (torch.bfloat16, torch.float16, torch.float32) and self.weight.requires_grad):
            return _OverlapConvFn.apply(x, self.weight, self)
        if _inference_f32(self, x):
            # no gradient will be asked for (the evaluation phases embed images under torch.no_grad()): liblecone's fp32 kernel directly, with
            # the statistics of a train-mode BatchNorm behind it in its epilogue (the reference embeds 'train'-phase images in train mode)
            w = self.weight if self.weight.is_contiguous(memory_format=torch.channels_last) else self.weight.contiguous(memory_format=torch.channels_last)
            if self.in_channels == 3:
                x, w = _pad_c4(x), _pad_c4(w)
                if (self.kernel_size == (7, 7) and self.stride == (2, 2) and self.padding == (3, 3) and self.out_channels == 64 and _ops().conv_f32_stem_supported(x)):
                    return _ops().conv_f32_stem_fwd(x, w, want_stats=self.training)         # the stem's own kernel (evaluation forwards too)
            return _ops().conv_f32_fwd(x, w, self.stride[0], self.padding[0], want_stats=self.training)
        if x.is_cuda:
            _lib_launch('module')                               # nn.Conv2d's own forward (and autograd's backward behind it): MIOpen
            if os.environ.get('LEC_TRACE_LIBCONV'):
                import traceback; traceback.print_stack(limit=14)
        return super().forward(x)


class MaxPool3x3s2(nn.MaxPool2d):
    """The stem's MaxPool2d(3, stride=2, padding=1): on the MI355X with NHWC bf16 input it runs liblecone's kernel
    (one-byte argmax instead of int64 indices, gather backward); anything else takes the stock op."""

    def __init__(self):
        super().__init__(kernel_size=3, stride=2, padding=1)

    def forward(self, x):
        if (BatchNormAct2d.fused_enabled and x.is_cuda and x.dtype in (torch.bfloat16, torch.float32) and x.dim() == 4 and x.shape[1] % 8 == 0
                and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and x.is_contiguous(memory_format=torch.channels_last)):
            from . import ops
            return ops.MaxPool3x3s2Fn.apply(x)
        return super().forward(x)


def conv3x3(cin, cout, stride=1):
    return Conv2d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=False)


def conv1x1(cin, cout, stride=1):
    return Conv2d(cin, cout, kernel_size=1, stride=stride, bias=False)


def _downsample(seq, x):
    """A block's downsample branch (conv1x1 + BatchNorm as nn.Sequential, torchvision's state-dict keys) through conv_bn."""
    if isinstance(seq, nn.Sequential) and len(seq) == 2 and isinstance(seq[1], BatchNormAct2d):
        return conv_bn(seq[0], seq[1], x)
    return seq(x)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv3x3(cin, planes, stride); self.bn1 = BatchNormAct2d(planes, relu=True)
        self.conv2 = conv3x3(planes, planes); self.bn2 = BatchNormAct2d(planes, relu=True)     # relu(bn2(.) + identity)
        self.conv1.bn_exclusive = self.conv2.bn_exclusive = True    # forward() hands each convolution's output to its BatchNorm and to nothing else
        self.downsample = downsample

    def forward(self, x, fork=False):
        xa, xb = x if isinstance(x, tuple) else (x, x)          # two handles on the block input: conv path / identity path
        idt = xb if self.downsample is None else _downsample(self.downsample, xb)
        out = conv_bn(self.conv1, self.bn1, xa)
        return conv_bn(self.conv2, self.bn2, out, idt, fork)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv1x1(cin, planes); self.bn1 = BatchNormAct2d(planes, relu=True)
        self.conv2 = conv3x3(planes, planes, stride); self.bn2 = BatchNormAct2d(planes, relu=True)   # stride on the 3x3 (v1.5)
        self.conv3 = conv1x1(planes, planes * 4); self.bn3 = BatchNormAct2d(planes * 4, relu=True)  # relu(bn3(.) + identity)
        self.conv3.defer_bn = True                             # forward() hands conv3's output to bn3 and to nothing else (FusionContext.deferred)
        self.conv1.bn_exclusive = self.conv2.bn_exclusive = self.conv3.bn_exclusive = True   # ... and so for every convolution of the block:
                                                               # its BatchNorm may leave pass 2 of the backward to the convolution's operand loads
        self.downsample = downsample

    def forward(self, x, fork=False):
        xa, xb = x if isinstance(x, tuple) else (x, x)          # two handles on the block input: conv path / identity path
        out = conv_bn(self.conv1, self.bn1, xa)
        # the downsample branch is built AFTER conv1 / bn1: autograd runs later-built nodes first, so in backward the branch's
        # gradient into the block input exists before conv1's data gradient runs and can be folded into it (FusionContext.forks)
        idt = xb if self.downsample is None else _downsample(self.downsample, xb)
        out = conv_bn(self.conv2, self.bn2, out)
        return conv_bn(self.conv3, self.bn3, out, idt, fork)


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000, width=64):
        """width: channels of the stem and of layer1 (torchvision: 64; a power of two >= 8 keeps every layer on liblecone's kernels --
        narrow instances are the stand-in backbones of the parity fixtures)."""
        super().__init__()
        self.inplanes = width
        self.conv1 = Conv2d(3, width, kernel_size=7, stride=2, padding=3, bias=False)
        self.conv1.is_stem = True                               # its backward is the last node of the backbone's backward (FusionContext.in_flight)
        self.bn1 = BatchNormAct2d(width, relu=True)
        self.maxpool = MaxPool3x3s2()
        self.layer1 = self._make_layer(block, width, layers[0])
        self.layer2 = self._make_layer(block, 2 * width, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 4 * width, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 8 * width, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(8 * width * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1); nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(conv1x1(self.inplanes, planes * block.expansion, stride),
                                       BatchNormAct2d(planes * block.expansion, relu=False))
            downsample[0].bn_exclusive = True
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    # Settings of THIS backbone's forward / backward passes, copied into the FusionContext of every forward (ops.FusionContext) -- per object,
    # not per process: two trainers / engines in one process, or driven from two threads, do not see each other's.
    wgrad_overlap = None            # the WgradOverlap its convolutions / BatchNorms report to (None: the process default WgradOverlap.instance)
    bn_grad_accumulate = False      # BatchNorm backward ADDS d gamma / d beta (a step of several backward passes zeroes the slots once)
    conv_schedule = -1              # LEC_SCHEDULE_* of the fp32 forward / data-gradient launches (-1: the library default)
    step_timers = None              # {'bn': [], 'conv': []}: per-launch HIP events while the owner profiles its step (bench.py)
    max_forwards_in_flight = 64     # training forwards of one backbone that may wait for their backward at once (each holds a fusion context + BatchNorm workspace)

    def forward(self, x, pooled_only=False, pass_order=None):
        """pooled_only: stop after the global average pooling ([n, 512 * expansion] features); `self.fc` is then the caller's to apply --
        the engine's concurrent half-batch passes pool per pass and run the fully connected layer ONCE over all rows.
        pass_order = (dict shared by the passes of a step, index of this pass): see ops.BNActFn.forward."""
        if x.is_cuda and torch.is_autocast_enabled() and x.dtype == torch.float32:
            x = x.to(torch.get_autocast_dtype('cuda'))          # the stem conv sees low-precision input like every other layer
        if not x.is_cuda:
            return self._forward(x, pooled_only)
        # The fused paths' hand-off records and BatchNorm workspace belong to ONE forward / backward pair of THIS instance
        # (ops.FusionContext).  A second backbone in the process has its own; so has a second forward of this one that starts before the
        # first one's backward has run (the engine's concurrent half-batch passes): the pool below hands out a context that is not in use.
        pool = self.__dict__.setdefault('_fusions', [])
        track = self.training and torch.is_grad_enabled()
        fc = next((c for c in pool if not c.busy()), None)
        if fc is None:
            # every context belongs to a forward whose backward can still come: the pool GROWS (reference_exact_batches keeps four forwards in flight; a
            # fifth used to take over the oldest one's BatchNorm workspace and records silently).  Contexts whose graph cannot be tracked (stock autograd
            # convolutions: no stem node of ours) are the only ones ever recycled, oldest first, and only once the pool is large.
            if len(pool) >= self.max_forwards_in_flight:
                old = next((c for c in pool if c.graph_ref is None), None)
                if old is None:
                    raise RuntimeError('%d forwards of this backbone are waiting for their backward (ResNet.max_forwards_in_flight = %d; each holds a fusion context and a '
                                       'BatchNorm workspace): run backward, drop the outputs, use torch.no_grad() for forwards that need no gradient, or raise the limit' % (len(pool), self.max_forwards_in_flight))
                pool.remove(old); pool.append(old); fc = old
            else:
                fc = _ops().FusionContext(); pool.append(fc)
        fc.reset()
        fc.overlap, fc.accumulate, fc.schedule, fc.pass_order = self.wgrad_overlap, bool(self.bn_grad_accumulate), int(self.conv_schedule), pass_order
        tm = self.step_timers
        fc.bn_timer, fc.conv_timer = (tm['bn'], tm['conv']) if tm is not None else (None, None)
        self.__dict__['_fusion'] = fc
        with _ops().use_fusion(fc):
            y = self._forward(x, pooled_only)
        if not track:
            fc.reset()                                          # no backward will come for these records
        else:
            fc.in_flight = True                                 # until the stem's backward has run (or the forward's graph is freed: FusionContext.busy)
        return y

    def __deepcopy__(self, memo):
        """The fusion contexts (hand-off records of forwards in flight, BatchNorm workspaces) are launch resources of THIS object, not state:
        a copy starts with none and creates its own."""
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k not in ('_fusions', '_fusion', 'wgrad_overlap', 'step_timers'):    # (streams, events: launch resources of the original's owner)
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    @property
    def fusion(self):
        """The FusionContext of the most recent forward."""
        fc = self.__dict__.get('_fusion')
        if fc is None:
            fc = _ops().FusionContext()
            self.__dict__.setdefault('_fusions', []).append(fc); self.__dict__['_fusion'] = fc
        return fc

    def _forward(self, x, pooled_only=False):
        x = conv_bn_pool(self.conv1, self.bn1, self.maxpool, x)
        blocks = [b for layer in (self.layer1, self.layer2, self.layer3, self.layer4) for b in layer]
        for i, b in enumerate(blocks):
            x = b(x, fork=i + 1 < len(blocks))                  # every block output but the last feeds two branches
        x = _global_avgpool(self.avgpool, x)
        return x if pooled_only else self.fc(x)


def resnet18(num_classes=1000):
    return ResNet(BasicBlock, [2, 2, 2, 2], num_classes)


def resnet50(num_classes=1000):
    return ResNet(Bottleneck, [3, 4, 6, 3], num_classes)


# analytic work per image (2 flops / MAC, backward = 2x forward): SURVEY.md 8(d)
GFLOP_FWD_BWD_PER_IMAGE = {('resnet18', 224): 10.881, ('resnet50', 224): 24.523, ('resnet18', 32): 0.222 * 3 / 1.0, ('resnet50', 32): 0.501 * 3}


def conv_macs(model, hw):
    """MACs of one forward pass at hw x hw (convs + fc), by shape walk -- used by bench.py for the MFMA roofline."""
    macs = 0
    hooks = []

    def conv_hook(m, inp, out):
        nonlocal macs
        macs += out.shape[1] * out.shape[2] * out.shape[3] * (m.in_channels // m.groups) * m.kernel_size[0] * m.kernel_size[1]

    def fc_hook(m, inp, out):
        nonlocal macs
        macs += m.in_features * m.out_features

    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            hooks.append(m.register_forward_hook(conv_hook))
        elif isinstance(m, nn.Linear):
            hooks.append(m.register_forward_hook(fc_hook))
    global _WALK_MODULES
    was = model.training
    model.eval()
    _WALK_MODULES = True                                        # the inference forward calls liblecone's fused conv + affine kernel directly (conv_bn): walk the MODULES
    try:                                                        # instead -- each Conv2d.forward still runs liblecone's kernel on the GPU (no library convolution for a FLOP count)
        with torch.no_grad():
            dev = next(model.parameters()).device
            model(torch.zeros(1, 3, hw, hw, device=dev, dtype=next(model.parameters()).dtype).contiguous(memory_format=torch.channels_last))
    finally:
        _WALK_MODULES = False
        model.train(was)
    for h in hooks:
        h.remove()
    return macs
