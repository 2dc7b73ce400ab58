"""ctypes binding of liblecone.so (include/lecone.h).  There is NO fallback: if the library is missing or a call fails,
this module raises.  The product never routes through oracle/ or any CPU re-implementation."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('LEC_LIB_PATH') or os.path.join(_HERE, 'liblecone.so')     # LEC_LIB_PATH: A/B builds of the same ABI
ABI_VERSION = 33

OK, E_ARG, E_HIP, E_EMPTY, E_STATE = 0, -1, -2, -3, -4
ENERGY_HYP_CONE, ENERGY_ORDER, ENERGY_EUC_CONE = 0, 1, 2
LABEL_RAW, LABEL_HYP, LABEL_SOFTCLIP_K = 0, 1, 2
IMAGE_RAW, IMAGE_SOFTCLIP, IMAGE_SOFTCLIP_K = 0, 1, 2
SCHEDULE_DEFAULT, SCHEDULE_TILE_WALK, SCHEDULE_AUTO, SCHEDULE_BALANCED = -1, 0, 1, 2


class LeconeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('liblecone error %d: %s' % (code, msg))
        self.code = code


class EmptyCandidates(LeconeError, IndexError):
    """The reference's random.choice raises IndexError on an empty candidate list (oe_h.py:901)."""


def _load():
    # torch bundles its own HIP runtime (torch/lib/libamdhip64.so).  Import it FIRST so that liblecone.so binds to the
    # runtime that owns torch's device context and streams; loading /opt/rocm's copy beside it gives a second runtime
    # with no device ("no ROCm-capable device is detected" at the first launch).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'liblecone.so not found at %s -- build it first: `python -c "import __graft_entry__ as g; g.build()"` '
            'or `make -C learning_embeddings_amd/csrc`.  There is no CPU fallback.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.lec_last_error.restype = C.c_char_p
    lib.lec_abi_version.restype = C.c_int
    if lib.lec_abi_version() != ABI_VERSION:
        raise ImportError('liblecone.so ABI %d != expected %d: rebuild' % (lib.lec_abi_version(), ABI_VERSION))
    p, i32, i64, f32, u64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64
    sig = {
        'lec_loss_workspace_bytes': (i64, [i32, i32, i32]),
        'lec_pair_energy_fwd': (i32, [i32, p, i64, p, i64, i64, i32, f32, p, p]),
        'lec_pair_energy_bwd': (i32, [i32, p, i64, p, i64, p, i64, i32, f32, p, p, i64, p]),
        'lec_pair_energy_matrix': (i32, [i32, p, i64, i64, p, i64, i64, i32, f32, p, i64, p]),
        'lec_level_topk': (i32, [i32, p, i64, i64, p, i64, i64, i32, p, i32, i32, f32, p, p, p]),
        'lec_joint_loss_fwd_bwd': (i32, [i32, i32, i32, p, i64, i32, p, i64, i32, p, p, p, p, i32, i32, i32, f32, f32,
                                         p, p, p, p, p, p, i64, p]),
        'lec_joint_loss_fwd_bwd_f16': (i32, [i32, i32, i32, p, i64, i32, p, i64, i32, p, p, p, p, i32, i32, i32, f32, f32,
                                             p, p, p, p, p, p, i64, p]),
        'lec_joint_loss_fwd_bwd_window': (i32, [i32, i32, i32, p, p, i64, i32, p, i64, i32, p, p, p, p, i32, i32, i32, f32, f32, i32, i32, i32, p,
                                                p, p, p, p, p, p, i64, p]),
        'lec_table_step_adam_f16': (i32, [p, p, p, p, i64, i32, i32, f32, f32, f32, f32, i32, f32, i32, i32, p, p]),
        'lec_label_project_fwd': (i32, [i32, p, i64, i32, p, i64, i32, f32, p, i64, p]),
        'lec_label_project_bwd': (i32, [i32, p, i64, i32, p, i64, i32, f32, p, i64, p, p]),
        'lec_image_softclip_fwd': (i32, [i32, p, i64, i64, i32, f32, p, i64, p]),
        'lec_image_softclip_bwd': (i32, [i32, p, i64, p, i64, i64, i32, f32, p, i64, p]),
        'lec_table_step_adam': (i32, [p, p, p, p, i64, i32, i32, f32, f32, f32, f32, i32, f32, i32, i32, p]),
        'lec_table_step_rsgd': (i32, [p, p, i64, i32, i32, f32, f32, p]),
        'lec_adam_flat': (i32, [p, p, p, p, i64, f32, f32, f32, f32, i32, f32, p, p]),
        'lec_sampler_create': (i32, [C.POINTER(p), p, i32, p, i64, p, p, i64, i32, i32, u64]),
        'lec_sampler_destroy': (None, [p]),
        'lec_sampler_seed': (i32, [p, u64]),
        'lec_sampler_set_levels_to_hide': (i32, [p, p, i32]),
        'lec_sampler_visible_slots': (i32, [p, p, p]),
        'lec_sampler_draw': (i32, [p, i32, i32, i32, p]),
        'lec_sampler_draw_batch': (i32, [p, p, p, i32, i32, p]),
        'lec_sampler_next_u32': (i32, [p, p]),
        'lec_sampler_tc_edges': (i64, [p]),
        'lec_sampler_tc_export': (i32, [p, p, p]),
        'lec_dp_unique_id': (i32, [p]),
        'lec_dp_init': (i32, [C.POINTER(p), i32, i32, p, i32]),
        'lec_dp_allreduce_sum': (i32, [p, p, i64, i32, p]),
        'lec_dp_destroy': (None, [p]),
        'lec_multilevel_ce_fwd_bwd': (i32, [p, i64, p, i32, i32, p, p, p, i32, p, p, p, i64, p]),
        'lec_bn_workspace_bytes': (i64, [i32]),
        'lec_bn_fwd': (i32, [p, p, i64, i32, p, p, f32, f32, p, p, i32, p, p, p, i32, p, p, i64, p]),
        'lec_bn_bwd': (i32, [p, p, p, p, p, i64, i32, p, p, p, p, p, p, p, i32, p, i64, i32, p]),
        'lec_conv1x1_supported': (i32, [i32, i32, i64]),
        'lec_conv1x1_fwd': (i32, [p, p, i32, i64, i32, i32, p, p, i64, p, p]),
        'lec_conv1x1_bnapply_supported': (i32, [i32, i32, i64]),
        'lec_conv1x1_stats': (i32, [p, p, i64, i32, i32, p, i64, p, p]),
        'lec_bn_workspace_coeff_offset': (i64, [i32]),
        'lec_bn_fwd_finalize': (i32, [i64, i32, p, p, f32, f32, p, p, i32, p, p, p, i64, p]),
        'lec_conv1x1_fwd_bnapply': (i32, [p, p, i64, i32, i32, p, p, p, p, p, p, p]),
        'lec_conv1x1_dgrad_bnfold_supported': (i32, [i32, i32, i64]),
        'lec_conv1x1_dgrad_bnfold': (i32, [p, p, i32, i64, i32, i32, p, p, p, p, p, p, p, i64, p, p]),
        'lec_bn_bwd_prereduced': (i32, [p, p, i64, i32, p, p, p, i32, p, p, p, p, i64, i32, p]),
        'lec_conv3x3_c64_wgrad_supported': (i32, [i32, i32, i32]),
        'lec_conv3x3_c64_wgrad': (i32, [p, p, i32, i32, i32, p, p]),
        'lec_bn_bwd_pass1': (i32, [p, p, p, p, i64, i32, p, p, p, p, p, p, i64, i32, p]),
        'lec_bn_bwd_finalize': (i32, [i64, i32, i32, p, p, p, i64, i32, p]),
        'lec_bn_bwd_apply': (i32, [p, p, i64, i32, p, p, p, p, p, i64, p]),
        'lec_conv1x1_wgrad_bnapply_supported': (i32, [i32, i32, i64]),
        'lec_conv1x1_wgrad_bnapply': (i32, [p, p, p, i64, i32, i32, p, p, p, p, p, p, p, p]),
        'lec_conv1x1_wgrad_supported': (i32, [i32, i32, i64]),
        'lec_conv1x1_wgrad': (i32, [p, p, i64, i32, i32, p, p]),
        'lec_conv3x3_c64_fwd': (i32, [p, p, i32, i32, i32, i32, p, p, i64, p, p]),
        'lec_conv3x3_c128_fwd': (i32, [p, p, i32, i32, i32, p, p, i64, p, p]),
        'lec_bn_fwd_prestat': (i32, [p, p, i64, i32, p, p, f32, f32, p, p, i32, p, p, p, i32, p, p, i64, p]),
        'lec_conv_f32_fwd': (i32, [p, p, i32, i32, i32, i32, i32, i32, i32, i32, i32, p, p, i64, p, i32, p]),
        'lec_conv_f32_stem_supported': (i32, [i32, i32, i32]),
        'lec_conv_f32_stem_fwd': (i32, [p, p, i32, i32, i32, p, p, i64, p, p]),
        'lec_conv_f32_fwd_affine': (i32, [p, p, i32, i32, i32, i32, i32, i32, i32, i32, i32, p, p, p, p, i32, i32, p]),
        'lec_conv_f32_dgrad': (i32, [p, p, i32, i32, i32, i32, i32, i32, i32, i32, i32, p, i32, p]),
        'lec_conv_f32_wgrad': (i32, [p, p, i32, i32, i32, i32, i32, i32, i32, i32, i32, p, p]),
        'lec_conv_f32_dgrad_fused': (i32, [p, p, i32, i32, i32, i32, i32, i32, i32, i32, i32, p, p, p, p, p, p, p, p, p, i64, p, i32, p]),
        'lec_conv_f32_wgrad_c3': (i32, [p, p, i32, i32, i32, i32, i32, i32, i32, i32, p, p]),
        'lec_conv_f32_wgrad_fused': (i32, [p, p, i32, i32, i32, i32, i32, i32, i32, i32, i32, p, p, p, p]),
        'lec_bn_eval_coeffs_f32': (i32, [i32, p, p, f32, p, p, p, p, p]),
        'lec_conv_f32_scratch_bytes': (i64, []),
        'lec_conv_f32_scratch': (i32, [p, p, i64]),
        'lec_bn_bwd_coeffs_f32': (i32, [i64, i32, i32, p, p, p, p, p, p, p, i64, i32, p]),
        'lec_bn_bwd_pass1_coeffs_f32': (i32, [p, p, p, p, i64, i32, p, p, p, p, p, p, p, p, i64, i32, p]),
        'lec_conv_f32x3_planes_elems': (i64, [i32, i32, i32, i32]),
        'lec_conv_f32x3_split_weights': (i32, [p, i32, i32, i32, p, p, p]),
        'lec_conv_f32x3_fwd': (i32, [p, p, i32, i32, i32, i32, i32, i32, i32, i32, i32, p, p, i64, p, p]),
        'lec_conv_f32x3_dgrad': (i32, [p, p, i32, i32, i32, i32, i32, i32, i32, i32, i32, p, p]),
        'lec_conv_f32x3_wgrad_supported': (i32, [i32, i32, i32, i32]),
        'lec_conv_f32x3_wgrad': (i32, [p, p, i32, i32, i32, i32, i32, i32, i32, i32, i32, p, p]),
        'lec_conv_bf16_supported': (i32, [i32, i32, i32, i32, i32, i32]),
        'lec_conv_bf16_wt_transpose': (i32, [p, p, i32, i32, i32, p]),
        'lec_conv_bf16_wt_transpose_flat': (i32, [p, p, p, i32, i32, p]),
        'lec_conv_bf16_fwd': (i32, [p, p, i32, i32, i32, i32, i32, i32, i32, i32, i32, p, p, i64, p, p]),
        'lec_conv_bf16_stem_supported': (i32, [i32, i32]),
        'lec_conv_bf16_stem_fwd': (i32, [p, p, i32, i32, i32, p, p, i64, p, p]),
        'lec_conv_bf16_dgrad': (i32, [p, p, i32, i32, i32, i32, i32, i32, i32, i32, i32, p, p, p, p, p, p, p, i64, p, p]),
        'lec_conv_bf16_wgrad': (i32, [p, p, i32, i32, i32, i32, i32, i32, i32, i32, i32, p, i32, p]),
        'lec_maxpool3x3s2_fwd': (i32, [p, i32, i32, i32, i32, p, p, p]),
        'lec_maxpool3x3s2_bwd': (i32, [p, p, i32, i32, i32, i32, p, p]),
        'lec_image_gather_u8': (i32, [p, i64, p, p, i32, i32, i32, i32, p, p]),
        'lec_bn_relu_maxpool_fwd_f32': (i32, [p, i32, i32, i32, i32, p, p, p, p, p]),
        'lec_bn_relu_maxpool_bwd_f32': (i32, [p, p, p, p, i32, i32, i32, i32, p, p, p, p, p, p, p, p, i64, i32, p]),
    }
    for base in ('lec_bn_fwd', 'lec_bn_bwd', 'lec_bn_bwd_pass1', 'lec_bn_bwd_apply', 'lec_bn_bwd_prereduced', 'lec_bn_fwd_prestat',
                 'lec_maxpool3x3s2_fwd', 'lec_maxpool3x3s2_bwd'):
        sig[base + '_f32'] = sig[base]            # fp32-activation twins: identical arguments
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)           # AttributeError here = header and library disagree: fail loudly
        fn.restype = res; fn.argtypes = args
    return lib, sorted(sig)


lib, EXPORTS = _load()


def check(rc):
    if rc == OK:
        return
    msg = lib.lec_last_error().decode('utf-8', 'replace')
    if rc == E_EMPTY:
        raise EmptyCandidates(rc, msg)
    raise LeconeError(rc, msg)


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dptr(t):
    """Raw device pointer of a CUDA(HIP) tensor (or None)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('liblecone kernels run on the MI355X only: got a %s tensor (there is no CPU fallback)' % t.device)
    return C.c_void_p(t.data_ptr())
