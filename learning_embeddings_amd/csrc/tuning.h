// Launch-policy knobs of liblecone.so, resolved ONCE (first use, thread-safe) into one immutable struct: no launch path reads the environment.
// Every field has the default the round's measurements settled on; the LEC_* variable named beside it overrides it for A/B runs
// (tools/*.sh set them before the process starts -- changing the environment afterwards has no effect, by design).
#pragma once

namespace lec {
struct Tuning {
  int bn_apply_blocks;        // LEC_BN_BLOCKS          blocks of a BatchNorm apply pass (1536: leaves wave slots for the other pass's dependent launches)
  int cf_sk;                  // LEC_CF_SK              balanced (stream-K) forward / data gradient: 0 never, 1 where the last round of slots is under cf_sk_fill, 2 always
  double cf_sk_fill;          // LEC_CF_SK_FILL
  int cf_sk_min_chunks;       // LEC_CF_SK_MIN_CHUNKS
  int cf_stem;                // LEC_CF_STEM            the fp32 7x7 / stride-2 stem (forward and weight gradient) on its own LDS-patch kernels (1) or on the generic kernels (0)
  int cf_sk_wgs;              // LEC_CF_SK_WGS          workgroups of a balanced launch (512: every resident slot; 256: one per CU -- two passes' launches side by side)
  int cf_xcd;                 // LEC_CF_XCD             XCD-contiguous tile runs (measured: no gain; off)
  int cf_lds_pad;             // LEC_CF_LDS_PAD         extra LDS bytes per workgroup of the forward / data-gradient kernels (experiments: residency)
  int dgrad_classes;          // LEC_DGRAD_CLASSES      parity classes of a strided data gradient as ONE launch
  int wg_dense_tile;          // LEC_WGRAD_DENSE_TILE   tile rule of dense 1x1 weight gradients
  int wg_bm128;               // LEC_WGRAD_BM128
  int wg_shift;               // LEC_WGRAD_SHIFT        shifted-dense weight-gradient form
  int wg_shift64;             // LEC_WGRAD_SHIFT64
  int wg_items;               // LEC_WGRAD_ITEMS        work items (tiles x K split) a weight gradient aims for
  int wg_split_floor;         // LEC_WGRAD_SPLIT_FLOOR
  int wg_lds_pad;             // LEC_WGRAD_LDS_PAD
  int wg_wgs;                 // LEC_WGRAD_WGS          cap on the weight-gradient grid
  int wg_smask;               // LEC_WGRAD_SMASK
  int bf_dma;                 // LEC_BF16_DMA           bf16 forward / data gradient: operands by LDS-DMA through a ring of whole-line stages (1) or staged through registers (0)
  int bf_stem;                // LEC_BF16_STEM          the 7x7 / stride-2 stem forward on its own LDS-patch kernel (1) or on the generic per-piece-tap kernel (0: lec_conv_bf16_stem_supported says no)
  int bf_tile;                // LEC_BF16_TILE          bf16 forward / data gradient tile policy: 1 = 256 x 256 (long K, >= 160 tiles) and 128 x 256 (1x1, 256 destination channels) where they pay, 0 = 128 x 128 always, 2 / 3 = 128 x 256 / 256 x 256 wherever they fit
  int x3_wgs;                 // LEC_X3_WGS             workgroups of the split-product (x3) kernels
  int x3_force_narrow;        // LEC_X3_FORCE_NARROW
  int x3_chain;               // LEC_X3_CHAIN           longest fp32 accumulation chain of an x3 weight gradient, in chunks
  int c3_strip;               // LEC_C3_STRIP           bf16 3x3: the strip kernel instead of the LDS-halo one
  int jl_T, jl_EPL, jl_iters; // LEC_JOINT_GEOM="T,EPL[,iters]"  geometry override of the fused loss (sweeps)
  int jl_stage;               // LEC_JOINT_STAGE        lane-per-pair rows through LDS
  int jl_fixed_point;         // LEC_JOINT_FIXED_POINT  loss hand-off as one integer atomic per block where the loss is bounded at launch (1; 0: always the ticket form)
  int jl_wpb;                 // LEC_JOINT_WPB          waves per block of the fused loss (0: the geometry's rule; 1..8)
};
const Tuning& tuning();       // abi.cpp
}  // namespace lec
