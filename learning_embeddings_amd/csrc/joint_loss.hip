// Fused joint (label, image) entailment-cone loss: forward + backward in ONE launch (lec_joint_loss_fwd_bwd).
//
// Replaces, for one training step after negative sampling (reference file:line):
//   oe_h.py:929-967   criterion.forward train branch (embedding of positives and negatives, E+, E-, hinge, sum)
//   oe_h.py:77-104    Embedder.forward + no-grad clip, for the label rows the step touches
//   oe_h.py:323-328   FeatCNN18.soft_clip for the image rows
//   loss.backward()   down to d/d table (dense, accumulated) and d/d raw CNN outputs
//
// Work decomposition (MI355X: 256 CUs x 4 SIMDs, 64-lane waves).  A "positive group" b owns the rows
// {u_b, v_b, K corrupt v', K corrupt u'} and the 1+2K pairs built from them.  A wave takes one TASK = (group b,
// chunk c of its pair list); T lanes cooperate on one pair (row elements d = t, t+T, ... live in registers, EPL per
// lane), so a wave evaluates 64/T pairs per iteration.  Row statistics (|x|^2, |y|^2, <x,y>, |x-y|^2) and the
// projection norms are reduced with xor-butterflies inside the T-lane group; the gradient of u_b / v_b is accumulated
// in registers across the chunk, reduced across the wave's pair slots by butterflies, and leaves the wave as ONE
// atomic row-add per task.  Corrupt rows get one atomic row-add per live pair (dead hinges add nothing).  The loss
// is reduced wave -> block -> grid deterministically (lec_common.h block_publish_and_finalize).
//
// Roofline: HBM / L2 bound gather + scatter; algorithmic bytes per positive (fwd+bwd, rows de-duplicated inside a
// group) = (2+2K)(2*D*4 + 4*D + 4) + (1+2K)*8   (SURVEY.md 8d).
#include <climits>
#include <cstdio>
#include <cstdlib>
#include "lec_common.h"
#include "tuning.h"

namespace lec {

struct JointParams {
  const float* table; int64_t ld_table; int n_labels;
  const _Float16* table_h;   // optional 2-byte shadow of the label table (config 5: fp16 rows + fp32 master): the loss READS this one
  const float* feat; int64_t ld_feat; int n_feat;
  const int32_t* pos_from; const int32_t* pos_to; const int32_t* neg; const float* weights;
  int B, K, D;
  float K_cone, alpha, r_in, r_in_h;
  float lab_add, img_add;   // additive constant of the soft_clip forms (r_in for oe_h.py:328, K for oe.py:80,240)
  int label_proj, image_proj;
  float* e_pos; float* e_neg; float* loss;
  float* grad_table; float* grad_feat;
  float* partials; unsigned int* counter;
  // fixed-point hand-off of the loss (lec_common.h block_publish_fixed_point): used when an upper bound of the loss is known at launch (no per-positive weights,
  // a bounded energy) and the grid has fewer than 4096 blocks; fx_scale == 0: the ticket form
  unsigned long long* fx_acc; unsigned int* fx_flag; double fx_scale, fx_inv; float fx_bound;
  int iters;            // pair iterations per task
  int tasks_per_group;
  int lds_stage;        // T == 1 only: move rows through LDS (coalesced gather/scatter)
  // Row window (lec_joint_loss_fwd_bwd_window): this launch evaluates only the pairs it OWNS -- a pair with an image end point whose feature row r lies in
  // [row_lo, row_hi), and, when labels_too is set, the pairs between two labels.  A step whose CNN rows go through the backbone in chunks launches the
  // loss once per chunk, right behind that chunk's forward: every pair is evaluated exactly once over the launches, rows outside the window are never trusted
  // (they may not have been computed yet), e_pos / e_neg entries of pairs owned by other launches are left alone.  The plain entry: [0, INT_MAX), labels_too.
  int row_lo, row_hi, labels_too;
  int feat_base;        // feature row r of a node code lives at feat / grad_feat row r - feat_base (a chunk-sized buffer: feat_base = row_lo; else 0)
  const int32_t* window_dev;   // optional {row_lo, row_hi, labels_too, feat_base} in DEVICE memory, read at kernel start: one captured launch serves every chunk of a step
#ifdef LEC_JL_STAMP
  unsigned long long* stamps;   // instrumentation build only (tools/cone_timeline.sh): 10 words per wave, see jl_stamp below
#endif
};

// Instrumentation (NOT a kernel variant: the arithmetic is untouched).  `hipcc -DLEC_JL_STAMP` -- tools/cone_timeline.sh builds this file alone into its own
// library -- makes every wave record where its time goes: the constant 100 MHz clock at entry and exit (launch ramp, tail), and shader-clock cycles per phase.
#ifdef LEC_JL_STAMP
__device__ __forceinline__ unsigned long long jl_cycles() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
__device__ __forceinline__ unsigned long long jl_realtime() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
#define JL_STAMP(var) const unsigned long long var = jl_cycles()
#define JL_ACC(acc, a, b) acc += (b) - (a)
#else
#define JL_STAMP(var)
#define JL_ACC(acc, a, b)
#endif

// the launch owning the pair (a, b) of node codes: the window holding its image row.  CONTRACT of the windowed entry (include/lecone.h): a pair has AT MOST ONE
// image end point -- or both image rows lie in the SAME window.  A pair of two image rows in different windows (negatives of a trainer that does not pick per
// level) cannot be evaluated by any single window's launch (the other row is not addressable / not computed yet): the launch owning the larger row
// POISONS the loss with a NaN instead of returning energies from a zero row (pair_splits_windows below).
__device__ __forceinline__ bool pair_owned(const JointParams& P, int a, int b) {
  const int ra = a < 0 ? -1 - a : -1, rb = b < 0 ? -1 - b : -1;
  const int r = ra > rb ? ra : rb;
  return r < 0 ? P.labels_too != 0 : (r >= P.row_lo && r < P.row_hi);
}
__device__ __forceinline__ bool row_in_window(const JointParams& P, int code) {       // may this launch read / add to the row of node `code`?
  return code >= 0 || ((-1 - code) >= P.row_lo && (-1 - code) < P.row_hi);
}
__device__ __forceinline__ bool pair_splits_windows(const JointParams& P, int a, int b) {    // two image rows, not both inside this launch's window
  return a < 0 && b < 0 && !(row_in_window(P, a) && row_in_window(P, b));
}

// One projected row held across T lanes: raw e (after the +1e-15 of Embedder.forward), projected p, and the two
// scalars of the projection's Jacobian:  d/d e = A * go + Bc * <e, go> * e.
template <int EPL>
struct Row {
  float e[EPL];
  float p[EPL];
  float A, Bc;
};

// Packed fp32 arithmetic for the per-lane row loops (round 4).  At K = 256 the kernel is bound by vector-ALU issue (profiles/r04_cone_pmc.md: ~700 vector
// instructions per pair, most of them one multiply or one add of a row element); gfx950 issues v_pk_mul_f32 / v_pk_add_f32 on TWO elements at the rate of one.
// Element arithmetic is unchanged (IEEE multiply and add per element, nothing contracted); a dot product keeps two running sums (even / odd elements) and
// adds them at the end -- another summation order than the former element-by-element loop, like every order a fixed one against the reference's reduction.
typedef float f2v __attribute__((ext_vector_type(2)));
template <int N> __device__ __forceinline__ f2v pk_at(const float (&a)[N], int i) { f2v v; v.x = a[2 * i]; v.y = a[2 * i + 1]; return v; }
template <int N> __device__ __forceinline__ void pk_put(float (&a)[N], int i, f2v v) { a[2 * i] = v.x; a[2 * i + 1] = v.y; }
template <int N> __device__ __forceinline__ float pk_dot(const float (&a)[N], const float (&b)[N]) {
  static_assert(N % 2 == 0, "rows are held in pairs of elements");
  f2v acc = pk_at(a, 0) * pk_at(b, 0);
#pragma unroll
  for (int i = 1; i < N / 2; ++i) acc += pk_at(a, i) * pk_at(b, i);
  return acc.x + acc.y;
}
template <int N> __device__ __forceinline__ void pk_scale(float (&o)[N], const float (&a)[N], float k) {       // o = a * k
  const f2v kk = {k, k};
#pragma unroll
  for (int i = 0; i < N / 2; ++i) pk_put(o, i, pk_at(a, i) * kk);
}
template <int N> __device__ __forceinline__ void pk_axpby(float (&o)[N], float ca, const float (&a)[N], float cb, const float (&b)[N]) {   // o = ca a + cb b
  const f2v va = {ca, ca}, vb = {cb, cb};
#pragma unroll
  for (int i = 0; i < N / 2; ++i) pk_put(o, i, va * pk_at(a, i) + vb * pk_at(b, i));
}

// ---- T == 1 (one lane per pair, D <= 16): rows move between HBM/L2 and registers THROUGH LDS.  A lane-per-row global
// access makes every load/atomic instruction touch 64 different cache lines (one dword each); staged, consecutive lanes
// move consecutive elements of the gathered row list, so an instruction touches ~64/D rows in D-element runs, and each
// lane then reads its own row from LDS at an odd stride (conflict-free).
constexpr int kNoRow = INT_MIN;

// A row of the label table (fp32 master, or its fp16 shadow when one is given) or of the image features; reads element d.
struct RowSrc { const float* f; const _Float16* h; };
__device__ __forceinline__ bool code_in_range(const JointParams& P, int code) {      // a stale / corrupt node code must not become an
  if (code >= 0) return code < P.n_labels;                                            // out-of-bounds read or atomic: it is skipped
  const int r = (-1 - code) - P.feat_base;
  return r >= 0 && r < P.n_feat;
}
__device__ __forceinline__ RowSrc row_src(const JointParams& P, int code) {
  RowSrc r; r.f = nullptr; r.h = nullptr;
  if (code >= 0) { if (P.table_h) r.h = P.table_h + (int64_t)code * P.ld_table; else r.f = P.table + (int64_t)code * P.ld_table; }
  else r.f = P.feat + (int64_t)(-1 - code - P.feat_base) * P.ld_feat;
  return r;
}
__device__ __forceinline__ float row_ld(const RowSrc& r, int d) { return r.h ? (float)r.h[d] : r.f[d]; }

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int EPL>
__device__ __forceinline__ void wave_gather_rows(const JointParams& P, int code_or_norow, float* stage, float (&e)[EPL]) {
  constexpr int LDW = EPL + 1;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < EPL; ++k) {
    const int eidx = k * 64 + lane;
    const int r = eidx / EPL, d = eidx - r * EPL;
    const int rc = __shfl(code_or_norow, r, kWave);
    float v = 0.0f;
    if (rc != kNoRow && d < P.D && code_in_range(P, rc)) v = row_ld(row_src(P, rc), d);
    stage[r * LDW + d] = v;
  }
  wave_sync();
#pragma unroll
  for (int i = 0; i < EPL; ++i) e[i] = stage[lane * LDW + i];
  wave_sync();
}

template <int EPL>
__device__ __forceinline__ void wave_scatter_rows(const JointParams& P, int code_or_norow, float* stage, const float (&g)[EPL]) {
  constexpr int LDW = EPL + 1;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < EPL; ++i) stage[lane * LDW + i] = g[i];
  wave_sync();
#pragma unroll
  for (int k = 0; k < EPL; ++k) {
    const int eidx = k * 64 + lane;
    const int r = eidx / EPL, d = eidx - r * EPL;
    const int rc = __shfl(code_or_norow, r, kWave);
    if (rc != kNoRow && d < P.D && code_in_range(P, rc)) {
      float* dst = rc >= 0 ? P.grad_table + (int64_t)rc * P.ld_table : P.grad_feat + (int64_t)(-1 - rc - P.feat_base) * P.ld_feat;
      atomicAdd(dst + d, stage[r * LDW + d]);
    }
  }
  wave_sync();
}

// raw row elements of node `code` owned by this lane (d = t, t+T, ...); kNoRow -> zeros, no memory access.
// Lane-per-row geometries (T == 1): every load instruction of the wave touches 64 different rows, so the texture path sees one request per lane per
// instruction: fp32 rows are fetched as 8-byte pairs when the row pitch keeps them aligned (D = 10: five requests per row instead of ten --
// profiles/r05_cone_timeline.md: the row phase was 44 % of a wave's life at 4 096 x 256 x 10).
template <int T, int EPL>
__device__ __forceinline__ void fetch_row(const JointParams& P, int code, int t, float (&raw)[EPL]) {
  const bool ok = code != kNoRow && code_in_range(P, code);
  RowSrc src; src.f = P.table; src.h = nullptr;
  if (ok) src = row_src(P, code);
  if (T == 1 && EPL % 2 == 0) {
    const int64_t ld = code >= 0 ? P.ld_table : P.ld_feat;
    if (src.h == nullptr && (ld & 1) == 0 && (P.D & 1) == 0) {             // (wave-uniform but for mixed label / image rows with different pitches: both even here)
      // 16-byte pieces at 8-byte alignment (gfx950 global loads need dword alignment only), then an 8-byte tail: D = 10 -> 16 + 16 + 8: three requests per row
      typedef float f4u __attribute__((ext_vector_type(4), aligned(8)));
      constexpr int NQ = EPL / 4;
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        if (ok && 4 * i + 4 <= P.D) {
          const f4u v = *(const f4u*)(src.f + 4 * i);
          raw[4 * i] = v.x; raw[4 * i + 1] = v.y; raw[4 * i + 2] = v.z; raw[4 * i + 3] = v.w;
        } else {
          f2v a = {0.0f, 0.0f}, b2 = {0.0f, 0.0f};
          if (ok && 4 * i < P.D) a = *(const f2v*)(src.f + 4 * i);
          if (ok && 4 * i + 2 < P.D) b2 = *(const f2v*)(src.f + 4 * i + 2);
          raw[4 * i] = a.x; raw[4 * i + 1] = a.y; raw[4 * i + 2] = b2.x; raw[4 * i + 3] = b2.y;
        }
      }
#pragma unroll
      for (int i = 2 * NQ; i < EPL / 2; ++i) {
        f2v v = {0.0f, 0.0f};
        if (ok && 2 * i < P.D) v = *(const f2v*)(src.f + 2 * i);
        raw[2 * i] = v.x; raw[2 * i + 1] = v.y;
      }
      return;
    }
    if (src.h != nullptr && (P.ld_table & 1) == 0 && (P.D & 1) == 0) {     // fp16 shadow rows (config 5): two elements per 4-byte request
      typedef _Float16 h2v __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int i = 0; i < EPL / 2; ++i) {
        h2v v = {(_Float16)0.0f, (_Float16)0.0f};
        if (ok && 2 * i < P.D) v = *(const h2v*)(src.h + 2 * i);
        raw[2 * i] = (float)v.x; raw[2 * i + 1] = (float)v.y;
      }
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < EPL; ++i) {
    const int d = t + i * T;
    raw[i] = (ok && d < P.D) ? row_ld(src, d) : 0.0f;
  }
}

// sum over the wave's pair slots (lanes l, l + T, l + 2T, ...) for every t: the total lands in EVERY lane.  Inside a 16-lane row the steps are DPP rotations
// (full-rate vector instructions, no LDS crossbar); the two cross-row steps stay ds_bpermute butterflies.  (Was: six ds_bpermute butterflies per value --
// 144 of them for the two gradient rows of a lane-per-pair wave: 1.95 us of a 15.6 us wave at config 5, profiles/r05_cone_timeline.md.)
template <int T>
__device__ __forceinline__ float slot_sum(float v) {
  if (T <= 1) v += dpp_move<0x121>(v);                 // row_ror:1
  if (T <= 2) v += dpp_move<0x122>(v);                 // row_ror:2
  if (T <= 4) v += dpp_move<0x124>(v);                 // row_ror:4
  if (T <= 8) v += dpp_move<0x128>(v);                 // row_ror:8
  if (T <= 16) v += __shfl_xor(v, 16, kWave);
  if (T <= 32) v += __shfl_xor(v, 32, kWave);
  return v;
}

template <int T, int EPL>
__device__ __forceinline__ void load_project(const JointParams& P, int code, bool valid, int t, Row<EPL>& r, float* stage,
                                             const float* prefetched = nullptr) {
  const bool is_label = code >= 0;
  valid = valid && code_in_range(P, code);
  RowSrc src; src.f = nullptr; src.h = nullptr;
  if (valid) src = row_src(P, code);
  const bool hyp = valid && is_label && P.label_proj == LEC_LABEL_HYP;
  const bool img = valid && (is_label ? P.label_proj == LEC_LABEL_SOFTCLIP_K : P.image_proj != LEC_IMAGE_RAW);   // soft_clip forms
  float nn;
  if (prefetched != nullptr) {
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      float v = valid ? prefetched[i] : 0.0f;                                       // (a pair this launch does not own: its row may not exist yet)
      if (hyp && (t + i * T) < P.D) v += 1e-15f;                                    // oe_h.py:79
      r.e[i] = v;
    }
  } else if (T == 1 && stage != nullptr) {
    wave_gather_rows<EPL>(P, valid ? code : kNoRow, stage, r.e);
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      float v = r.e[i];
      if (hyp && i < P.D) v += 1e-15f;                                              // oe_h.py:79
      r.e[i] = v;
    }
  } else {
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      int d = t + i * T;
      float v = 0.0f;
      if (valid && d < P.D) { v = row_ld(src, d); if (hyp) v += 1e-15f; }          // oe_h.py:79
      r.e[i] = v;
    }
  }
  nn = group_sum<T>(pk_dot(r.e, r.e));
  const float n = sqrtf(nn);                                                       // oe_h.py:81 / :327
  const float den = fmaxf(n, 1e-12f);                                              // F.normalize eps
  const float denp = n >= 1e-12f ? 1.0f : 0.0f;
  float mul = 1.0f, A = 1.0f, Bc = 0.0f;
  if (hyp) {
    float arg = P.r_in_h + n;
    float argc = arg < -15.0f ? -15.0f : (arg > 15.0f ? 15.0f : arg);
    float th = tanhf(argc);                                                        // oe_h.py:83
    float tp = (arg >= -15.0f && arg <= 15.0f) ? 1.0f - th * th : 0.0f;
    mul = th; A = th / den;
    Bc = n > 0.0f ? (tp / den - th * denp / (den * den)) / n : 0.0f;
  } else if (img) {
    float sc = n + (is_label ? P.lab_add : P.img_add);                             // oe_h.py:328 / oe.py:80,240
    mul = sc; A = sc / den;
    Bc = n > 0.0f ? (1.0f / den - sc * denp / (den * den)) / n : 0.0f;
  }
  const float mul_over_den = mul / den;                 // one correctly rounded divide per row, then multiplies
  pk_scale(r.p, r.e, (hyp || img) ? mul_over_den : 1.0f);                          // (raw rows: times 1, exact)
  // no-grad clip of label points into [r_in, 1-1e-5] (oe_h.py:100-103); images are NOT clipped (:323-328)
  const float pp = group_sum<T>(pk_dot(r.p, r.p));
  if (hyp) {
    float no = sqrtf(pp);
    if (no <= P.r_in) pk_scale(r.p, r.p, P.r_in / no);
    else if (no >= 1.0f) pk_scale(r.p, r.p, (float)(1.0 - 1e-5) / no);
  }
  r.A = A; r.Bc = Bc;
}

// chain a gradient w.r.t. the projected row back to the raw row and add it to the owner buffer
template <int T, int EPL>
__device__ __forceinline__ void scatter_row_grad(const JointParams& P, int code, bool active, int t,
                                                 const Row<EPL>& r, const float (&go)[EPL], float* stage) {
  const float dot = group_sum<T>(pk_dot(r.e, go));
  float graw[EPL];
  pk_axpby(graw, r.A, go, r.Bc * dot, r.e);
  if (T == 1 && stage != nullptr) {
    if (__ballot(active) == 0ull) return;                                           // wave-uniform: nothing to add
    wave_scatter_rows<EPL>(P, active ? code : kNoRow, stage, graw);
    return;
  }
  if (!active) return;
  float* dst = code >= 0 ? P.grad_table + (int64_t)code * P.ld_table : P.grad_feat + (int64_t)(-1 - code - P.feat_base) * P.ld_feat;
#pragma unroll
  for (int i = 0; i < EPL; ++i) {
    int d = t + i * T;
    if (d < P.D) atomicAdd(dst + d, graw[i]);
  }
}

template <int T, int EPL, int ENERGY, bool GRAD, bool STAGE>
// (Occupancy caps were measured in round 4: the lane-per-pair instance (T = 1, 12 elements per lane) holds 181 registers = 2 waves per SIMD; capped to 3 waves (168
// registers, 9 spilled) / 4 waves (128, 53 spilled), us per launch: 256 x 256 x 10: 22.1 / 22.8 / 27.0; 4 096 x 256 x 10: 89.0 / 97.1 / 112.4; 256 x 256 x 128:
// 58.8 / 58.8 / 60.8 -- the spills cost more than the third wave hides.  Removed.)
__global__ __launch_bounds__(512) void joint_loss_kernel(JointParams P_in) {
  JointParams P = P_in;
  if (P.window_dev) {                                 // (uniform scalar loads; the arguments' own window is the fallback)
    P.row_lo = P.window_dev[0]; P.row_hi = P.window_dev[1]; P.labels_too = P.window_dev[2]; P.feat_base = P.window_dev[3];
  }
  constexpr int PPW = kWave / T;                      // pairs per wave iteration
  const int lane = threadIdx.x & 63;
  const int t = lane % T, slot = lane / T;
  const int wave_global = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int b = wave_global / P.tasks_per_group;
  const int c = wave_global - b * P.tasks_per_group;
  const bool task_valid = b < P.B;
  const int NP = 1 + 2 * P.K;
  float lsum = 0.0f;
#ifdef LEC_JL_STAMP
  const unsigned long long rt0 = jl_realtime();
  unsigned long long c_rows = 0, c_energy = 0, c_bwd = 0;
#endif
  JL_STAMP(ck0);
  constexpr int kStage = (T == 1 && STAGE) ? kWave * (EPL + 1) : 1;
  __shared__ float s_stage[8 * kStage];
  float* stage = (T == 1 && STAGE) ? s_stage + (threadIdx.x >> 6) * kStage : nullptr;

  // NOTE: every lane of the wave walks the same control flow (butterflies need all 64 lanes); validity is a predicate.
  int ucode = 0, vcode = 0; float w = 1.0f;
  if (task_valid) {
    ucode = P.pos_from[b]; vcode = P.pos_to[b];
    if (P.weights) w = P.weights[b];
  }
  float gu[EPL], gv[EPL];
#pragma unroll
  for (int i = 0; i < EPL; ++i) { gu[i] = 0.0f; gv[i] = 0.0f; }

  const int q0 = c * P.iters * PPW;
  // Software pipeline: node codes are fetched two iterations ahead and raw rows one iteration ahead, so the
  // code -> row -> arithmetic dependency chain of iteration `it` overlaps the arithmetic of iteration `it - 1`
  // (the waves of this kernel spent 65 % of their cycles in s_waitcnt before this: profiles/r01_cone_pmc.md).
  auto fetch_code = [&](int it) -> int {
    const int q = q0 + it * PPW + slot;
    return (task_valid && it < P.iters && q > 0 && q < NP) ? P.neg[(int64_t)b * 2 * P.K + (q - 1)] : kNoRow;
  };
  constexpr bool pipelined = !(T == 1 && STAGE);
  // the first two iterations' codes and the first iteration's rows are requested BEFORE u_b / v_b are projected: their two dependent round trips ride under the
  // projection's own (code -> row -> sqrt / tanh / divide chain), instead of starting after it (config 5: 3.5 us + 2.8 us back to back in a 15.6 us wave)
  // (Rows TWO iterations ahead, codes three, were tried in round 5: same launch times at 4 waves per block, 5 % slower at 8 (197 registers instead of 181):
  // the row phase is bound by the texture path's one-address-per-lane rate, not by the distance of the prefetch.)
  int code_a = fetch_code(0), code_b = fetch_code(1);
  float raw_a[EPL];
  if (pipelined) fetch_row<T, EPL>(P, code_a, t, raw_a);
  Row<EPL> U, V;
  load_project<T, EPL>(P, ucode, task_valid && row_in_window(P, ucode), t, U, nullptr);     // every lane needs u_b, v_b: broadcast loads
  load_project<T, EPL>(P, vcode, task_valid && row_in_window(P, vcode), t, V, nullptr);     // (an image row outside this launch's window stays zeros: it may not exist yet; no owned pair uses it -- pair_splits_windows)
  JL_STAMP(ck1);                                      // u_b, v_b gathered and projected
  for (int it = 0; it < P.iters; ++it) {
    JL_STAMP(ci0);
    const int q = q0 + it * PPW + slot;               // pair index in the group: 0 = positive, 1+k = negative slot k
    const int kind_q = (!task_valid || q >= NP) ? 0 : (q == 0 ? 0 : (q - 1 < P.K ? 1 : 2));
    // (whole-batch launches own every pair; a windowed launch skips the pairs of other windows: their rows may not exist yet)
    const bool valid = task_valid && q < NP && pair_owned(P, kind_q == 2 ? (code_a == kNoRow ? 0 : code_a) : ucode, kind_q == 1 ? (code_a == kNoRow ? 0 : code_a) : vcode);
    const int kind = !valid ? 0 : kind_q;   // 1: u fixed (corrupt `to`), 2: v fixed
    const int code_c = fetch_code(it + 2);
    float raw_b[EPL];
    if (pipelined) fetch_row<T, EPL>(P, code_b, t, raw_b);
    const int ocode = code_a == kNoRow ? 0 : code_a;
    Row<EPL> O;
    load_project<T, EPL>(P, ocode, valid && q > 0, t, O, stage, pipelined ? raw_a : nullptr);
    code_a = code_b; code_b = code_c;
    if (pipelined) {
#pragma unroll
      for (int i = 0; i < EPL; ++i) raw_a[i] = raw_b[i];
    }

    float x[EPL], y[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      x[i] = kind == 2 ? O.p[i] : U.p[i];
      y[i] = kind == 1 ? O.p[i] : V.p[i];
    }
    JL_STAMP(ci1);                                    // the iteration's rows have arrived (issued one iteration ahead) and are projected
    JL_ACC(c_rows, ci0, ci1);
    // ---- forward: energy of every pair of this iteration
    float E;
    ConeFwd cf;
    ConeEval ev;
    if (ENERGY == LEC_ENERGY_HYP_CONE) {
      float df[EPL];
#pragma unroll
      for (int i = 0; i < EPL / 2; ++i) pk_put(df, i, pk_at(x, i) - pk_at(y, i));
      const float xx = group_sum<T>(pk_dot(x, x)), yy = group_sum<T>(pk_dot(y, y)), s = group_sum<T>(pk_dot(x, y)), dd = group_sum<T>(pk_dot(df, df));
      cf = cone_forward(xx, yy, s, dd, P.K_cone);
      E = cf.E;
    } else if (ENERGY == LEC_ENERGY_EUC_CONE) {                                    // oe.py:721-739
      float xx = 0.f, u = 0.f, dd = 0.f;
#pragma unroll
      for (int i = 0; i < EPL; ++i) {
        float df = y[i] - x[i];
        xx += x[i] * x[i]; u += x[i] * df; dd += df * df;
      }
      xx = group_sum<T>(xx); u = group_sum<T>(u); dd = group_sum<T>(dd);
      ev = euc_cone_eval<GRAD>(xx, dd, u, P.K_cone);
      E = ev.E;
    } else {                                                                       // order_embeddings.py:818-824
      float e = 0.0f;
#pragma unroll
      for (int i = 0; i < EPL; ++i) { float m = fmaxf(x[i] - y[i], 0.0f); e += m * m; }
      E = group_sum<T>(e);
    }
    if (valid && pair_splits_windows(P, kind == 2 ? ocode : ucode, kind == 1 ? ocode : vcode)) E = __builtin_nanf("");    // contract violation: loud, not silently wrong
    if (valid && t == 0) {
      if (q == 0) { P.e_pos[b] = E; lsum += w * E; }
      else {
        P.e_neg[(int64_t)b * 2 * P.K + (q - 1)] = E;
        float h = P.alpha - E;
        lsum += w * (h < 0.0f ? 0.0f : h);                                         // oe_h.py:839,846
      }
    }
    JL_STAMP(ci2);                                    // energies evaluated and stored
    JL_ACC(c_energy, ci1, ci2);
    // ---- backward: only pairs whose loss term is live carry gradient (positives; negatives inside the margin).
    // The whole wave skips the gradient arithmetic, the Jacobian chain and the scatter when none of its pairs is live.
    if (GRAD) {
      const float g = !valid ? 0.0f : (q == 0 ? w : ((P.alpha - E) >= 0.0f ? -w : 0.0f));   // oe_h.py:846 (+ clamp mask)
      const bool act = valid && g != 0.0f;
      if (__ballot(act) != 0ull) {
        float go[EPL];
        if (ENERGY == LEC_ENERGY_HYP_CONE || ENERGY == LEC_ENERGY_EUC_CONE) {
          float cxx, cxy, cyy;
          if (ENERGY == LEC_ENERGY_HYP_CONE) cone_grad_coeffs(cf, P.K_cone, cxx, cxy, cyy);
          else { cxx = ev.cxx; cxy = ev.cxy; cyy = ev.cyy; }
          // (pairs that are not live carry exact zeros -- selected, not multiplied: a pair another window owns may hold a row that is not computed yet)
          cxx = act ? cxx * g : 0.0f; cxy = act ? cxy * g : 0.0f; cyy = act ? cyy * g : 0.0f;
          float gx[EPL], gy[EPL];
          pk_axpby(gx, cxx, x, cxy, y); pk_axpby(gy, cxy, x, cyy, y);
          const float mu_ = kind != 2 ? 1.0f : 0.0f, mv_ = kind != 1 ? 1.0f : 0.0f;
          const f2v mu2 = {mu_, mu_}, mv2 = {mv_, mv_};
#pragma unroll
          for (int i = 0; i < EPL / 2; ++i) {
            pk_put(gu, i, pk_at(gu, i) + pk_at(gx, i) * mu2);
            pk_put(gv, i, pk_at(gv, i) + pk_at(gy, i) * mv2);
          }
#pragma unroll
          for (int i = 0; i < EPL; ++i) go[i] = kind == 1 ? gy[i] : gx[i];
        } else {
#pragma unroll
          for (int i = 0; i < EPL; ++i) {
            const float m = 2.0f * fmaxf(x[i] - y[i], 0.0f) * g;
            if (act && kind != 2) gu[i] += m;
            if (act && kind != 1) gv[i] -= m;
            go[i] = kind == 1 ? -m : m;
          }
        }
        scatter_row_grad<T, EPL>(P, ocode, act && kind != 0, t, O, go, stage);
      }
    }
    JL_STAMP(ci3);                                    // gradient coefficients, Jacobian chain, atomics issued
    JL_ACC(c_bwd, ci2, ci3);
  }
  JL_STAMP(ck2);

  if (GRAD) {
    // sum the per-slot partial gradients of u_b and v_b across the wave's pair slots, then one row-add each
#pragma unroll
    for (int i = 0; i < EPL; ++i) { gu[i] = slot_sum<T>(gu[i]); gv[i] = slot_sum<T>(gv[i]); }
    // (Re-fetching the raw rows of u_b / v_b here instead of carrying them through the pair loop -- 181 -> 159 registers, a third wave per SIMD -- was measured in
    // round 5: 18.4 -> 20.4 us at config 5, 73.5 -> 76.5 at 4 096 x 256 x 10: one more dependent round trip in every wave's tail costs more than the wave buys.)
    scatter_row_grad<T, EPL>(P, ucode, task_valid && slot == 0 && row_in_window(P, ucode), t, U, gu, stage);
    scatter_row_grad<T, EPL>(P, vcode, task_valid && slot == 0 && row_in_window(P, vcode), t, V, gv, stage);
  }

  JL_STAMP(ck3);                                      // u_b / v_b gradients reduced over the wave and added
  lsum = group_sum<64>(lsum);
  if (P.fx_scale != 0.0) block_publish_fixed_point(lsum, P.fx_acc, P.fx_flag, P.loss, P.fx_scale, P.fx_inv, P.fx_bound);
  else block_publish_and_finalize(lsum, P.partials, P.counter, P.loss, 1.0f);
#ifdef LEC_JL_STAMP
  if (P.stamps && lane == 0) {
    const unsigned long long ck4 = jl_cycles(), rt1 = jl_realtime();
    unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* o = P.stamps + (size_t)wave_global * 10;
    o[0] = rt0; o[1] = rt1; o[2] = ck1 - ck0; o[3] = c_rows; o[4] = c_energy; o[5] = c_bwd; o[6] = ck3 - ck2; o[7] = ck4 - ck3; o[8] = ck4 - ck0;
    o[9] = ((unsigned long long)xcc << 32) | hw;
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------
template <int T, int EPL>
static int launch(const JointParams& P, bool grad, int energy, int nblocks, int wpb, hipStream_t st) {
#define LEC_JL(E_, G_, S_) hipLaunchKernelGGL((joint_loss_kernel<T, EPL, E_, G_, S_>), dim3(nblocks), dim3(64 * wpb), 0, st, P)
  if (T == 1 && P.lds_stage) {
    if (energy == LEC_ENERGY_HYP_CONE) { if (grad) LEC_JL(LEC_ENERGY_HYP_CONE, true, (T == 1)); else LEC_JL(LEC_ENERGY_HYP_CONE, false, (T == 1)); }
    else if (energy == LEC_ENERGY_EUC_CONE) { if (grad) LEC_JL(LEC_ENERGY_EUC_CONE, true, (T == 1)); else LEC_JL(LEC_ENERGY_EUC_CONE, false, (T == 1)); }
    else { if (grad) LEC_JL(LEC_ENERGY_ORDER, true, (T == 1)); else LEC_JL(LEC_ENERGY_ORDER, false, (T == 1)); }
  } else {
    if (energy == LEC_ENERGY_HYP_CONE) { if (grad) LEC_JL(LEC_ENERGY_HYP_CONE, true, false); else LEC_JL(LEC_ENERGY_HYP_CONE, false, false); }
    else if (energy == LEC_ENERGY_EUC_CONE) { if (grad) LEC_JL(LEC_ENERGY_EUC_CONE, true, false); else LEC_JL(LEC_ENERGY_EUC_CONE, false, false); }
    else { if (grad) LEC_JL(LEC_ENERGY_ORDER, true, false); else LEC_JL(LEC_ENERGY_ORDER, false, false); }
  }
#undef LEC_JL
  LEC_CHECK_LAUNCH("joint_loss_kernel");
  return LEC_OK;
}

struct JointGeom { int T, EPL, iters, tasks_per_group, nblocks, wpb; };

// (T lanes per pair, EPL row elements per lane) candidates: T*EPL >= D.  Small T = less redundant scalar math per pair
// (the cone evaluation is computed by every lane of the T-group) and fewer butterflies; large T = coalesced row loads
// for long rows and fewer registers.
static const int kGeoms[][2] = {{1, 4}, {1, 12}, {1, 16}, {2, 8}, {4, 4}, {4, 8}, {8, 8}, {16, 4}, {16, 8}, {32, 8}, {64, 4}, {64, 8}, {64, 16}};

static bool joint_geometry(int B, int K, int D, JointGeom& g) {
  if (D <= 4) { g.T = 1; g.EPL = 4; }
  else if (D <= 12) { if (1 + 2 * K <= 32) { g.T = 4; g.EPL = 4; } else { g.T = 1; g.EPL = 12; } }   // short groups: keep lanes busy
  else if (D <= 16) { g.T = 2; g.EPL = 8; }
  else if (D <= 32) { g.T = 4; g.EPL = 8; }
  else if (D <= 64) { g.T = 8; g.EPL = 8; }
  else if (D <= 128) { g.T = 16; g.EPL = 8; }
  else if (D <= 256) { g.T = 32; g.EPL = 8; }
  else if (D <= 512) { g.T = 64; g.EPL = 8; }
  else if (D <= 1024) { g.T = 64; g.EPL = 16; }
  else return false;
  int iters_override = 0;
  if (tuning().jl_T > 0 && tuning().jl_T * tuning().jl_EPL >= D) {            // sweep hook (LEC_JOINT_GEOM="T,EPL[,iters]", resolved at load: tuning.h)
    for (auto& c : kGeoms) if (c[0] == tuning().jl_T && c[1] == tuning().jl_EPL) { g.T = c[0]; g.EPL = c[1]; iters_override = tuning().jl_iters; }
  }
  const int ppw = 64 / g.T, NP = 1 + 2 * K;
  const int64_t group_iters = (NP + ppw - 1) / ppw;                      // wave iterations one group needs
  // How many waves share one positive's group of 1 + 2K pairs.  Every wave projects u_b and v_b itself and ends with two row scatters, a loss
  // partial and a ticket: fixed work per wave, so FEWER, LONGER waves win as long as the chip stays full.  Measured on the MI355X
  // (tools/sweep_cone_r4b.sh, profiles/r04_cone_sweep.md): lane-per-pair geometries (T <= 2) want ~1.5k waves (B x K x D = 256 x 256 x 10:
  // 34.0 us at 2 304 waves, 22.3 at 1 280, 22.7 at 768), lanes-per-row geometries (T >= 4) ~3k (256 x 256 x 128: 77.8 us at 5 632 waves, 59.2 at
  // 3 072, 69.6 at 1 536); beyond that ONE wave per group, however many iterations that takes (4 096 x 256 x 10: 123.7 us with two waves
  // of 6 + 3 iterations per group -- the old cap of 6 -- 88.5 with one wave of 9; 4 096 x 64 x 128: 287.8 -> 183.3 us).  The split is balanced:
  // `tasks` waves of ceil(group_iters / tasks) iterations each.
  const int64_t target_waves = g.T <= 2 ? 1536 : 3072;
  int64_t tasks = (target_waves + B - 1) / B;
  if (tasks > group_iters) tasks = group_iters;
  if (tasks < 1) tasks = 1;
  int64_t iters = (group_iters + tasks - 1) / tasks;
  if (iters < 1) iters = 1;
  if (iters_override > 0) iters = iters_override;
  g.iters = (int)iters;
  g.tasks_per_group = (int)((group_iters + iters - 1) / iters);
  int64_t waves = (int64_t)B * g.tasks_per_group;
  // waves per block: every block ends with ONE returning atomic on the launch's ticket, and same-address atomics serialize (~15 ns each): 320 blocks of 4 waves
  // spend ~4.6 us of a 19 us launch there (profiles/r05_cone_timeline.md).  Lane-per-pair instances (2 waves per SIMD fit) take 8-wave blocks.
  g.wpb = tuning().jl_wpb > 0 ? tuning().jl_wpb : (g.T <= 2 ? 8 : 4);
  g.nblocks = (int)((waves + g.wpb - 1) / g.wpb);
  return true;
}

static int dispatch(const JointGeom& g, const JointParams& P, bool grad, int energy, hipStream_t st) {
#define LEC_G(T_, E_) if (g.T == T_ && g.EPL == E_) return launch<T_, E_>(P, grad, energy, g.nblocks, g.wpb, st)
  LEC_G(1, 4); LEC_G(1, 12); LEC_G(1, 16); LEC_G(2, 8); LEC_G(4, 4); LEC_G(4, 8); LEC_G(8, 8); LEC_G(16, 4); LEC_G(16, 8);
  LEC_G(32, 8); LEC_G(64, 4); LEC_G(64, 8); LEC_G(64, 16);
#undef LEC_G
  set_error("joint_loss: no kernel for geometry T=%d EPL=%d", g.T, g.EPL);
  return LEC_E_ARG;
}

}  // namespace lec

extern "C" int64_t lec_loss_workspace_bytes(int B, int K, int D) {
  lec::JointGeom g;
  if (B < 0 || K < 0 || !lec::joint_geometry(B > 0 ? B : 1, K, D > 0 ? D : 1, g)) return LEC_E_ARG;
  return 256 + (int64_t)g.nblocks * sizeof(float);
}

#ifdef LEC_JL_STAMP
static unsigned long long* g_jl_stamps = nullptr;
static unsigned long long* lec_jl_stamp_target() { return g_jl_stamps; }
extern "C" void lec_jl_set_stamp_buffer(void* p) { g_jl_stamps = (unsigned long long*)p; }      // 10 x waves 64-bit words, device memory (instrumentation build only)
extern "C" int lec_jl_geometry(int B, int K, int D, int* T, int* EPL, int* iters, int* tasks_per_group, int* waves) {
  lec::JointGeom g;
  if (!lec::joint_geometry(B, K, D, g)) return -1;
  *T = g.T; *EPL = g.EPL; *iters = g.iters; *tasks_per_group = g.tasks_per_group; *waves = B * g.tasks_per_group;
  return 0;
}
#endif

static int joint_loss_impl(int energy, int label_proj, int image_proj,
                                      const float* table, const void* table_f16, int64_t ld_table, int n_labels,
                                      const float* feat, int64_t ld_feat, int n_feat,
                                      const int32_t* pos_from, const int32_t* pos_to, const int32_t* neg,
                                      const float* weights, int B, int K, int D, float K_cone, float alpha,
                                      float* e_pos, float* e_neg, float* loss, float* grad_table, float* grad_feat,
                                      void* workspace, int64_t workspace_bytes, lec_stream_t stream,
                                      int row_lo = 0, int row_hi = INT_MAX, int labels_too = 1, const int32_t* window_dev = nullptr) {
  using namespace lec;
  LEC_CHECK_ARG(energy >= LEC_ENERGY_HYP_CONE && energy <= LEC_ENERGY_EUC_CONE, "joint_loss: unknown energy %d", energy);
  LEC_CHECK_ARG(label_proj >= LEC_LABEL_RAW && label_proj <= LEC_LABEL_SOFTCLIP_K, "joint_loss: unknown label_proj %d", label_proj);
  LEC_CHECK_ARG(image_proj >= LEC_IMAGE_RAW && image_proj <= LEC_IMAGE_SOFTCLIP_K, "joint_loss: unknown image_proj %d", image_proj);
  LEC_CHECK_ARG(B > 0 && K >= 0 && D > 0, "joint_loss: B=%d K=%d D=%d must be positive (K >= 0)", B, K, D);
  LEC_CHECK_ARG((table || table_f16) && n_labels > 0 && ld_table >= D, "joint_loss: table null or ld_table < D");
  LEC_CHECK_ARG(n_feat == 0 || (feat && ld_feat >= D), "joint_loss: feat null or ld_feat < D");
  LEC_CHECK_ARG(pos_from && pos_to && (K == 0 || neg), "joint_loss: null index arrays");
  LEC_CHECK_ARG(e_pos && (K == 0 || e_neg) && loss, "joint_loss: null outputs");
  LEC_CHECK_ARG((grad_table == nullptr) == (grad_feat == nullptr) || n_feat == 0,
                "joint_loss: pass both grad_table and grad_feat, or neither");
  JointGeom g;
  LEC_CHECK_ARG(joint_geometry(B, K, D, g), "joint_loss: embedding_dim %d not supported (max 1024)", D);
  const int64_t need = 256 + (int64_t)g.nblocks * sizeof(float);
  LEC_CHECK_ARG(workspace && workspace_bytes >= need, "joint_loss: workspace too small (%lld < %lld)",
                (long long)workspace_bytes, (long long)need);
  hipStream_t st = (hipStream_t)stream;
  JointParams P;
  P.table = table; P.table_h = (const _Float16*)table_f16; P.ld_table = ld_table; P.n_labels = n_labels;
  P.feat = feat; P.ld_feat = ld_feat; P.n_feat = n_feat;
  P.pos_from = pos_from; P.pos_to = pos_to; P.neg = neg; P.weights = weights;
  P.B = B; P.K = K; P.D = D; P.K_cone = K_cone; P.alpha = alpha;
  P.r_in = inner_radius_f(K_cone); P.r_in_h = inner_radius_h_f(K_cone);
  P.label_proj = label_proj; P.image_proj = image_proj;
  P.lab_add = K_cone; P.img_add = image_proj == LEC_IMAGE_SOFTCLIP_K ? K_cone : P.r_in;
  P.e_pos = e_pos; P.e_neg = e_neg; P.loss = loss; P.grad_table = grad_table; P.grad_feat = grad_feat;
  P.counter = (unsigned int*)workspace; P.partials = (float*)((char*)workspace + 256);
  // The loss as ONE integer atomic per block when its total is bounded at launch: sum_b (E+_b + sum_k max(0, alpha - E-_bk)) <= B (E_max + 2K alpha), with
  // E_max = pi + pi/2 for the hyperbolic cone (acos - asin, oe_h.py:826-833) and 2 for the Euclidean cone (theta in [-1, 1], psi in [-1, 0], oe.py:733-739); the
  // order-embedding energy is unbounded and per-positive weights are the caller's: both keep the ticket form.  (workspace bytes 16..31: accumulator + flag.)
  P.fx_acc = (unsigned long long*)((char*)workspace + 16); P.fx_flag = (unsigned int*)((char*)workspace + 24);
  P.fx_scale = 0.0; P.fx_inv = 0.0; P.fx_bound = 0.0f;
  if (weights == nullptr && energy != LEC_ENERGY_ORDER && g.nblocks < 4096 && alpha >= 0.0f && tuning().jl_fixed_point) {
    const double e_max = energy == LEC_ENERGY_HYP_CONE ? 4.7123889803846899 + 1e-3 : 2.0 + 1e-3;
    const double bound = (double)B * (e_max + 2.0 * (double)K * (double)alpha) * 1.0001 + 1.0;
    int bits = 1; while (bits < 40 && (double)(1ull << bits) <= bound) ++bits;      // integer bits of the total
    if (bits < 40) { const int F = 51 - bits; P.fx_scale = (double)(1ull << F); P.fx_inv = 1.0 / P.fx_scale; P.fx_bound = (float)bound; }
  }
  P.iters = g.iters; P.tasks_per_group = g.tasks_per_group;
  P.lds_stage = tuning().jl_stage;
  LEC_CHECK_ARG(row_lo >= 0 && row_hi >= row_lo, "joint_loss: row window [%d, %d)", row_lo, row_hi);
  P.row_lo = row_lo; P.row_hi = row_hi; P.labels_too = labels_too ? 1 : 0;
  P.feat_base = 0; P.window_dev = window_dev;
#ifdef LEC_JL_STAMP
  P.stamps = lec_jl_stamp_target();
#endif
  const bool grad = grad_table != nullptr;
  return dispatch(g, P, grad, energy, st);
}

extern "C" int lec_joint_loss_fwd_bwd(int energy, int label_proj, int image_proj,
                                      const float* table, int64_t ld_table, int n_labels,
                                      const float* feat, int64_t ld_feat, int n_feat,
                                      const int32_t* pos_from, const int32_t* pos_to, const int32_t* neg,
                                      const float* weights, int B, int K, int D, float K_cone, float alpha,
                                      float* e_pos, float* e_neg, float* loss, float* grad_table, float* grad_feat,
                                      void* workspace, int64_t workspace_bytes, lec_stream_t stream) {
  LEC_CHECK_ARG(table, "joint_loss: table null");
  return joint_loss_impl(energy, label_proj, image_proj, table, nullptr, ld_table, n_labels, feat, ld_feat, n_feat, pos_from, pos_to, neg, weights,
                         B, K, D, K_cone, alpha, e_pos, e_neg, loss, grad_table, grad_feat, workspace, workspace_bytes, stream);
}

// Config 5 of BASELINE.json ("fp16+fp32-master"): the label table is READ from a 2-byte fp16 shadow [n_labels, D] (ld_table in
// elements) -- half the gather traffic of the 2(1+K) rows per positive -- while gradients still go to the fp32 grad_table and
// the fp32 master is what lec_table_step_adam_f16 updates (refreshing the shadow in the same pass).
extern "C" int lec_joint_loss_fwd_bwd_f16(int energy, int label_proj, int image_proj,
                                          const void* table_f16, int64_t ld_table, int n_labels,
                                          const float* feat, int64_t ld_feat, int n_feat,
                                          const int32_t* pos_from, const int32_t* pos_to, const int32_t* neg,
                                          const float* weights, int B, int K, int D, float K_cone, float alpha,
                                          float* e_pos, float* e_neg, float* loss, float* grad_table, float* grad_feat,
                                          void* workspace, int64_t workspace_bytes, lec_stream_t stream) {
  LEC_CHECK_ARG(table_f16, "joint_loss_f16: table null");
  return joint_loss_impl(energy, label_proj, image_proj, nullptr, table_f16, ld_table, n_labels, feat, ld_feat, n_feat, pos_from, pos_to, neg, weights,
                         B, K, D, K_cone, alpha, e_pos, e_neg, loss, grad_table, grad_feat, workspace, workspace_bytes, stream);
}

// The loss of a step whose CNN rows exist CHUNK BY CHUNK (config 5: 7 424 rows per step do not fit one pass): one launch per chunk, right behind the chunk's
// forward, evaluates the pairs whose image row lies in [row_lo, row_hi) -- and the label-label pairs when labels_too != 0 (pass it with exactly one chunk).
// Over the launches of a step every pair is evaluated once: e_pos / e_neg fill up, the loss values add up (one scalar per launch), gradients add into
// grad_table / grad_feat as always.  feat rows outside the window are never read into a result.  table_f16 != NULL: the label rows are read from the
// fp16 shadow (as lec_joint_loss_fwd_bwd_f16); else from `table`.  window_dev != NULL: {row_lo, row_hi, labels_too, feat_base} are read from DEVICE memory at
// kernel start instead (the arguments are ignored) and feat / grad_feat are CHUNK buffers whose row 0 is feature row feat_base: the launch can be captured once
// into a hipGraph and replayed for every chunk of every step, the host only rewrites four integers.
extern "C" int lec_joint_loss_fwd_bwd_window(int energy, int label_proj, int image_proj,
                                             const float* table, const void* table_f16, int64_t ld_table, int n_labels,
                                             const float* feat, int64_t ld_feat, int n_feat,
                                             const int32_t* pos_from, const int32_t* pos_to, const int32_t* neg,
                                             const float* weights, int B, int K, int D, float K_cone, float alpha,
                                             int row_lo, int row_hi, int labels_too, const int32_t* window_dev,
                                             float* e_pos, float* e_neg, float* loss, float* grad_table, float* grad_feat,
                                             void* workspace, int64_t workspace_bytes, lec_stream_t stream) {
  LEC_CHECK_ARG(table || table_f16, "joint_loss_window: table null");
  return joint_loss_impl(energy, label_proj, image_proj, table_f16 ? nullptr : table, table_f16, ld_table, n_labels, feat, ld_feat, n_feat, pos_from, pos_to, neg,
                         weights, B, K, D, K_cone, alpha, e_pos, e_neg, loss, grad_table, grad_feat, workspace, workspace_bytes, stream, row_lo, row_hi, labels_too, window_dev);
}
