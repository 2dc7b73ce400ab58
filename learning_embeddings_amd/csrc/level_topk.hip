// Fused all-pairs scoring + per-level top-k (lec_level_topk): for every point (image) i and every level l of the label
// hierarchy, the k apexes (labels) j in [level_start[l], level_start[l+1]) with the smallest energy E(x_j, y_i).
//
// Replaces the core of calculate_classification_metrics (oe_h.py:2018-2036): the reference scores ONE image against all
// labels per python iteration (E_operator on a [1, N, D] repeat) and calls torch.topk(k=5, largest=False) per level.
// Here the M x N energy matrix is never written: one lane owns one image (its row lives in registers or in a
// lane-private LDS column), walks labels of a level -- the label row is wave-uniform, so it comes through the scalar
// cache -- and keeps its k best in registers: no cross-lane traffic.  A block is S waves that own the SAME 64 images
// and split the level's labels S ways (wave w takes j0 + w, j0 + w + S, ...) so that long levels fill all four SIMDs of
// a CU; their S sorted lists meet once, in LDS, at the end.
//
// Roofline: compute-bound on the vector ALUs, not HBM: algorithmic bytes are (N + M) * D * 4 in + M * L * k * 8 out
// (a few MB), against ~(4 D + 60) flops per (image, label) pair.
#include "lec_common.h"

namespace lec {

constexpr int kTopKMax = 8;
constexpr int kMaxLevels = 32;
struct LevelStarts { int32_t v[kMaxLevels + 1]; };       // by value in the kernel arguments (scalar registers)

template <int KK>
struct TopList {
  float v[KK];
  int id[KK];
  __device__ __forceinline__ void reset() {
#pragma unroll
    for (int i = 0; i < KK; ++i) { v[i] = __builtin_inff(); id[i] = -1; }
  }
  // keep ascending order; strict '<' so that among equal energies the lowest label index stays first.  NaN never enters
  // (torch.topk(largest=False) ranks NaN last).
  // merge form: ties go to the lower label index whatever the arrival order
  __device__ __forceinline__ void push_tie(float e, int j) {
    if (e < v[KK - 1] || (e == v[KK - 1] && j >= 0 && j < id[KK - 1])) {
      v[KK - 1] = e; id[KK - 1] = j;
#pragma unroll
      for (int i = KK - 1; i > 0; --i) {
        const bool sw = v[i] < v[i - 1] || (v[i] == v[i - 1] && id[i] < id[i - 1]);
        const float tv = sw ? v[i - 1] : v[i]; const int ti = sw ? id[i - 1] : id[i];
        v[i - 1] = sw ? v[i] : v[i - 1]; id[i - 1] = sw ? id[i] : id[i - 1];
        v[i] = tv; id[i] = ti;
      }
    }
  }
  __device__ __forceinline__ void push(float e, int j) {
    if (e < v[KK - 1]) {
      v[KK - 1] = e; id[KK - 1] = j;
#pragma unroll
      for (int i = KK - 1; i > 0; --i) {
        const bool sw = v[i] < v[i - 1];
        const float tv = sw ? v[i - 1] : v[i]; const int ti = sw ? id[i - 1] : id[i];
        v[i - 1] = sw ? v[i] : v[i - 1]; id[i - 1] = sw ? id[i] : id[i - 1];
        v[i] = tv; id[i] = ti;
      }
    }
  }
};

// DR > 0: the image row sits in DR registers (D <= DR; EXACT: D == DR, so the wave-uniform label-row loads carry no bound
// checks and merge into s_load_dwordx2/x4/x8).  DR == 0: any D, the row sits in LDS, transposed so that lane `t` reads
// column t (consecutive lanes, consecutive banks).
template <int ENERGY, int DR, int KK, bool EXACT>
__global__ __launch_bounds__(1024) void level_topk_kernel(const float* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ y, int64_t ldy, int64_t M, int D,
                                                        const LevelStarts level_start, int L, int k, float K,
                                                        int32_t* __restrict__ out_idx, float* __restrict__ out_val) {
  extern __shared__ float smem[];                        // [S-1][KK][2][64] merge area, then (DR == 0) ys [D][64]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, S = blockDim.x >> 6;
  float* ys = smem + (size_t)(S - 1) * KK * 2 * 64;
  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  const bool valid = i < M;
  const float* yrow = y + (valid ? i : 0) * ldy;
  float yr[DR > 0 ? DR : 1];
  float yy = 0.0f;
  if (DR > 0) {
#pragma unroll
    for (int d = 0; d < DR; ++d) { yr[d] = (EXACT || d < D) ? yrow[d] : 0.0f; yy += yr[d] * yr[d]; }
  } else {
    for (int d = 0; d < D; ++d) { const float b = yrow[d]; if (w == 0) ys[d * 64 + lane] = b; yy += b * b; }
    __syncthreads();
  }
  {
    const int l = blockIdx.y;                                          // one level per block row: L x more waves
    const int j0 = level_start.v[l], j1 = level_start.v[l + 1];           // wave-uniform
    TopList<KK> top; top.reset();
    for (int j = j0 + w; j < j1; j += S) {
      const float* xr = x + (int64_t)j * ldx;                          // wave-uniform address: scalar loads
      float xx = 0.f, s = 0.f, dd = 0.f;
      if (DR > 0) {
#pragma unroll
        for (int d = 0; d < DR; ++d) {
          const float a = (EXACT || d < D) ? xr[d] : 0.0f, b = yr[d], df = a - b;
          if (ENERGY == LEC_ENERGY_HYP_CONE) { xx += a * a; s += a * b; dd += df * df; }
          else if (ENERGY == LEC_ENERGY_EUC_CONE) { xx += a * a; s -= a * df; dd += df * df; }
          else { const float m = fmaxf(df, 0.0f); s += m * m; }
        }
      } else {
        for (int d = 0; d < D; ++d) {
          const float a = xr[d], b = ys[d * 64 + lane], df = a - b;
          if (ENERGY == LEC_ENERGY_HYP_CONE) { xx += a * a; s += a * b; dd += df * df; }
          else if (ENERGY == LEC_ENERGY_EUC_CONE) { xx += a * a; s -= a * df; dd += df * df; }
          else { const float m = fmaxf(df, 0.0f); s += m * m; }
        }
      }
      const float e = ENERGY == LEC_ENERGY_HYP_CONE ? cone_eval<false>(xx, yy, s, dd, K).E
                    : ENERGY == LEC_ENERGY_EUC_CONE ? euc_cone_eval<false>(xx, dd, s, K).E : s;
      top.push(e, j);
    }
    if (S > 1) {                                                       // S sorted lists per image -> wave 0
      if (w > 0) {
        float* dst = smem + (size_t)(w - 1) * KK * 2 * 64;
#pragma unroll
        for (int q = 0; q < KK; ++q) { dst[(q * 2) * 64 + lane] = top.v[q]; dst[(q * 2 + 1) * 64 + lane] = __int_as_float(top.id[q]); }
      }
      __syncthreads();
      if (w == 0) {
        for (int o = 0; o < S - 1; ++o) {
          const float* src = smem + (size_t)o * KK * 2 * 64;
#pragma unroll
          for (int q = 0; q < KK; ++q) top.push_tie(src[(q * 2) * 64 + lane], __float_as_int(src[(q * 2 + 1) * 64 + lane]));
        }
      }
    }
    if (valid && w == 0) {
#pragma unroll
      for (int q = 0; q < KK; ++q) {
        if (q < k) {
          out_idx[(i * L + l) * k + q] = top.id[q];
          out_val[(i * L + l) * k + q] = top.v[q];
        }
      }
    }
  }
}

template <int ENERGY>
static int launch_topk(const float* x, int64_t ldx, const float* y, int64_t ldy, int64_t M, int D,
                       const LevelStarts& level_start, int L, int k, float K, int32_t* out_idx, float* out_val,
                       int max_level, hipStream_t st) {
  const int nblocks = (int)((M + 63) / 64);
  // label split S: enough waves for ~2 per SIMD chip-wide on the longest level, at least ~32 labels per wave
  int S = (int)((2048 + nblocks - 1) / nblocks);
  if (S > max_level / 32) S = max_level / 32;
  if (S > 16) S = 16;
  const size_t ys_bytes = D > 16 ? (size_t)D * 64 * sizeof(float) : 0;
  const int s_cap = 1 + (int)((64 * 1024 - ys_bytes) / (kTopKMax * 2 * 64 * sizeof(float)));   // merge area + ys <= 64 KB of LDS
  if (S > s_cap) S = s_cap;
  if (S < 1) S = 1;
#define LEC_TK(DR_, KK_, EX_, SM_) hipLaunchKernelGGL((level_topk_kernel<ENERGY, DR_, KK_, EX_>), dim3(nblocks, L), dim3(64 * S), \
                                                      (SM_) + (size_t)(S - 1) * KK_ * 2 * 64 * sizeof(float), st, \
                                                      x, ldx, y, ldy, M, D, level_start, L, k, K, out_idx, out_val)
#define LEC_TKK(DR_, EX_, SM_) do { if (k <= 1) LEC_TK(DR_, 1, EX_, SM_); else if (k <= 5) LEC_TK(DR_, 5, EX_, SM_); else LEC_TK(DR_, 8, EX_, SM_); } while (0)
  if (D == 2) LEC_TKK(2, true, 0);
  else if (D == 4) LEC_TKK(4, true, 0);
  else if (D == 8) LEC_TKK(8, true, 0);
  else if (D == 10) LEC_TKK(10, true, 0);                  // the reference's default embedding_dim (oe_h.py:2411)
  else if (D == 16) LEC_TKK(16, true, 0);
  else if (D < 16) LEC_TKK(16, false, 0);
  else LEC_TKK(0, false, (size_t)D * 64 * sizeof(float));
#undef LEC_TKK
#undef LEC_TK
  LEC_CHECK_LAUNCH("level_topk_kernel");
  return LEC_OK;
}

}  // namespace lec

extern "C" int lec_level_topk(int energy, const float* x, int64_t ldx, int64_t N, const float* y, int64_t ldy, int64_t M,
                              int D, const int32_t* level_start, int L, int k, float K_cone, int32_t* out_idx,
                              float* out_val, lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(energy >= LEC_ENERGY_HYP_CONE && energy <= LEC_ENERGY_EUC_CONE, "level_topk: unknown energy %d", energy);
  LEC_CHECK_ARG(N >= 0 && M >= 0 && D > 0 && ldx >= D && ldy >= D, "level_topk: bad sizes");
  LEC_CHECK_ARG(L > 0 && L <= kMaxLevels && level_start, "level_topk: L=%d outside 1..%d or level_start null", L, kMaxLevels);
  LEC_CHECK_ARG(k >= 1 && k <= kTopKMax, "level_topk: k=%d outside 1..%d", k, kTopKMax);
  LEC_CHECK_ARG(D <= 224, "level_topk: embedding_dim %d too large for the LDS-resident image rows (max 224)", D);
  LevelStarts ls;
  int max_level = 0;
  for (int l = 0; l <= L; ++l) {
    ls.v[l] = level_start[l];
    LEC_CHECK_ARG(level_start[l] >= 0 && level_start[l] <= N && (l == 0 || level_start[l] >= level_start[l - 1]),
                  "level_topk: level_start must be non-decreasing offsets into the %lld apex rows", (long long)N);
    if (l > 0 && level_start[l] - level_start[l - 1] > max_level) max_level = level_start[l] - level_start[l - 1];
  }
  if (M == 0) return LEC_OK;
  LEC_CHECK_ARG(x && y && out_idx && out_val, "level_topk: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (energy == LEC_ENERGY_HYP_CONE) return launch_topk<LEC_ENERGY_HYP_CONE>(x, ldx, y, ldy, M, D, ls, L, k, K_cone, out_idx, out_val, max_level, st);
  if (energy == LEC_ENERGY_EUC_CONE) return launch_topk<LEC_ENERGY_EUC_CONE>(x, ldx, y, ldy, M, D, ls, L, k, K_cone, out_idx, out_val, max_level, st);
  return launch_topk<LEC_ENERGY_ORDER>(x, ldx, y, ldy, M, D, ls, L, k, K_cone, out_idx, out_val, max_level, st);
}
