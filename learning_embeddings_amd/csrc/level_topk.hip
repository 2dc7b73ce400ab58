// Fused all-pairs scoring + per-level top-k (lec_level_topk): for every point (image) i and every level l of the label
// hierarchy, the k apexes (labels) j in [level_start[l], level_start[l+1]) with the smallest energy E(x_j, y_i).
//
// Replaces the core of calculate_classification_metrics (oe_h.py:2018-2036): the reference scores ONE image against all
// labels per python iteration (E_operator on a [1, N, D] repeat) and calls torch.topk(k=5, largest=False) per level.
// Here the M x N energy matrix is never written: one lane owns one image (its row lives in registers or in a
// lane-private LDS column), walks the labels of a level -- the label row is wave-uniform, so it comes through the
// scalar cache / a broadcast -- and keeps its k best in registers.  No cross-lane traffic at all.
//
// Roofline: compute-bound on the vector ALUs, not HBM: algorithmic bytes are (N + M) * D * 4 in + M * L * k * 8 out
// (a few MB), against ~(4 D + 60) flops per (image, label) pair.
#include "lec_common.h"

namespace lec {

constexpr int kTopKMax = 8;

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
__global__ __launch_bounds__(64) void level_topk_kernel(const float* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ y, int64_t ldy, int64_t M, int D,
                                                        const int32_t* __restrict__ level_start, int L, int k, float K,
                                                        int32_t* __restrict__ out_idx, float* __restrict__ out_val) {
  extern __shared__ float ys[];                          // DR == 0: [D][64]
  const int lane = threadIdx.x;
  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  const bool valid = i < M;
  const float* yrow = y + (valid ? i : 0) * ldy;
  float yr[DR > 0 ? DR : 1];
  float yy = 0.0f;
  if (DR > 0) {
#pragma unroll
    for (int d = 0; d < DR; ++d) { yr[d] = (EXACT || d < D) ? yrow[d] : 0.0f; yy += yr[d] * yr[d]; }
  } else {
    for (int d = 0; d < D; ++d) { const float b = yrow[d]; ys[d * 64 + lane] = b; yy += b * b; }
  }
  {
    const int l = blockIdx.y;                                          // one level per block row: L x more waves
    const int j0 = level_start[l], j1 = level_start[l + 1];           // wave-uniform
    TopList<KK> top; top.reset();
    for (int j = j0; j < j1; ++j) {
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
    if (valid) {
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
                       const int32_t* level_start, int L, int k, float K, int32_t* out_idx, float* out_val,
                       hipStream_t st) {
  const int nblocks = (int)((M + 63) / 64);
#define LEC_TK(DR_, KK_, EX_, SM_) hipLaunchKernelGGL((level_topk_kernel<ENERGY, DR_, KK_, EX_>), dim3(nblocks, L), dim3(64), SM_, st, \
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
  LEC_CHECK_ARG(N >= 0 && M >= 0 && D > 0 && ldx >= D && ldy >= D && L > 0 && L <= 65535, "level_topk: bad sizes");
  LEC_CHECK_ARG(k >= 1 && k <= kTopKMax, "level_topk: k=%d outside 1..%d", k, kTopKMax);
  LEC_CHECK_ARG(D <= 256, "level_topk: embedding_dim %d too large for the LDS-resident image rows (max 256)", D);
  if (M == 0) return LEC_OK;
  LEC_CHECK_ARG(x && y && level_start && out_idx && out_val, "level_topk: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (energy == LEC_ENERGY_HYP_CONE) return launch_topk<LEC_ENERGY_HYP_CONE>(x, ldx, y, ldy, M, D, level_start, L, k, K_cone, out_idx, out_val, st);
  if (energy == LEC_ENERGY_EUC_CONE) return launch_topk<LEC_ENERGY_EUC_CONE>(x, ldx, y, ldy, M, D, level_start, L, k, K_cone, out_idx, out_val, st);
  return launch_topk<LEC_ENERGY_ORDER>(x, ldx, y, ldy, M, D, level_start, L, k, K_cone, out_idx, out_val, st);
}
