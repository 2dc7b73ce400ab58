// Pair energies on dense rows: E_operator and its autograd (oe_h.py:811-833, order_embeddings.py:818-824), and the
// all-pairs score matrix used by calculate_classification_metrics (oe_h.py:2018-2036).
// T lanes cooperate on one pair (elements d = t, t+T, ...); row statistics are reduced by xor-butterflies.
// HBM-bound: fwd reads 2*D*4 B and writes 4 B per pair; bwd reads 2*D*4+4 and writes 2*D*4.
#include "lec_common.h"

namespace lec {

template <int T, int ENERGY>
__global__ __launch_bounds__(256) void pair_energy_fwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                              const float* __restrict__ y, int64_t ldy, int64_t P,
                                                              int D, float K, float* __restrict__ E) {
  constexpr int PPW = kWave / T;
  const int lane = threadIdx.x & 63, t = lane % T, slot = lane / T;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwave = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t base = wave * PPW; base < P; base += nwave * PPW) {       // wave-uniform loop bound
    const int64_t p = base + slot;
    const bool valid = p < P;
    const float* xr = x + (valid ? p : 0) * ldx;
    const float* yr = y + (valid ? p : 0) * ldy;
    float xx = 0.f, yy = 0.f, s = 0.f, dd = 0.f;
    for (int d = t; d < D; d += T) {
      float a = xr[d], b = yr[d], df = a - b;
      if (ENERGY == LEC_ENERGY_HYP_CONE) { xx += a * a; yy += b * b; s += a * b; dd += df * df; }
      else if (ENERGY == LEC_ENERGY_EUC_CONE) { xx += a * a; s -= a * df; dd += df * df; }        // s = <x, y-x>
      else { float m = fmaxf(df, 0.0f); xx += m * m; }
    }
    float e;
    if (ENERGY == LEC_ENERGY_HYP_CONE) {
      xx = group_sum<T>(xx); yy = group_sum<T>(yy); s = group_sum<T>(s); dd = group_sum<T>(dd);
      e = cone_eval<false>(xx, yy, s, dd, K).E;
    } else if (ENERGY == LEC_ENERGY_EUC_CONE) {
      xx = group_sum<T>(xx); s = group_sum<T>(s); dd = group_sum<T>(dd);
      e = euc_cone_eval<false>(xx, dd, s, K).E;
    } else {
      e = group_sum<T>(xx);
    }
    if (valid && t == 0) E[p] = e;
  }
}

template <int T, int ENERGY>
__global__ __launch_bounds__(256) void pair_energy_bwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                              const float* __restrict__ y, int64_t ldy,
                                                              const float* __restrict__ gE, int64_t P, int D, float K,
                                                              float* __restrict__ gx, float* __restrict__ gy,
                                                              int64_t ldg) {
  constexpr int PPW = kWave / T;
  const int lane = threadIdx.x & 63, t = lane % T, slot = lane / T;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwave = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t base = wave * PPW; base < P; base += nwave * PPW) {
    const int64_t p = base + slot;
    const bool valid = p < P;
    const float* xr = x + (valid ? p : 0) * ldx;
    const float* yr = y + (valid ? p : 0) * ldy;
    const float g = valid ? gE[p] : 0.0f;
    if (ENERGY == LEC_ENERGY_HYP_CONE || ENERGY == LEC_ENERGY_EUC_CONE) {
      float xx = 0.f, yy = 0.f, s = 0.f, dd = 0.f;
      for (int d = t; d < D; d += T) {
        float a = xr[d], b = yr[d], df = a - b;
        xx += a * a; dd += df * df;
        if (ENERGY == LEC_ENERGY_HYP_CONE) { yy += b * b; s += a * b; } else s -= a * df;
      }
      xx = group_sum<T>(xx); s = group_sum<T>(s); dd = group_sum<T>(dd);
      if (ENERGY == LEC_ENERGY_HYP_CONE) yy = group_sum<T>(yy);
      ConeEval ev = ENERGY == LEC_ENERGY_HYP_CONE ? cone_eval<true>(xx, yy, s, dd, K) : euc_cone_eval<true>(xx, dd, s, K);
      if (valid) {
        for (int d = t; d < D; d += T) {
          float a = xr[d], b = yr[d];
          gx[p * ldg + d] = g * (ev.cxx * a + ev.cxy * b);
          gy[p * ldg + d] = g * (ev.cxy * a + ev.cyy * b);
        }
      }
    } else if (valid) {
      for (int d = t; d < D; d += T) {
        float m = 2.0f * fmaxf(xr[d] - yr[d], 0.0f) * g;
        gx[p * ldg + d] = m; gy[p * ldg + d] = -m;
      }
    }
  }
}

// E[i, j] = E(x_j, y_i).  One block scores a tile of TI images against all apexes j streamed through LDS in tiles of
// TJ rows; each lane owns one image row in registers-by-LDS and walks the apex tile.  Output rows are written
// coalesced (lanes = consecutive j).
template <int ENERGY>
__global__ __launch_bounds__(256) void pair_energy_matrix_kernel(const float* __restrict__ x, int64_t ldx, int64_t N,
                                                                 const float* __restrict__ y, int64_t ldy, int64_t M,
                                                                 int D, float K, float* __restrict__ E, int64_t ldE,
                                                                 int TI) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Dp = D | 1;                                  // odd row stride: conflict-free column walks
  float* ys = smem;                                      // [TI][Dp]
  float* xs = smem + (int64_t)TI * Dp;                   // [256][Dp]
  float* ystat = xs + 256 * Dp;                          // [TI] |y|^2
  const int64_t i0 = (int64_t)blockIdx.x * TI;
  for (int e = threadIdx.x; e < TI * D; e += blockDim.x) {
    int r = e / D, d = e - r * D;
    ys[r * Dp + d] = (i0 + r < M) ? y[(i0 + r) * ldy + d] : 0.0f;
  }
  __syncthreads();
  for (int r = threadIdx.x; r < TI; r += blockDim.x) {
    float a = 0.0f;
    for (int d = 0; d < D; ++d) a += ys[r * Dp + d] * ys[r * Dp + d];
    ystat[r] = a;
  }
  for (int64_t j0 = 0; j0 < N; j0 += 256) {
    __syncthreads();
    for (int e = threadIdx.x; e < 256 * D; e += blockDim.x) {
      int r = e / D, d = e - r * D;
      xs[r * Dp + d] = (j0 + r < N) ? x[(j0 + r) * ldx + d] : 0.0f;
    }
    __syncthreads();
    const int64_t j = j0 + threadIdx.x;
    const float* xr = xs + threadIdx.x * Dp;
    float xx = 0.0f;
    for (int d = 0; d < D; ++d) xx += xr[d] * xr[d];
    for (int r = 0; r < TI && i0 + r < M; ++r) {
      const float* yr = ys + r * Dp;                     // broadcast reads
      float s = 0.f, dd = 0.f;
      for (int d = 0; d < D; ++d) {
        float a = xr[d], b = yr[d], df = a - b;
        if (ENERGY == LEC_ENERGY_HYP_CONE) { s += a * b; dd += df * df; }
        else if (ENERGY == LEC_ENERGY_EUC_CONE) { s -= a * df; dd += df * df; }
        else { float m = fmaxf(df, 0.0f); s += m * m; }
      }
      float e = ENERGY == LEC_ENERGY_HYP_CONE ? cone_eval<false>(xx, ystat[r], s, dd, K).E
              : ENERGY == LEC_ENERGY_EUC_CONE ? euc_cone_eval<false>(xx, dd, s, K).E : s;
      if (j < N) E[(i0 + r) * ldE + j] = e;
    }
  }
}

static int pick_T(int D) { return D <= 4 ? 1 : (D <= 16 ? 4 : (D <= 64 ? 16 : 64)); }

}  // namespace lec

extern "C" int lec_pair_energy_fwd(int energy, const float* x, int64_t ldx, const float* y, int64_t ldy, int64_t P,
                                   int D, float K_cone, float* E, lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(energy >= LEC_ENERGY_HYP_CONE && energy <= LEC_ENERGY_EUC_CONE, "pair_energy_fwd: unknown energy %d", energy);
  LEC_CHECK_ARG(P >= 0 && D > 0 && ldx >= D && ldy >= D, "pair_energy_fwd: bad sizes P=%lld D=%d", (long long)P, D);
  if (P == 0) return LEC_OK;
  LEC_CHECK_ARG(x && y && E, "pair_energy_fwd: null pointer");
  const int T = pick_T(D);
  int64_t waves = (P + (64 / T) - 1) / (64 / T);
  int nblocks = (int)((waves + 3) / 4 > 4096 ? 4096 : (waves + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
#define L(T_) do { if (energy == LEC_ENERGY_HYP_CONE) hipLaunchKernelGGL((pair_energy_fwd_kernel<T_, LEC_ENERGY_HYP_CONE>), dim3(nblocks), dim3(256), 0, st, x, ldx, y, ldy, P, D, K_cone, E); \
                   else if (energy == LEC_ENERGY_EUC_CONE) hipLaunchKernelGGL((pair_energy_fwd_kernel<T_, LEC_ENERGY_EUC_CONE>), dim3(nblocks), dim3(256), 0, st, x, ldx, y, ldy, P, D, K_cone, E); \
                   else hipLaunchKernelGGL((pair_energy_fwd_kernel<T_, LEC_ENERGY_ORDER>), dim3(nblocks), dim3(256), 0, st, x, ldx, y, ldy, P, D, K_cone, E); } while (0)
  if (T == 1) L(1); else if (T == 4) L(4); else if (T == 16) L(16); else L(64);
#undef L
  LEC_CHECK_LAUNCH("pair_energy_fwd_kernel");
  return LEC_OK;
}

extern "C" int lec_pair_energy_bwd(int energy, const float* x, int64_t ldx, const float* y, int64_t ldy,
                                   const float* gE, int64_t P, int D, float K_cone, float* gx, float* gy, int64_t ldg,
                                   lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(energy >= LEC_ENERGY_HYP_CONE && energy <= LEC_ENERGY_EUC_CONE, "pair_energy_bwd: unknown energy %d", energy);
  LEC_CHECK_ARG(P >= 0 && D > 0 && ldx >= D && ldy >= D && ldg >= D, "pair_energy_bwd: bad sizes");
  if (P == 0) return LEC_OK;
  LEC_CHECK_ARG(x && y && gE && gx && gy, "pair_energy_bwd: null pointer");
  const int T = pick_T(D);
  int64_t waves = (P + (64 / T) - 1) / (64 / T);
  int nblocks = (int)((waves + 3) / 4 > 4096 ? 4096 : (waves + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
#define L(T_) do { if (energy == LEC_ENERGY_HYP_CONE) hipLaunchKernelGGL((pair_energy_bwd_kernel<T_, LEC_ENERGY_HYP_CONE>), dim3(nblocks), dim3(256), 0, st, x, ldx, y, ldy, gE, P, D, K_cone, gx, gy, ldg); \
                   else if (energy == LEC_ENERGY_EUC_CONE) hipLaunchKernelGGL((pair_energy_bwd_kernel<T_, LEC_ENERGY_EUC_CONE>), dim3(nblocks), dim3(256), 0, st, x, ldx, y, ldy, gE, P, D, K_cone, gx, gy, ldg); \
                   else hipLaunchKernelGGL((pair_energy_bwd_kernel<T_, LEC_ENERGY_ORDER>), dim3(nblocks), dim3(256), 0, st, x, ldx, y, ldy, gE, P, D, K_cone, gx, gy, ldg); } while (0)
  if (T == 1) L(1); else if (T == 4) L(4); else if (T == 16) L(16); else L(64);
#undef L
  LEC_CHECK_LAUNCH("pair_energy_bwd_kernel");
  return LEC_OK;
}

extern "C" int lec_pair_energy_matrix(int energy, const float* x, int64_t ldx, int64_t N, const float* y, int64_t ldy,
                                      int64_t M, int D, float K_cone, float* E, int64_t ldE, lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(energy >= LEC_ENERGY_HYP_CONE && energy <= LEC_ENERGY_EUC_CONE, "pair_energy_matrix: unknown energy %d", energy);
  LEC_CHECK_ARG(N >= 0 && M >= 0 && D > 0 && ldx >= D && ldy >= D && ldE >= N, "pair_energy_matrix: bad sizes");
  if (N == 0 || M == 0) return LEC_OK;
  LEC_CHECK_ARG(x && y && E, "pair_energy_matrix: null pointer");
  const int Dp = D | 1;
  int TI = 32;
  while (TI > 1 && (int64_t)(TI + 256) * Dp * 4 + TI * 4 > 64 * 1024) TI >>= 1;
  const int64_t smem = (int64_t)(TI + 256) * Dp * 4 + TI * 4;
  LEC_CHECK_ARG(smem <= 160 * 1024, "pair_energy_matrix: embedding_dim %d too large for the LDS tile", D);
  const int nblocks = (int)((M + TI - 1) / TI);
  hipStream_t st = (hipStream_t)stream;
  if (energy == LEC_ENERGY_HYP_CONE)
    hipLaunchKernelGGL((pair_energy_matrix_kernel<LEC_ENERGY_HYP_CONE>), dim3(nblocks), dim3(256), smem, st, x, ldx, N, y, ldy, M, D, K_cone, E, ldE, TI);
  else if (energy == LEC_ENERGY_EUC_CONE)
    hipLaunchKernelGGL((pair_energy_matrix_kernel<LEC_ENERGY_EUC_CONE>), dim3(nblocks), dim3(256), smem, st, x, ldx, N, y, ldy, M, D, K_cone, E, ldE, TI);
  else
    hipLaunchKernelGGL((pair_energy_matrix_kernel<LEC_ENERGY_ORDER>), dim3(nblocks), dim3(256), smem, st, x, ldx, N, y, ldy, M, D, K_cone, E, ldE, TI);
  LEC_CHECK_LAUNCH("pair_energy_matrix_kernel");
  return LEC_OK;
}
