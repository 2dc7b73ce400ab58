// Fused BatchNorm (+ residual add) (+ ReLU) for NHWC bf16 activations: the HBM-bound half of the ResNet backbone.
//
// In the reference these are torchvision ResNet's `bn(conv(x))`, `relu(...)`, `out += identity` inside FeatCNN18's
// backbone (oe_h.py:311,317 -> torchvision BasicBlock/Bottleneck.forward): 3-4 separate framework kernels per layer in
// each direction.  The rocprof trace of the first end-to-end step (profiles/r01_bench_cfg3_steady_state.md) put
// batchnorm at 37 % and relu/add at a further ~17 % of the step, all streaming work running at 2.4-3.4 TB/s.
// Here a layer is   stats pass (read x) -> apply pass (read x [+ residual], write y = relu(x*scale+shift [+ r]))
// and backward is   reduce pass (read dy, y, x) -> apply pass (read dy, y, x; write dx [+ d residual]).
//
// Layout: x is [M, C] with C innermost (NHWC, M = N*H*W), bf16; C % 8 == 0.  One thread owns 8 consecutive channels
// (one 16-byte load); a 256-thread block covers 256/(C/8) rows per sweep and strides the row range with several
// independent 16-byte loads in flight.  Per-channel reductions: registers -> LDS across the block's row groups ->
// one fp32 partial per block and channel -> finalize kernel (double accumulation over <= 1024 partials).
// Roofline: HBM. Algorithmic bytes per element (bf16): fwd 2+2+2 (+2 residual); bwd 6 (reduce) + 6+2 (+2 d residual).
#include <hip/hip_bf16.h>
#include <cstdlib>
#include "lec_common.h"
#include "tuning.h"

namespace lec {

struct alignas(16) bf16x8 { unsigned short v[8]; };
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// 16-byte streaming accesses, non-temporal (measured: plain accesses cost the bench step 2.4 %): activations are touched once per pass and are far
// larger than L2 / Infinity Cache; keeping them out of the caches leaves room for the small per-channel vectors.
__device__ __forceinline__ u32x4 ld16(const void* p) { return __builtin_nontemporal_load((const u32x4*)p); }
__device__ __forceinline__ void st16(void* p, u32x4 r) { __builtin_nontemporal_store(r, (u32x4*)p); }

__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float(((unsigned int)u) << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {
  __hip_bfloat16 h = __float2bfloat16(f);                 // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
  return *reinterpret_cast<unsigned short*>(&h);
}

// Element types of the activations.  A "vector" is 8 consecutive channels of one row: 16 bytes of bf16 (the MI355X-native
// storage) or 32 bytes of fp32 (the reference's precision, oe_h.py:281-328: no AMP anywhere).  Kernels hold a vector as
// float[8]; `rnd` is the storage rounding (bf16: round-to-nearest-even and back; fp32: identity), applied wherever the
// bf16 kernels reuse a value they have just stored so that both passes of a layer see the same numbers.
struct EBf16 {
  static constexpr int kBytes = 2;
  static __device__ __forceinline__ int chan(int cv, int j, int) { return cv * 8 + j; }      // channel of a thread's j-th value
  static __device__ __forceinline__ void ld(const void* p, int64_t i, float (&f)[8], int = 0, int = 0) {
    u32x4 r = ld16((const char*)p + i * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) { f[2 * q] = __uint_as_float(r[q] << 16); f[2 * q + 1] = __uint_as_float(r[q] & 0xffff0000u); }
  }
  static __device__ __forceinline__ void st(void* p, int64_t i, const float (&f)[8], int = 0, int = 0) {
    u32x4 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = (unsigned int)f2bf(f[2 * q]) | ((unsigned int)f2bf(f[2 * q + 1]) << 16);
    st16((char*)p + i * 16, r);
  }
  static __device__ __forceinline__ float rnd(float v) { return bf2f(f2bf(v)); }
};
// fp32: a thread's 8 channels are TWO runs of 4 -- channels 4 cv .. 4 cv + 3 and C / 2 + 4 cv .. + 3 -- so that each of its two 16-byte
// accesses is, across the wave, one contiguous kilobyte.  (With 8 consecutive channels per thread both accesses stride 32 bytes from
// lane to lane and touch every cache line of a 2 KB span half: 4.8 against 5.3 TB/s on a 2-read / 1-write pass,
// tools/microbench/bn_access.hip.)  i = row * CV + cv as for bf16; the byte address of the first run is 32 i - 16 cv.
struct EF32 {
  static constexpr int kBytes = 4;
  static __device__ __forceinline__ int chan(int cv, int j, int CV) { return j < 4 ? cv * 4 + j : CV * 4 + cv * 4 + (j - 4); }
  static __device__ __forceinline__ void ld(const void* p, int64_t i, float (&f)[8], int cv, int CV) {
    const char* q0 = (const char*)p + i * 32 - cv * 16;
    u32x4 a = ld16(q0), b = ld16(q0 + CV * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) { f[q] = __uint_as_float(a[q]); f[4 + q] = __uint_as_float(b[q]); }
  }
  static __device__ __forceinline__ void st(void* p, int64_t i, const float (&f)[8], int cv, int CV) {
    u32x4 a, b;
#pragma unroll
    for (int q = 0; q < 4; ++q) { a[q] = __float_as_uint(f[q]); b[q] = __float_as_uint(f[4 + q]); }
    char* q0 = (char*)p + i * 32 - cv * 16;
    st16(q0, a); st16(q0 + CV * 16, b);
  }
  static __device__ __forceinline__ float rnd(float v) { return v; }
};

// (Register caps on the streaming kernels -- so that their waves fit beside the two 192 - 256-register convolution waves a SIMD of the OTHER pass stream holds --
// were measured in round 3: 64 / 96 registers, no spills, bench step 131.3 / 131.5 ms against 131.0 - 131.5 uncapped.  Removed.)
constexpr int kBnThreads = 256;
constexpr int kBnMaxBlocks = 512;           // blocks of a reduction pass of this file
constexpr int kBnMaxRows = 2048;            // partial rows the workspace holds (layout constant): a convolution's balanced form leaves one row per m-tile

// Reduction passes: a block owns a channel chunk of CVB 16-byte vectors (<= 64: up to 1 KiB contiguous per row) and a
// strided set of rows; grid = (nrb row blocks, NCH channel chunks), nrb * NCH <= 512 blocks (2 per CU, 8 loads in flight
// per thread).  Per channel there are nrb partials, reduced by a 1024-thread finalize kernel (32 channels x 32 splits).
// Apply passes: a block covers whole rows (CV vectors), up to 1536 blocks.  (2048 blocks of 4 waves are ALL 32 wave slots of all 256 CUs, held for the whole
// pass by grid-stride loops: the other pass's small dependent launches -- its statistics finalize -- then wait for the pass to END before a workgroup of
// theirs is placed (finalize launches of 230 us in the step's trace).  With 1536 blocks 8 slots per CU stay free.  MEASURED (round 4, same box, bench
// step, three alternating runs each, ms): 2048: 126.45, 125.87, 125.56; 1536: 125.59, 125.27, 125.47; 1280: 125.24, 125.90, 125.29; 1024: 125.90, 125.97, 125.62.)
static inline int bn_apply_cap() {                         // blocks of an apply pass (experiments: LEC_BN_BLOCKS)
  return tuning().bn_apply_blocks;
}
struct BnGeom { int CV, RPI, CVB, NCH, RPIB, nrb; };
static inline BnGeom bn_geom(int64_t M, int C) {
  BnGeom g; g.CV = C / 8; g.RPI = kBnThreads / g.CV;
  g.CVB = g.CV > 64 ? 64 : g.CV; g.NCH = g.CV / g.CVB; g.RPIB = kBnThreads / g.CVB;
  int64_t nb = (M + g.RPIB - 1) / g.RPIB;
  int64_t want = (nb + 7) / 8;                              // >= 8 sweeps per block
  int64_t cap = kBnMaxBlocks / g.NCH;
  g.nrb = (int)(want < 1 ? 1 : (want > cap ? cap : want));
  return g;
}

// block-level reduction of NV per-thread float[8] accumulators over the RPI row groups; result valid for threads
// of row group 0 (tid < CV).  smem must hold kBnThreads * 8 floats.
template <int NV>
__device__ __forceinline__ void block_reduce_rows(float (&acc)[NV][8], int CV, int RPI, float* smem) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int a = 0; a < NV; ++a) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) smem[j * kBnThreads + tid] = acc[a][j];       // [j][tid]: conflict-free
    __syncthreads();
    if (tid < CV) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float s = 0.0f;
        for (int r = 0; r < RPI; ++r) s += smem[j * kBnThreads + r * CV + tid];
        acc[a][j] = s;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// forward, pass 1: per-block partial sum / sum of squares per channel -> part[rb][2][C]
template <typename E>
__global__ __launch_bounds__(kBnThreads) void bn_stats_kernel(const void* __restrict__ x, int64_t M, int C, int CV, int CVB,
                                                              int RPIB, float* __restrict__ part) {
  __shared__ float smem[kBnThreads * 8];
  const int tid = threadIdx.x;
  const int cvl = tid % CVB, rg = tid / CVB;
  const int cv = blockIdx.y * CVB + cvl;
  const bool live = rg < RPIB;
  float acc[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { acc[0][j] = 0.f; acc[1][j] = 0.f; }
  const int64_t stride = (int64_t)gridDim.x * RPIB;
  constexpr int U = E::kBytes == 2 ? 8 : 4;                                  // 8 independent 16-byte loads in flight either way
  if (live) {
    int64_t r = (int64_t)blockIdx.x * RPIB + rg;
    for (; r + (U - 1) * stride < M; r += U * stride) {
      float v[U][8];
#pragma unroll
      for (int u = 0; u < U; ++u) E::ld(x, (r + u * stride) * CV + cv, v[u], cv, CV);
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { float f = v[u][j]; acc[0][j] += f; acc[1][j] += f * f; }
      }
    }
    for (; r < M; r += stride) {
      float a[8];
      E::ld(x, r * CV + cv, a, cv, CV);
#pragma unroll
      for (int j = 0; j < 8; ++j) { acc[0][j] += a[j]; acc[1][j] += a[j] * a[j]; }
    }
  }
  block_reduce_rows<2>(acc, CVB, RPIB, smem);
  if (tid < CVB) {
    float* p = part + (int64_t)blockIdx.x * 2 * C;
#pragma unroll
    for (int j = 0; j < 8; ++j) { p[E::chan(cv, j, CV)] = acc[0][j]; p[C + E::chan(cv, j, CV)] = acc[1][j]; }
  }
}

// sum of the nblk partials of two statistics: a 256-thread block owns 8 channels x 32 splits (double accumulation).
// Small blocks on purpose: these kernels are a few microseconds of latency-bound work that must find a free slot on a
// GPU whose CUs are full of weight-gradient workgroups from the second stream; a 1024-thread block waited 30-60 us
// for one CU to drain (rocprof, profiles/r01_*_final.md), a 4-wave block is placed at once.
constexpr int kFinCh = 8, kFinSplit = 32, kFinThreads = kFinCh * kFinSplit;
__device__ __forceinline__ void reduce_partials_256(const float* __restrict__ part, int nblk, int C, int c, int split,
                                                    double& s, double& q) {
  __shared__ double sh[2][kFinSplit][kFinCh + 1];
  double a = 0.0, b = 0.0;
  if (c < C) {
    // the partials were written by other CUs' blocks a moment ago (L2 / Infinity Cache round trips): issue a thread's
    // loads as one independent batch instead of a load -> add -> load chain
    for (int p0 = split; p0 < nblk; p0 += kFinSplit * 8) {
      float va[8], vb[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int p = p0 + kFinSplit * u;
        const bool ok = p < nblk;
        va[u] = ok ? part[(int64_t)p * 2 * C + c] : 0.0f;
        vb[u] = ok ? part[(int64_t)p * 2 * C + C + c] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { a += (double)va[u]; b += (double)vb[u]; }
    }
  }
  const int cl = threadIdx.x % kFinCh;
  sh[0][split][cl] = a; sh[1][split][cl] = b;
  __syncthreads();
  s = 0.0; q = 0.0;
  if (split == 0) {
    for (int k = 0; k < kFinSplit; ++k) { s += sh[0][k][cl]; q += sh[1][k][cl]; }
  }
}

// forward, pass 1b: reduce the partials (double), produce scale/shift, saved mean/invstd, running statistics
__global__ void bn_stats_finalize_kernel(const float* __restrict__ part, int nblk, int C, int64_t M,
                                         const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                         float momentum, float* running_mean, float* running_var,
                                         float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                         float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * kFinCh + threadIdx.x % kFinCh, split = threadIdx.x / kFinCh;
  double s, q;
  reduce_partials_256(part, nblk, C, c, split, s, q);
  if (split != 0 || c >= C) return;
  const double mean = s / (double)M;
  double var = q / (double)M - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  save_mean[c] = (float)mean; save_invstd[c] = invstd;
  const float sc = gamma[c] * invstd;
  scale[c] = sc; shift[c] = beta[c] - (float)mean * sc;
  if (running_mean) {
    const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
    running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// eval mode: scale/shift from the running statistics
__global__ void bn_eval_coeff_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                     const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                     float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] / sqrtf(running_var[c] + eps);
  scale[c] = sc; shift[c] = beta[c] - running_mean[c] * sc;
}

// forward, pass 2: y = [relu]( x * scale + shift [+ residual] )
// `mask` (optional, RELU only): one byte per thread-vector, bit j = [y_j > 0] -- backward reads it instead of y (1/16 the bytes)
template <typename E, bool RES, bool RELU>
__global__ __launch_bounds__(kBnThreads) void bn_apply_kernel(const void* __restrict__ x, const void* __restrict__ res,
                                                              int64_t M, int CV, int RPI, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, void* __restrict__ y,
                                                              unsigned char* __restrict__ mask) {
  const int tid = threadIdx.x;
  const int cv = tid % CV, rg = tid / CV;
  if (rg >= RPI) return;
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = scale[E::chan(cv, j, CV)]; sh[j] = shift[E::chan(cv, j, CV)]; }
  const int64_t stride = (int64_t)gridDim.x * RPI;
  auto one = [&](float (&a)[8], const float (&r)[8], int64_t idx) {
    unsigned int bits = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = a[j] * sc[j] + sh[j];
      if (RES) v += r[j];
      if (RELU) v = v > 0.0f ? v : 0.0f;
      a[j] = v;
      if (RELU) bits |= (E::rnd(v) > 0.0f ? 1u : 0u) << j;           // decided on the ROUNDED output, like a y-based mask
    }
    if (RELU && mask) mask[idx] = (unsigned char)bits;
    E::st(y, idx, a, cv, CV);
  };
  constexpr int U = E::kBytes == 2 ? 4 : 2;
  int64_t r = (int64_t)blockIdx.x * RPI + rg;
  for (; r + (U - 1) * stride < M; r += U * stride) {
    float a[U][8], rr[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) E::ld(x, (r + u * stride) * CV + cv, a[u], cv, CV);
    if (RES) {
#pragma unroll
      for (int u = 0; u < U; ++u) E::ld(res, (r + u * stride) * CV + cv, rr[u], cv, CV);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) one(a[u], rr[u], (r + u * stride) * CV + cv);
  }
  for (; r < M; r += stride) {
    const int64_t i0 = r * CV + cv;
    float a[8], rr[8];
    E::ld(x, i0, a, cv, CV);
    if (RES) E::ld(res, i0, rr, cv, CV);
    one(a, rr, i0);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// backward, pass 1: per-block partials of  dbeta = sum g,  dgamma = sum g * xhat,   g = dy * [y > 0]
// RELU: 0 none, 1 mask from the saved output y, 2 mask from the saved bitmask (y is then a byte array [M, C/8])
// gout (layers with a residual branch): the pass also WRITES g -- it is the gradient of the identity branch, and
// pass 2 then reads this one tensor instead of dy, dy2 and the mask again (8.1 -> 7.1 bytes-units per element on the
// forked block outputs).  The sums are taken over the rounded g so that both passes see the same values.
template <typename E, int RELU>
__global__ __launch_bounds__(kBnThreads) void bn_bwd_reduce_kernel(const void* __restrict__ dy, const void* __restrict__ dy2,
                                                                   const void* __restrict__ y,
                                                                   const void* __restrict__ x, int64_t M, int C, int CV,
                                                                   int CVB, int RPI, const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd, float* __restrict__ part,
                                                                   void* __restrict__ gout) {
  __shared__ float smem[kBnThreads * 8];
  const int tid = threadIdx.x;
  const int cvl = tid % CVB, rg = tid / CVB;
  const int cv = blockIdx.y * CVB + cvl;
  const bool live = rg < RPI;
  float acc[2][8], mu[8], is[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { acc[0][j] = 0.f; acc[1][j] = 0.f; mu[j] = mean[E::chan(cv, j, CV)]; is[j] = invstd[E::chan(cv, j, CV)]; }
  const int64_t stride = (int64_t)gridDim.x * RPI;
  auto one = [&](float (&g)[8], const float (&h)[8], const float (&yv)[8], unsigned int m, const float (&xv)[8], int64_t i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a = g[j];
      if (dy2) a += h[j];
      if (RELU == 1) a = yv[j] > 0.0f ? a : 0.0f;
      if (RELU == 2) a = (m >> j) & 1u ? a : 0.0f;
      if (gout) a = E::rnd(a);
      g[j] = a;
      acc[0][j] += a;
      acc[1][j] += a * ((xv[j] - mu[j]) * is[j]);
    }
    if (gout) E::st(gout, i, g, cv, CV);
  };
  if (live) {
    int64_t r = (int64_t)blockIdx.x * RPI + rg;
    constexpr int U = E::kBytes == 2 ? 2 : 1;
    for (; r + (U - 1) * stride < M; r += U * stride) {
      float g[U][8], xv[U][8], h[U][8], yv[U][8];
      unsigned int m[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = (r + u * stride) * CV + cv;
        E::ld(dy, i, g[u], cv, CV); E::ld(x, i, xv[u], cv, CV);
        if (dy2) E::ld(dy2, i, h[u], cv, CV);                                        // second gradient stream of a forked activation
        m[u] = 0xff;
        if (RELU == 1) E::ld(y, i, yv[u], cv, CV);
        if (RELU == 2) m[u] = ((const unsigned char*)y)[i];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) one(g[u], h[u], yv[u], m[u], xv[u], (r + u * stride) * CV + cv);
    }
    for (; r < M; r += stride) {
      const int64_t i = r * CV + cv;
      float g[8], xv[8], h[8], yv[8];
      unsigned int m = 0xff;
      E::ld(dy, i, g, cv, CV); E::ld(x, i, xv, cv, CV);
      if (dy2) E::ld(dy2, i, h, cv, CV);
      if (RELU == 1) E::ld(y, i, yv, cv, CV);
      if (RELU == 2) m = ((const unsigned char*)y)[i];
      one(g, h, yv, m, xv, i);
    }
  }
  block_reduce_rows<2>(acc, CVB, RPI, smem);
  if (tid < CVB) {
    float* p = part + (int64_t)blockIdx.x * 2 * C;
#pragma unroll
    for (int j = 0; j < 8; ++j) { p[E::chan(cv, j, CV)] = acc[0][j]; p[C + E::chan(cv, j, CV)] = acc[1][j]; }
  }
}

// Parameter gradients: overwritten, or (the entry point's `accumulate` argument) ADDED with float atomics -- a step that runs several backward
// passes over the same parameters, possibly on concurrent streams (the engine's half-batch passes), zeroes the slots once per step.

__global__ void bn_bwd_finalize_kernel(const float* __restrict__ part, int nblk, int C, int64_t M,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ c1, float* __restrict__ c2, int accumulate) {
  const int c = blockIdx.x * kFinCh + threadIdx.x % kFinCh, split = threadIdx.x / kFinCh;
  double s, q;
  reduce_partials_256(part, nblk, C, c, split, s, q);
  if (split != 0 || c >= C) return;
  if (accumulate) { atomicAdd(dbeta + c, (float)s); atomicAdd(dgamma + c, (float)q); }
  else { dbeta[c] = (float)s; dgamma[c] = (float)q; }
  c1[c] = (float)(s / (double)M); c2[c] = (float)(q / (double)M);
}

// the same finalize for consumers that run pass 2 on their operand load (lec_conv_f32_dgrad_fused / _wgrad_fused): pass 2 is
// dx = gamma invstd (g - c1 - xhat c2), xhat = (x - mean) invstd, i.e. dx = A g + B x + D with the per-channel A, B, D written here
__global__ void bn_bwd_coeffs_kernel(const float* __restrict__ part, int nblk, int C, int64_t M, const float* __restrict__ gamma,
                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                     float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef, int accumulate) {
  const int c = blockIdx.x * kFinCh + threadIdx.x % kFinCh, split = threadIdx.x / kFinCh;
  double s, q;
  reduce_partials_256(part, nblk, C, c, split, s, q);
  if (split != 0 || c >= C) return;
  if (accumulate) { atomicAdd(dbeta + c, (float)s); atomicAdd(dgamma + c, (float)q); }
  else { dbeta[c] = (float)s; dgamma[c] = (float)q; }
  const double c1 = s / (double)M, c2 = q / (double)M;
  const double is = (double)invstd[c], gs = (double)gamma[c] * is;
  coef[c] = (float)gs; coef[C + c] = (float)(-gs * is * c2); coef[2 * C + c] = (float)(gs * (is * c2 * (double)mean[c] - c1));
}

// backward, pass 2: dx = gamma*invstd * (g - mean(g) - xhat * mean(g*xhat));  d residual = g
template <typename E, bool RES, int RELU>
__global__ __launch_bounds__(kBnThreads) void bn_bwd_apply_kernel(const void* __restrict__ dy, const void* __restrict__ dy2,
                                                                  const void* __restrict__ y,
                                                                  const void* __restrict__ x, int64_t M, int CV, int RPI,
                                                                  const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd, const float* __restrict__ c1,
                                                                  const float* __restrict__ c2, void* __restrict__ dx,
                                                                  void* __restrict__ dres) {
  const int tid = threadIdx.x;
  const int cv = tid % CV, rg = tid / CV;
  if (rg >= RPI) return;
  float gs[8], mu[8], is[8], k1[8], k2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = E::chan(cv, j, CV);
    is[j] = invstd[c]; mu[j] = mean[c]; gs[j] = gamma[c] * is[j]; k1[j] = c1[c]; k2[j] = c2[c];
  }
  const int64_t stride = (int64_t)gridDim.x * RPI;
  auto one = [&](int64_t i) {
    float g[8], xv[8], h[8], yv[8], o[8];
    E::ld(dy, i, g, cv, CV); E::ld(x, i, xv, cv, CV);
    if (dy2) E::ld(dy2, i, h, cv, CV);
    unsigned int m0 = 0xff;
    if (RELU == 1) E::ld(y, i, yv, cv, CV);
    if (RELU == 2) m0 = ((const unsigned char*)y)[i];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a = g[j];
      if (dy2) a += h[j];
      if (RELU == 1) a = yv[j] > 0.0f ? a : 0.0f;
      if (RELU == 2) a = (m0 >> j) & 1u ? a : 0.0f;
      const float xh = (xv[j] - mu[j]) * is[j];
      o[j] = gs[j] * (a - k1[j] - xh * k2[j]);
      g[j] = a;
    }
    E::st(dx, i, o, cv, CV);
    if (RES) E::st(dres, i, g, cv, CV);
  };
  int64_t r = (int64_t)blockIdx.x * RPI + rg;
  for (; r + stride < M; r += 2 * stride) { one(r * CV + cv); one((r + stride) * CV + cv); }
  for (; r < M; r += stride) one(r * CV + cv);
}

// ---------------------------------------------------------------------------------------------------------------
// The stem's tail as ONE op (fp32, round 4): p = maxpool3x3s2(relu(bn(x))) -- torchvision ResNet `maxpool(relu(bn1(conv1(x))))`, oe_h.py:311,317.
// As three ops the 112 x 112 x 64 activation crossed HBM seven times forward (apply: read x, write z; pool: read z) and backward (pool backward:
// write dz; reduce: read dz, x; apply: read dz, x) at the two ends of every pass, where no convolution of that pass overlaps with them.  Here z and dz
// never exist in memory: forward applies scale / shift / ReLU to x on the pooling window's loads (p and the one-byte argmax are the outputs: the
// same bits the separate kernels produce); backward rebuilds dz for a 2 x 2 input patch from the <= 4 pooled gradients covering it (the gather of
// maxpool_bwd_kernel), masks it with [x * scale + shift > 0] and feeds BOTH BatchNorm passes from that.
// Layout as everywhere in this file (EF32): a thread holds 8 channels of a pixel as two runs of 4; argmax: one byte per pooled element, 8 per thread.
struct alignas(8) u8x8v { unsigned char v[8]; };

__global__ __launch_bounds__(kBnThreads) void bn_relu_pool_fwd_kernel(const void* __restrict__ x, int N, int H, int W, int CV, const float* __restrict__ scale,
                                                                      const float* __restrict__ shift, void* __restrict__ p, u8x8v* __restrict__ idx) {
  const int Ho = H / 2, Wo = W / 2;
  const int64_t total = (int64_t)N * Ho * Wo * CV;
  const int cv = threadIdx.x % CV;                              // (the grid stride is a multiple of CV: a thread keeps its channels)
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = scale[EF32::chan(cv, j, CV)]; sh[j] = shift[EF32::chan(cv, j, CV)]; }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i / CV;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho); const int n = (int)(r / Ho);
    float best[8]; unsigned char arg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { best[j] = -INFINITY; arg[j] = 0; }
    bool first = true;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int h = 2 * ho - 1 + kh;
      if (h < 0 || h >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int w = 2 * wo - 1 + kw;
        if (w < 0 || w >= W) continue;
        float v[8];
        EF32::ld(x, (((int64_t)n * H + h) * W + w) * CV + cv, v, cv, CV);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float f = v[j] * sc[j] + sh[j];                        // bn_apply_kernel's arithmetic, then its ReLU
          f = f > 0.0f ? f : 0.0f;
          if (first || f > best[j] || f != f) { best[j] = f; arg[j] = (unsigned char)(kh * 3 + kw); }   // maxpool_fwd_kernel's rule: first max wins; NaN propagates
        }
        first = false;
      }
    }
    u8x8v a;
#pragma unroll
    for (int j = 0; j < 8; ++j) a.v[j] = arg[j];
    EF32::st(p, i, best, cv, CV);
    idx[i] = a;
  }
}

// g (the gradient of the BatchNorm OUTPUT) and x for the 2 x 2 input patch (rows 2i, 2i+1; columns 2j, 2j+1) of one channel vector
__device__ __forceinline__ void bn_pool_patch(const void* __restrict__ dp, const void* __restrict__ dp2, const u8x8v* __restrict__ idx, const void* __restrict__ x, int n, int i, int j, int cv,
                                              int H, int W, int CV, const float (&sc)[8], const float (&sh)[8], float (&g)[4][8], float (&xv)[4][8]) {
  const int Ho = H / 2, Wo = W / 2;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    EF32::ld(x, (((int64_t)n * H + 2 * i + (q >> 1)) * W + 2 * j + (q & 1)) * CV + cv, xv[q], cv, CV);
#pragma unroll
    for (int c = 0; c < 8; ++c) g[q][c] = 0.0f;
  }
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int ho = i + a, wo = j + b;
      if (ho >= Ho || wo >= Wo) continue;
      const int64_t o = (((int64_t)n * Ho + ho) * Wo + wo) * CV + cv;
      const u8x8v am = idx[o];
      float d[8];
      EF32::ld(dp, o, d, cv, CV);
      if (dp2) {                                                // the pooled activation fed two branches: their gradients are added here, not by a pass of their own
        float d2[8];
        EF32::ld(dp2, o, d2, cv, CV);
#pragma unroll
        for (int c = 0; c < 8; ++c) d[c] += d2[c];
      }
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {                          // window (ho, wo) covers rows 2 ho - 1 .. 2 ho + 1: patch row pr sits at kh = pr + 1 - 2 a
        const int kh = pr + 1 - 2 * a;
        if (kh < 0 || kh > 2) continue;
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
          const int kw = pc + 1 - 2 * b;
          if (kw < 0 || kw > 2) continue;
          const int code = kh * 3 + kw;
#pragma unroll
          for (int c = 0; c < 8; ++c) if (am.v[c] == code) g[pr * 2 + pc][c] += d[c];     // (the order of maxpool_bwd_kernel's sums)
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int c = 0; c < 8; ++c) g[q][c] = xv[q][c] * sc[c] + sh[c] > 0.0f ? g[q][c] : 0.0f;      // the ReLU between the BatchNorm and the pooling
}

// pass 1: partial sums of g and g * xhat per block -> part[block][2][C]  (bn_bwd_reduce_kernel's layout: bn_bwd_finalize_kernel takes it from there)
__global__ __launch_bounds__(kBnThreads) void bn_pool_bwd_reduce_kernel(const void* __restrict__ dp, const void* __restrict__ dp2, const u8x8v* __restrict__ idx, const void* __restrict__ x,
                                                                        int N, int H, int W, int C, int CV, const float* __restrict__ gamma,
                                                                        const float* __restrict__ beta, const float* __restrict__ mean,
                                                                        const float* __restrict__ invstd, float* __restrict__ part) {
  __shared__ float smem[kBnThreads * 8];
  const int Ho = H / 2, Wo = W / 2;
  const int64_t total = (int64_t)N * Ho * Wo * CV;
  const int cv = threadIdx.x % CV;
  float sc[8], sh[8], mu[8], is[8], acc[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = EF32::chan(cv, j, CV);
    mu[j] = mean[c]; is[j] = invstd[c]; sc[j] = gamma[c] * is[j]; sh[j] = beta[c] - mu[j] * sc[j];       // bn_stats_finalize_kernel's scale / shift, bit for bit
    acc[0][j] = 0.f; acc[1][j] = 0.f;
  }
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = t / CV;
    const int j = (int)(r % Wo); r /= Wo;
    const int i = (int)(r % Ho); const int n = (int)(r / Ho);
    float g[4][8], xv[4][8];
    bn_pool_patch(dp, dp2, idx, x, n, i, j, cv, H, W, CV, sc, sh, g, xv);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int c = 0; c < 8; ++c) { acc[0][c] += g[q][c]; acc[1][c] += g[q][c] * ((xv[q][c] - mu[c]) * is[c]); }
  }
  block_reduce_rows<2>(acc, CV, kBnThreads / CV, smem);
  if (threadIdx.x < CV) {
    float* pp = part + (int64_t)blockIdx.x * 2 * C;
#pragma unroll
    for (int j = 0; j < 8; ++j) { pp[EF32::chan(cv, j, CV)] = acc[0][j]; pp[C + EF32::chan(cv, j, CV)] = acc[1][j]; }
  }
}

// pass 2: dx = gamma invstd (g - c1 - xhat c2) for the four positions of the patch
__global__ __launch_bounds__(kBnThreads) void bn_pool_bwd_apply_kernel(const void* __restrict__ dp, const void* __restrict__ dp2, const u8x8v* __restrict__ idx, const void* __restrict__ x,
                                                                       int N, int H, int W, int CV, const float* __restrict__ gamma,
                                                                       const float* __restrict__ beta, const float* __restrict__ mean,
                                                                       const float* __restrict__ invstd, const float* __restrict__ c1,
                                                                       const float* __restrict__ c2, void* __restrict__ dx) {
  const int Ho = H / 2, Wo = W / 2;
  const int64_t total = (int64_t)N * Ho * Wo * CV;
  const int cv = threadIdx.x % CV;
  float sc[8], sh[8], mu[8], is[8], k1[8], k2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = EF32::chan(cv, j, CV);
    mu[j] = mean[c]; is[j] = invstd[c]; sc[j] = gamma[c] * is[j]; sh[j] = beta[c] - mu[j] * sc[j]; k1[j] = c1[c]; k2[j] = c2[c];
  }
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = t / CV;
    const int j = (int)(r % Wo); r /= Wo;
    const int i = (int)(r % Ho); const int n = (int)(r / Ho);
    float g[4][8], xv[4][8];
    bn_pool_patch(dp, dp2, idx, x, n, i, j, cv, H, W, CV, sc, sh, g, xv);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float o[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) { const float xh = (xv[q][c] - mu[c]) * is[c]; o[c] = sc[c] * (g[q][c] - k1[c] - xh * k2[c]); }   // (sc = gamma invstd)
      EF32::st(dx, (((int64_t)n * H + 2 * i + (q >> 1)) * W + 2 * j + (q & 1)) * CV + cv, o, cv, CV);
    }
  }
}

static int bn_check(const char* who, int64_t M, int C) {
  LEC_CHECK_ARG(M > 0 && C > 0 && C % 8 == 0 && C <= 2048 && (C <= 512 || C % 512 == 0),
                "%s: need M > 0 and C a multiple of 8 up to 512, or 1024 / 1536 / 2048 (M=%lld C=%d)", who, (long long)M, C);
  return LEC_OK;
}

}  // namespace lec

extern "C" int64_t lec_bn_workspace_bytes(int C) {
  if (C <= 0) return LEC_E_ARG;
  return ((int64_t)lec::kBnMaxRows * 2 * C + 4 * (int64_t)C) * sizeof(float);
}

template <typename E> static int bn_fwd_impl(const void* x, const void* residual, int64_t M, int C, const float* gamma, const float* beta,
                          float eps, float momentum, float* running_mean, float* running_var, int training,
                          float* save_mean, float* save_invstd, void* y, int relu, uint8_t* relu_mask,
                          void* workspace, int64_t workspace_bytes, lec_stream_t stream) {
  using namespace lec;
  if (int rc = bn_check("bn_fwd", M, C)) return rc;
  LEC_CHECK_ARG(x && gamma && beta && y && workspace, "bn_fwd: null pointer");
  LEC_CHECK_ARG(workspace_bytes >= lec_bn_workspace_bytes(C), "bn_fwd: workspace too small");
  LEC_CHECK_ARG(training ? (save_mean && save_invstd) : (running_mean && running_var), "bn_fwd: statistics buffers missing");
  hipStream_t st = (hipStream_t)stream;
  BnGeom g = bn_geom(M, C);
  float* part = (float*)workspace;
  float* scale = part + (int64_t)kBnMaxRows * 2 * C; float* shift = scale + C;
  if (training > 1) {                                      // statistics partials already in the workspace (training - 2 rows)
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(kFinThreads), 0, st, part, training - 2, C, M, gamma, beta, eps,
                       momentum, running_mean, running_var, save_mean, save_invstd, scale, shift);
  } else if (training) {
    hipLaunchKernelGGL((bn_stats_kernel<E>), dim3(g.nrb, g.NCH), dim3(kBnThreads), 0, st, x, M, C, g.CV, g.CVB, g.RPIB, part);
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(kFinThreads), 0, st, part, g.nrb, C, M, gamma, beta, eps,
                       momentum, running_mean, running_var, save_mean, save_invstd, scale, shift);
  } else {
    hipLaunchKernelGGL(bn_eval_coeff_kernel, dim3((C + 255) / 256), dim3(256), 0, st, C, gamma, beta, eps, running_mean, running_var, scale, shift);
  }
  int64_t nb = (M + g.RPI - 1) / g.RPI; nb = (nb + 3) / 4;
  const int nblk = (int)(nb < 1 ? 1 : (nb > bn_apply_cap() ? bn_apply_cap() : nb));
#define A(RES_, RELU_) hipLaunchKernelGGL((bn_apply_kernel<E, RES_, RELU_>), dim3(nblk), dim3(kBnThreads), 0, st, x, residual, M, g.CV, g.RPI, scale, shift, y, relu_mask)
  if (residual) { if (relu) A(true, true); else A(true, false); } else { if (relu) A(false, true); else A(false, false); }
#undef A
  LEC_CHECK_LAUNCH("bn_fwd kernels");
  return LEC_OK;
}

template <typename E> static int bn_bwd_impl(const void* dy, const void* dy2, const void* y, const uint8_t* relu_mask, const void* x, int64_t M, int C,
                          const float* gamma, const float* save_mean, const float* save_invstd, void* dx,
                          void* dresidual, float* dgamma, float* dbeta, int relu, void* workspace,
                          int64_t workspace_bytes, int accumulate, lec_stream_t stream) {
  using namespace lec;
  if (int rc = bn_check("bn_bwd", M, C)) return rc;
  LEC_CHECK_ARG(dy && x && gamma && save_mean && save_invstd && dx && dgamma && dbeta && workspace, "bn_bwd: null pointer");
  LEC_CHECK_ARG(!relu || y || relu_mask, "bn_bwd: the forward output y or its bitmask is needed for the ReLU mask");
  const int rm = !relu ? 0 : (relu_mask ? 2 : 1);
  const void* ym = rm == 2 ? (const void*)relu_mask : y;
  LEC_CHECK_ARG(workspace_bytes >= lec_bn_workspace_bytes(C), "bn_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  BnGeom g = bn_geom(M, C);
  float* part = (float*)workspace;
  float* c1 = part + (int64_t)kBnMaxRows * 2 * C; float* c2 = c1 + C;
#define R(M_) hipLaunchKernelGGL((bn_bwd_reduce_kernel<E, M_>), dim3(g.nrb, g.NCH), dim3(kBnThreads), 0, st, dy, dy2, ym, x, M, C, g.CV, g.CVB, g.RPIB, save_mean, save_invstd, part, dresidual)
  if (rm == 0) R(0); else if (rm == 1) R(1); else R(2);
#undef R
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(kFinThreads), 0, st, part, g.nrb, C, M, dgamma, dbeta, c1, c2, accumulate ? 1 : 0);
  int64_t nb = (M + g.RPI - 1) / g.RPI; nb = (nb + 1) / 2;
  const int nblk = (int)(nb < 1 ? 1 : (nb > bn_apply_cap() ? bn_apply_cap() : nb));
#define A(RES_, RELU_) hipLaunchKernelGGL((bn_bwd_apply_kernel<E, RES_, RELU_>), dim3(nblk), dim3(kBnThreads), 0, st, dy, dy2, ym, x, M, g.CV, g.RPI, gamma, save_mean, save_invstd, c1, c2, dx, dresidual)
  if (dresidual) {
    // pass 1 has written g = masked(dy [+ dy2]) into dresidual: pass 2 reads that one tensor, no mask, no second stream
    hipLaunchKernelGGL((bn_bwd_apply_kernel<E, false, 0>), dim3(nblk), dim3(kBnThreads), 0, st, dresidual, nullptr,
                       nullptr, x, M, g.CV, g.RPI, gamma, save_mean, save_invstd, c1, c2, dx, nullptr);
  } else { if (rm == 0) A(false, 0); else if (rm == 1) A(false, 1); else A(false, 2); }
#undef A
  LEC_CHECK_LAUNCH("bn_bwd kernels");
  return LEC_OK;
}

extern "C" int64_t lec_bn_workspace_coeff_offset(int C) {                // byte offset of scale[C], shift[C] (forward) inside the workspace
  if (C <= 0) return LEC_E_ARG;
  return (int64_t)lec::kBnMaxRows * 2 * C * (int64_t)sizeof(float);
}

// scale / shift of an eval-mode BatchNorm (F.batch_norm(training=False)): the vectors lec_conv_f32_fwd_affine applies in its epilogue.  One small
// launch, in stream order: parameters and running statistics that earlier launches on the stream wrote (optimizer step, training forwards)
// are the ones it reads -- nothing on the host has to notice that they changed.
extern "C" int lec_bn_eval_coeffs_f32(int C, const float* gamma, const float* beta, float eps, const float* running_mean, const float* running_var,
                                      float* scale, float* shift, lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(C >= 1 && gamma && beta && running_mean && running_var && scale && shift, "bn_eval_coeffs: bad argument");
  hipLaunchKernelGGL(bn_eval_coeff_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, C, gamma, beta, eps, running_mean, running_var, scale, shift);
  LEC_CHECK_LAUNCH("bn_eval_coeff_kernel");
  return LEC_OK;
}

extern "C" int lec_bn_fwd_finalize(int64_t M, int C, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                   float* running_var, int n_partials, float* save_mean, float* save_invstd, void* workspace,
                                   int64_t workspace_bytes, lec_stream_t stream) {
  using namespace lec;
  if (int rc = bn_check("bn_fwd_finalize", M, C)) return rc;
  LEC_CHECK_ARG(gamma && beta && save_mean && save_invstd && workspace, "bn_fwd_finalize: null pointer");
  LEC_CHECK_ARG(n_partials >= 1 && n_partials <= kBnMaxRows, "bn_fwd_finalize: n_partials=%d outside 1..%d", n_partials, kBnMaxRows);
  LEC_CHECK_ARG(workspace_bytes >= lec_bn_workspace_bytes(C), "bn_fwd_finalize: workspace too small");
  float* part = (float*)workspace;
  float* scale = part + (int64_t)kBnMaxRows * 2 * C; float* shift = scale + C;
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(kFinThreads), 0, (hipStream_t)stream, part, n_partials, C, M, gamma, beta,
                     eps, momentum, running_mean, running_var, save_mean, save_invstd, scale, shift);
  LEC_CHECK_LAUNCH("bn_stats_finalize_kernel");
  return LEC_OK;
}

// The backward split into its stages, for callers that run pass 2 somewhere else (lec_conv1x1_wgrad_bnapply): pass 1 + finalize,
// finalize alone (partials left by a convolution epilogue), and pass 2 alone from the c1 / c2 the finalize left in the workspace.
template <typename E> static int bn_bwd_pass1_impl(const void* dy, const void* dy2, const uint8_t* relu_mask, const void* x, int64_t M, int C, const float* save_mean,
                                const float* save_invstd, void* g, float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes,
                                int accumulate, lec_stream_t stream) {
  using namespace lec;
  if (int rc = bn_check("bn_bwd_pass1", M, C)) return rc;
  LEC_CHECK_ARG(dy && x && save_mean && save_invstd && g && dgamma && dbeta && workspace, "bn_bwd_pass1: null pointer");
  LEC_CHECK_ARG(workspace_bytes >= lec_bn_workspace_bytes(C), "bn_bwd_pass1: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  BnGeom geo = bn_geom(M, C);
  float* part = (float*)workspace;
  float* c1 = part + (int64_t)kBnMaxRows * 2 * C; float* c2 = c1 + C;
#define R(M_) hipLaunchKernelGGL((bn_bwd_reduce_kernel<E, M_>), dim3(geo.nrb, geo.NCH), dim3(kBnThreads), 0, st, dy, dy2, relu_mask, x, M, C, geo.CV, geo.CVB, geo.RPIB, save_mean, save_invstd, part, g)
  if (relu_mask) R(2); else R(0);
#undef R
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(kFinThreads), 0, st, part, geo.nrb, C, M, dgamma, dbeta, c1, c2, accumulate ? 1 : 0);
  LEC_CHECK_LAUNCH("bn_bwd_pass1 kernels");
  return LEC_OK;
}

extern "C" int lec_bn_bwd_finalize(int64_t M, int C, int n_partials, float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes,
                                   int accumulate, lec_stream_t stream) {
  using namespace lec;
  if (int rc = bn_check("bn_bwd_finalize", M, C)) return rc;
  LEC_CHECK_ARG(dgamma && dbeta && workspace, "bn_bwd_finalize: null pointer");
  LEC_CHECK_ARG(n_partials >= 1 && n_partials <= kBnMaxRows, "bn_bwd_finalize: n_partials=%d outside 1..%d", n_partials, kBnMaxRows);
  LEC_CHECK_ARG(workspace_bytes >= lec_bn_workspace_bytes(C), "bn_bwd_finalize: workspace too small");
  float* part = (float*)workspace;
  float* c1 = part + (int64_t)kBnMaxRows * 2 * C; float* c2 = c1 + C;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(kFinThreads), 0, (hipStream_t)stream, part, n_partials, C, M, dgamma, dbeta,
                     c1, c2, accumulate ? 1 : 0);
  LEC_CHECK_LAUNCH("bn_bwd_finalize_kernel");
  return LEC_OK;
}

extern "C" int lec_bn_bwd_coeffs_f32(int64_t M, int C, int n_partials, const float* gamma, const float* save_mean, const float* save_invstd,
                                     float* dgamma, float* dbeta, float* coef, void* workspace, int64_t workspace_bytes, int accumulate, lec_stream_t stream) {
  using namespace lec;
  if (int rc = bn_check("bn_bwd_coeffs", M, C)) return rc;
  LEC_CHECK_ARG(gamma && save_mean && save_invstd && dgamma && dbeta && coef && workspace, "bn_bwd_coeffs: null pointer");
  LEC_CHECK_ARG(n_partials >= 1 && n_partials <= kBnMaxRows, "bn_bwd_coeffs: n_partials=%d outside 1..%d", n_partials, kBnMaxRows);
  LEC_CHECK_ARG(workspace_bytes >= lec_bn_workspace_bytes(C), "bn_bwd_coeffs: workspace too small");
  hipLaunchKernelGGL(bn_bwd_coeffs_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(kFinThreads), 0, (hipStream_t)stream, (const float*)workspace, n_partials, C, M,
                     gamma, save_mean, save_invstd, dgamma, dbeta, coef, accumulate ? 1 : 0);
  LEC_CHECK_LAUNCH("bn_bwd_coeffs_kernel");
  return LEC_OK;
}

// pass 1 (writes g = mask * (dy [+ dy2])) + the coefficient finalize, for a consumer that runs pass 2 on its operand load
extern "C" int lec_bn_bwd_pass1_coeffs_f32(const void* dy, const void* dy2, const uint8_t* relu_mask, const void* x, int64_t M, int C, const float* gamma,
                                           const float* save_mean, const float* save_invstd, void* g, float* dgamma, float* dbeta, float* coef,
                                           void* workspace, int64_t workspace_bytes, int accumulate, lec_stream_t stream) {
  using namespace lec;
  typedef EF32 E;
  if (int rc = bn_check("bn_bwd_pass1_coeffs", M, C)) return rc;
  LEC_CHECK_ARG(dy && x && gamma && save_mean && save_invstd && g && dgamma && dbeta && coef && workspace, "bn_bwd_pass1_coeffs: null pointer");
  LEC_CHECK_ARG(workspace_bytes >= lec_bn_workspace_bytes(C), "bn_bwd_pass1_coeffs: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  BnGeom geo = bn_geom(M, C);
  float* part = (float*)workspace;
#define R(M_) hipLaunchKernelGGL((bn_bwd_reduce_kernel<E, M_>), dim3(geo.nrb, geo.NCH), dim3(kBnThreads), 0, st, dy, dy2, relu_mask, x, M, C, geo.CV, geo.CVB, geo.RPIB, save_mean, save_invstd, part, g)
  if (relu_mask) R(2); else R(0);
#undef R
  hipLaunchKernelGGL(bn_bwd_coeffs_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(kFinThreads), 0, st, part, geo.nrb, C, M, gamma, save_mean, save_invstd,
                     dgamma, dbeta, coef, accumulate ? 1 : 0);
  LEC_CHECK_LAUNCH("bn_bwd_pass1_coeffs kernels");
  return LEC_OK;
}

template <typename E> static int bn_bwd_apply_impl(const void* g, const void* x, int64_t M, int C, const float* gamma, const float* save_mean, const float* save_invstd,
                                void* dx, void* workspace, int64_t workspace_bytes, lec_stream_t stream) {
  using namespace lec;
  if (int rc = bn_check("bn_bwd_apply", M, C)) return rc;
  LEC_CHECK_ARG(g && x && gamma && save_mean && save_invstd && dx && workspace, "bn_bwd_apply: null pointer");
  LEC_CHECK_ARG(workspace_bytes >= lec_bn_workspace_bytes(C), "bn_bwd_apply: workspace too small");
  BnGeom geo = bn_geom(M, C);
  float* c1 = (float*)workspace + (int64_t)kBnMaxRows * 2 * C; float* c2 = c1 + C;
  int64_t nb = (M + geo.RPI - 1) / geo.RPI; nb = (nb + 1) / 2;
  const int nblk = (int)(nb < 1 ? 1 : (nb > bn_apply_cap() ? bn_apply_cap() : nb));
  hipLaunchKernelGGL((bn_bwd_apply_kernel<E, false, 0>), dim3(nblk), dim3(kBnThreads), 0, (hipStream_t)stream, g, nullptr,
                     nullptr, x, M, geo.CV, geo.RPI, gamma, save_mean, save_invstd, c1, c2, dx, nullptr);
  LEC_CHECK_LAUNCH("bn_bwd_apply_kernel");
  return LEC_OK;
}

template <typename E> static int bn_bwd_prereduced_impl(const void* g, const void* x, int64_t M, int C, const float* gamma, const float* save_mean,
                                     const float* save_invstd, int n_partials, void* dx, float* dgamma, float* dbeta, void* workspace,
                                     int64_t workspace_bytes, int accumulate, lec_stream_t stream) {
  using namespace lec;
  if (int rc = bn_check("bn_bwd_prereduced", M, C)) return rc;
  LEC_CHECK_ARG(g && x && gamma && save_mean && save_invstd && dx && dgamma && dbeta && workspace, "bn_bwd_prereduced: null pointer");
  LEC_CHECK_ARG(n_partials >= 1 && n_partials <= kBnMaxRows, "bn_bwd_prereduced: n_partials=%d outside 1..%d", n_partials, kBnMaxRows);
  LEC_CHECK_ARG(workspace_bytes >= lec_bn_workspace_bytes(C), "bn_bwd_prereduced: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  BnGeom geo = bn_geom(M, C);
  float* part = (float*)workspace;
  float* c1 = part + (int64_t)kBnMaxRows * 2 * C; float* c2 = c1 + C;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(kFinThreads), 0, st, part, n_partials, C, M, dgamma, dbeta, c1, c2, accumulate ? 1 : 0);
  int64_t nb = (M + geo.RPI - 1) / geo.RPI; nb = (nb + 1) / 2;
  const int nblk = (int)(nb < 1 ? 1 : (nb > bn_apply_cap() ? bn_apply_cap() : nb));
  hipLaunchKernelGGL((bn_bwd_apply_kernel<E, false, 0>), dim3(nblk), dim3(kBnThreads), 0, st, g, nullptr,
                     nullptr, x, M, geo.CV, geo.RPI, gamma, save_mean, save_invstd, c1, c2, dx, nullptr);
  LEC_CHECK_LAUNCH("bn_bwd_prereduced kernels");
  return LEC_OK;
}

template <typename E> static int bn_fwd_prestat_impl(const void* x, const void* residual, int64_t M, int C, const float* gamma, const float* beta,
                                  float eps, float momentum, float* running_mean, float* running_var, int n_partials,
                                  float* save_mean, float* save_invstd, void* y, int relu, uint8_t* relu_mask, void* workspace,
                                  int64_t workspace_bytes, lec_stream_t stream) {
  LEC_CHECK_ARG(n_partials >= 1 && n_partials <= lec::kBnMaxRows, "bn_fwd_prestat: n_partials=%d outside 1..%d", n_partials, lec::kBnMaxRows);
  return bn_fwd_impl<E>(x, residual, M, C, gamma, beta, eps, momentum, running_mean, running_var, 2 + n_partials, save_mean, save_invstd,
                    y, relu, relu_mask, workspace, workspace_bytes, stream);
}

// ---- C entry points: bf16 (the MI355X-native storage) and fp32 (the reference's precision) instances of the same kernels
extern "C" int lec_bn_fwd(const void* x, const void* residual, int64_t M, int C, const float* gamma, const float* beta, float eps, float momentum, float* running_mean, float* running_var, int training, float* save_mean, float* save_invstd, void* y, int relu, uint8_t* relu_mask, void* workspace, int64_t workspace_bytes, lec_stream_t stream) {
  return bn_fwd_impl<lec::EBf16>(x, residual, M, C, gamma, beta, eps, momentum, running_mean, running_var, training, save_mean, save_invstd, y, relu, relu_mask, workspace, workspace_bytes, stream);
}
extern "C" int lec_bn_fwd_f32(const void* x, const void* residual, int64_t M, int C, const float* gamma, const float* beta, float eps, float momentum, float* running_mean, float* running_var, int training, float* save_mean, float* save_invstd, void* y, int relu, uint8_t* relu_mask, void* workspace, int64_t workspace_bytes, lec_stream_t stream) {
  return bn_fwd_impl<lec::EF32>(x, residual, M, C, gamma, beta, eps, momentum, running_mean, running_var, training, save_mean, save_invstd, y, relu, relu_mask, workspace, workspace_bytes, stream);
}
extern "C" int lec_bn_bwd(const void* dy, const void* dy2, const void* y, const uint8_t* relu_mask, const void* x, int64_t M, int C, const float* gamma, const float* save_mean, const float* save_invstd, void* dx, void* dresidual, float* dgamma, float* dbeta, int relu, void* workspace, int64_t workspace_bytes, int accumulate, lec_stream_t stream) {
  return bn_bwd_impl<lec::EBf16>(dy, dy2, y, relu_mask, x, M, C, gamma, save_mean, save_invstd, dx, dresidual, dgamma, dbeta, relu, workspace, workspace_bytes, accumulate, stream);
}
extern "C" int lec_bn_bwd_f32(const void* dy, const void* dy2, const void* y, const uint8_t* relu_mask, const void* x, int64_t M, int C, const float* gamma, const float* save_mean, const float* save_invstd, void* dx, void* dresidual, float* dgamma, float* dbeta, int relu, void* workspace, int64_t workspace_bytes, int accumulate, lec_stream_t stream) {
  return bn_bwd_impl<lec::EF32>(dy, dy2, y, relu_mask, x, M, C, gamma, save_mean, save_invstd, dx, dresidual, dgamma, dbeta, relu, workspace, workspace_bytes, accumulate, stream);
}
extern "C" int lec_bn_bwd_pass1(const void* dy, const void* dy2, const uint8_t* relu_mask, const void* x, int64_t M, int C, const float* save_mean, const float* save_invstd, void* g, float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes, int accumulate, lec_stream_t stream) {
  return bn_bwd_pass1_impl<lec::EBf16>(dy, dy2, relu_mask, x, M, C, save_mean, save_invstd, g, dgamma, dbeta, workspace, workspace_bytes, accumulate, stream);
}
extern "C" int lec_bn_bwd_pass1_f32(const void* dy, const void* dy2, const uint8_t* relu_mask, const void* x, int64_t M, int C, const float* save_mean, const float* save_invstd, void* g, float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes, int accumulate, lec_stream_t stream) {
  return bn_bwd_pass1_impl<lec::EF32>(dy, dy2, relu_mask, x, M, C, save_mean, save_invstd, g, dgamma, dbeta, workspace, workspace_bytes, accumulate, stream);
}
extern "C" int lec_bn_bwd_apply(const void* g, const void* x, int64_t M, int C, const float* gamma, const float* save_mean, const float* save_invstd, void* dx, void* workspace, int64_t workspace_bytes, lec_stream_t stream) {
  return bn_bwd_apply_impl<lec::EBf16>(g, x, M, C, gamma, save_mean, save_invstd, dx, workspace, workspace_bytes, stream);
}
extern "C" int lec_bn_bwd_apply_f32(const void* g, const void* x, int64_t M, int C, const float* gamma, const float* save_mean, const float* save_invstd, void* dx, void* workspace, int64_t workspace_bytes, lec_stream_t stream) {
  return bn_bwd_apply_impl<lec::EF32>(g, x, M, C, gamma, save_mean, save_invstd, dx, workspace, workspace_bytes, stream);
}
extern "C" int lec_bn_bwd_prereduced(const void* g, const void* x, int64_t M, int C, const float* gamma, const float* save_mean, const float* save_invstd, int n_partials, void* dx, float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes, int accumulate, lec_stream_t stream) {
  return bn_bwd_prereduced_impl<lec::EBf16>(g, x, M, C, gamma, save_mean, save_invstd, n_partials, dx, dgamma, dbeta, workspace, workspace_bytes, accumulate, stream);
}
extern "C" int lec_bn_bwd_prereduced_f32(const void* g, const void* x, int64_t M, int C, const float* gamma, const float* save_mean, const float* save_invstd, int n_partials, void* dx, float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes, int accumulate, lec_stream_t stream) {
  return bn_bwd_prereduced_impl<lec::EF32>(g, x, M, C, gamma, save_mean, save_invstd, n_partials, dx, dgamma, dbeta, workspace, workspace_bytes, accumulate, stream);
}
extern "C" int lec_bn_fwd_prestat(const void* x, const void* residual, int64_t M, int C, const float* gamma, const float* beta, float eps, float momentum, float* running_mean, float* running_var, int n_partials, float* save_mean, float* save_invstd, void* y, int relu, uint8_t* relu_mask, void* workspace, int64_t workspace_bytes, lec_stream_t stream) {
  return bn_fwd_prestat_impl<lec::EBf16>(x, residual, M, C, gamma, beta, eps, momentum, running_mean, running_var, n_partials, save_mean, save_invstd, y, relu, relu_mask, workspace, workspace_bytes, stream);
}
extern "C" int lec_bn_fwd_prestat_f32(const void* x, const void* residual, int64_t M, int C, const float* gamma, const float* beta, float eps, float momentum, float* running_mean, float* running_var, int n_partials, float* save_mean, float* save_invstd, void* y, int relu, uint8_t* relu_mask, void* workspace, int64_t workspace_bytes, lec_stream_t stream) {
  return bn_fwd_prestat_impl<lec::EF32>(x, residual, M, C, gamma, beta, eps, momentum, running_mean, running_var, n_partials, save_mean, save_invstd, y, relu, relu_mask, workspace, workspace_bytes, stream);
}


// ---- the stem's tail as one op (kernels above) --------------------------------------------------------------------------------------------
static int bn_pool_check(const char* who, int N, int H, int W, int C) {
  LEC_CHECK_ARG(N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C >= 8 && C <= 512 && C % 8 == 0 && lec::kBnThreads % (C / 8) == 0,
                "%s: need even H, W and C / 8 a divisor of %d up to 64 (N=%d H=%d W=%d C=%d)", who, lec::kBnThreads, N, H, W, C);
  LEC_CHECK_ARG((int64_t)N * H * W * C * 4 < (1ll << 40), "%s: tensor too large", who);
  return LEC_OK;
}

extern "C" int lec_bn_relu_maxpool_fwd_f32(const void* x, int N, int H, int W, int C, const float* scale, const float* shift, void* p, uint8_t* argmax,
                                           lec_stream_t stream) {
  using namespace lec;
  if (int rc = bn_pool_check("bn_relu_maxpool_fwd", N, H, W, C)) return rc;
  LEC_CHECK_ARG(x && scale && shift && p && argmax, "bn_relu_maxpool_fwd: null pointer");
  const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 8);
  int64_t nb = (total + kBnThreads - 1) / kBnThreads; const int nblk = (int)(nb > 8192 ? 8192 : nb);
  hipLaunchKernelGGL(bn_relu_pool_fwd_kernel, dim3(nblk), dim3(kBnThreads), 0, (hipStream_t)stream, x, N, H, W, C / 8, scale, shift, p, (u8x8v*)argmax);
  LEC_CHECK_LAUNCH("bn_relu_pool_fwd_kernel");
  return LEC_OK;
}

extern "C" int lec_bn_relu_maxpool_bwd_f32(const void* dp, const void* dp2, const uint8_t* argmax, const void* x, int N, int H, int W, int C, const float* gamma, const float* beta,
                                           const float* save_mean, const float* save_invstd, void* dx, float* dgamma, float* dbeta, void* workspace,
                                           int64_t workspace_bytes, int accumulate, lec_stream_t stream) {
  using namespace lec;
  if (int rc = bn_pool_check("bn_relu_maxpool_bwd", N, H, W, C)) return rc;
  LEC_CHECK_ARG(dp && argmax && x && gamma && beta && save_mean && save_invstd && dx && dgamma && dbeta && workspace, "bn_relu_maxpool_bwd: null pointer");
  LEC_CHECK_ARG(workspace_bytes >= lec_bn_workspace_bytes(C), "bn_relu_maxpool_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int64_t M = (int64_t)N * H * W;
  const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 8);
  float* part = (float*)workspace;
  float* c1 = part + (int64_t)kBnMaxRows * 2 * C; float* c2 = c1 + C;
  int64_t nb = (total + kBnThreads - 1) / kBnThreads;
  const int nred = (int)(nb > kBnMaxBlocks ? kBnMaxBlocks : nb);
  hipLaunchKernelGGL(bn_pool_bwd_reduce_kernel, dim3(nred), dim3(kBnThreads), 0, st, dp, dp2, (const u8x8v*)argmax, x, N, H, W, C, C / 8, gamma, beta, save_mean,
                     save_invstd, part);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(kFinThreads), 0, st, part, nred, C, M, dgamma, dbeta, c1, c2, accumulate ? 1 : 0);
  const int napp = (int)(nb > 16384 ? 16384 : nb);
  hipLaunchKernelGGL(bn_pool_bwd_apply_kernel, dim3(napp), dim3(kBnThreads), 0, st, dp, dp2, (const u8x8v*)argmax, x, N, H, W, C / 8, gamma, beta, save_mean, save_invstd,
                     c1, c2, dx);
  LEC_CHECK_LAUNCH("bn_relu_maxpool_bwd kernels");
  return LEC_OK;
}
