// 1x1 stride-1 convolution forward on NHWC bf16 activations WITH the BatchNorm statistics of its output in the epilogue
// (lec_conv1x1_fwd).  In the reference this is torchvision Bottleneck's `bn3(conv3(out))` / `downsample` pair
// inside FeatCNN18's backbone (oe_h.py:311,317): a library convolution that writes Y, then a statistics pass that reads
// all of Y again.  For the wide block outputs of layer1 (Cin = 64 -> Cout = 256 at 56x56: 822 MB of Y at the bench batch)
// both are HBM-bound, and the statistics pass costs as much as the convolution's own write.  Here the convolution is
// Y[M, N] = X[M, K] * W[N, K]^T on MFMA (v_mfma_f32_32x32x16_bf16), HBM-bound by construction (K = 64: 640 B moved per
// row against 32 K flop), and every workgroup leaves per-channel sum / sum-of-squares partials of the bf16-rounded Y in the
// layout the BatchNorm finalize kernel consumes -- the statistics pass disappears.
//
// Work decomposition: a workgroup = 4 waves, W (N x K bf16, 32 KB) resident in LDS for the whole launch; a wave owns
// 32-row strips of X.  Per strip: the strip's B fragments come straight from global memory (16 B per lane, prefetched one
// strip ahead); 8 n-tiles x 4 k-steps of MFMA with A = W from LDS give D'[n][m] (n in registers, m on the lane), i.e. 4
// consecutive output channels per lane and register group -- converted to bf16, transposed through a per-wave LDS tile
// 64 channels at a time, read back as 16-byte row segments, accumulated into the statistics and stored coalesced
// (non-temporal: Y is consumed by a later kernel and is far larger than the caches).
//
// Roofline: HBM.  Algorithmic bytes per output row: 2 K (read X) + 2 N (write Y); the weights and partials are noise.
#include <cstdlib>
#include <hip/hip_bf16.h>
#include "lec_common.h"
#include "tuning.h"
#include <atomic>

namespace lec {

typedef short bf16x8_t __attribute__((ext_vector_type(8)));      // MFMA A/B fragment: 8 bf16 = 4 VGPRs
typedef float f32x16_t __attribute__((ext_vector_type(16)));     // 32x32 accumulator tile: 16 VGPRs per lane
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned short c1_f2bf(float f) {
  __hip_bfloat16 h = __float2bfloat16(f);
  return *reinterpret_cast<unsigned short*>(&h);
}
__device__ __forceinline__ float c1_bf2f(unsigned short u) { return __uint_as_float(((unsigned int)u) << 16); }

constexpr int kC1Threads = 256;
constexpr int kC1MaxBlocks = 512;          // = kBnMaxBlocks: the partials land in the BatchNorm workspace

// K: reduction width (input channels); N: output channels handled by ONE workgroup (blockIdx.y selects the N-wide column
// block of an Ntot-wide output: a layer wider than its weights' LDS budget re-reads X once per column block);
// STATS: leave the per-channel partials (forward) or not (the same kernel serves the data gradient: X := dY, W := W^T).
// FOLD = 2 / 3, the forward counterpart for conv3 -> bn3 (+ identity, ReLU): the convolution reads a quarter of what it writes, so
// it is cheaper to run it twice than to let the BatchNorm re-read its output.  FOLD = 3 is the product with the statistics
// epilogue and NO store; after the finalize kernel, FOLD = 2 forms the product again and its epilogue applies scale / shift, adds
// the residual (fa.dy2), clamps, and writes y (kept for backward), z (fa.xbn, written) and z's bitmask (fa.mask, written);
// fa.mean / fa.invstd carry scale / shift.  The BatchNorm apply pass (read y, residual; write z, mask) disappears.
// FOLD = 1 (data gradient feeding a forked BatchNorm+residual+ReLU output, STATS set): the epilogue is pass 1 of that
// BatchNorm's backward.  With the strip's rows in registers on their way out it reads the other branch's gradient dy2,
// the ReLU bitmask and the BatchNorm's input x at the same addresses, forms g = mask * (dY W + dy2) rounded to bf16, writes
// g instead of the raw product and leaves per-channel partials of (sum g, sum g * xhat) where the forward form leaves
// (sum y, sum y^2) -- the standalone reduce pass (read dy, dy2, x, mask; write g) and this kernel's own write of dy go away.
struct FoldArgs {
  const unsigned short* dy2;      // [M][Ntot] bf16
  const unsigned short* xbn;      // [M][Ntot] bf16: input of the BatchNorm whose output this gradient belongs to
  const unsigned char* mask;      // [M][Ntot / 8] ReLU bitmask of that output
  const float* mean;              // [Ntot]
  const float* invstd;            // [Ntot]
};

template <int K, int N, bool STATS, int WAVES, int FOLD>
__global__ __launch_bounds__(WAVES * 64) void conv1x1_fwd_stats_kernel(const unsigned short* __restrict__ X,
                                                                       const unsigned short* __restrict__ Wt, int64_t M, int Ntot,
                                                                       unsigned short* __restrict__ Y, float* __restrict__ part, int wtrans,
                                                                       FoldArgs fa) {
  constexpr int KS = K / 16;                 // k-steps of one MFMA
  constexpr int NC = N / 64;                 // 64-channel chunks of the epilogue
  Y += (int64_t)blockIdx.y * N;
  constexpr int WLD = K + 8;                 // padded LDS row (bf16 elements): 16-byte reads of 32 rows hit distinct banks
  constexpr int YLD = 64 + 8;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  unsigned short* Ws = smem;                                     // [N][WLD]
  unsigned short* Ys = Ws + N * WLD + (threadIdx.x >> 6) * 32 * YLD;   // per wave [32][YLD]
  float* Ss = (float*)(smem + N * WLD);                         // [4 waves][2][N]: end of the launch only, reuses the Y tiles
  float* Ms = (float*)(smem + N * WLD + WAVES * 32 * YLD);      // FOLD: [2][N] mean, invstd of this column block

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  if (FOLD == 1 || FOLD == 2) {
    for (int e = threadIdx.x; e < N; e += WAVES * 64) {
      Ms[e] = fa.mean[(int)blockIdx.y * N + e];
      Ms[N + e] = fa.invstd[(int)blockIdx.y * N + e];
    }
  }
  // ---- weights -> LDS, once.  wtrans: the matrix arrives as [K][Ntot] (the FORWARD weight of the layer whose data gradient
  // this launch is) and is transposed on the way in -- no transpose kernel per layer and step
  if (!wtrans) {
    const unsigned short* Wb = Wt + (int64_t)blockIdx.y * N * K;
    for (int e = threadIdx.x; e < N * (K / 8); e += WAVES * 64) {
      const int n = e / (K / 8), c = e - n * (K / 8);
      *(u32x4_t*)(Ws + n * WLD + c * 8) = *(const u32x4_t*)(Wb + (int64_t)n * K + c * 8);
    }
  } else {
    const unsigned short* Wb = Wt + (int64_t)blockIdx.y * N;
    for (int e = threadIdx.x; e < K * (N / 8); e += WAVES * 64) {
      const int k = e / (N / 8), c = e - k * (N / 8);
      const u32x4_t v = *(const u32x4_t*)(Wb + (int64_t)k * Ntot + c * 8);
      const unsigned int w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) Ws[(c * 8 + j) * WLD + k] = (unsigned short)(w4[j >> 1] >> ((j & 1) * 16));
    }
  }
  __syncthreads();

  float st_s[NC][8], st_q[NC][8];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) { st_s[c][j] = 0.0f; st_q[c][j] = 0.0f; }

  const int64_t nstrips = M / 32;
  const int64_t stride = (int64_t)gridDim.x * WAVES;
  int64_t s = (int64_t)blockIdx.x * WAVES + wave;
  bf16x8_t xb[KS], xn[KS];
  if (s < nstrips) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xb[ks] = *(const bf16x8_t*)(X + (s * 32 + r) * K + ks * 16 + h * 8);
  }
  for (; s < nstrips; s += stride) {
    const int64_t sn = s + stride;
    if (sn < nstrips) {                                            // prefetch the next strip's fragments
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) xn[ks] = *(const bf16x8_t*)(X + (sn * 32 + r) * K + ks * 16 + h * 8);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      // FOLD: this chunk's epilogue operands, requested before the MFMAs (lane -> rows (lane >> 3) + 8 i, 8 channels)
      u32x4_t f_d2[4], f_x[4];
      unsigned int f_m[4];
      float f_mu[8], f_is[8];
      if (FOLD == 1 || FOLD == 2) {
        // addresses as (wave-uniform 64-bit base) + (one 32-bit lane offset): spelled out, the compiler otherwise keeps a
        // 64-bit vector address per tensor, chunk and row alive across the strip loop (~100 registers)
        const int su = __builtin_amdgcn_readfirstlane((int)s);
        const int64_t ub = ((int64_t)su * 32) * Ntot + (int)blockIdx.y * N + c * 64;
        const int loff = (lane >> 3) * Ntot + (lane & 7) * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int64_t ue = ub + (int64_t)(8 * i) * Ntot;
          f_d2[i] = __builtin_nontemporal_load((const u32x4_t*)(fa.dy2 + ue + loff));
          if (FOLD == 1) {
            f_x[i] = __builtin_nontemporal_load((const u32x4_t*)(fa.xbn + ue + loff));
            f_m[i] = (unsigned int)(fa.mask + (ue >> 3))[loff >> 3];
          }
        }
      }
      f32x16_t acc[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.0f;
        const unsigned short* wrow = Ws + ((c * 2 + t) * 32 + r) * WLD + h * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8_t a = *(const bf16x8_t*)(wrow + ks * 16);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xb[ks], acc[t], 0, 0, 0);
        }
      }
      // D'[n][m]: lane holds m = r, n = t*32 + 8g + 4h + (0..3) in registers 4g..4g+3  ->  Ys[m][n_local]
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          u32x2_t pk;
          pk.x = (unsigned int)c1_f2bf(acc[t][4 * g + 0]) | ((unsigned int)c1_f2bf(acc[t][4 * g + 1]) << 16);
          pk.y = (unsigned int)c1_f2bf(acc[t][4 * g + 2]) | ((unsigned int)c1_f2bf(acc[t][4 * g + 3]) << 16);
          *(u32x2_t*)(Ys + r * YLD + t * 32 + 8 * g + 4 * h) = pk;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // rows back out as 16-byte segments: lane -> rows (lane >> 3) + 8 i, channels c*64 + (lane & 7) * 8 .. + 8
      const int cc = lane & 7, r0 = lane >> 3;
      if (FOLD == 1 || FOLD == 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { f_mu[j] = Ms[c * 64 + cc * 8 + j]; f_is[j] = Ms[N + c * 64 + cc * 8 + j]; }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = r0 + 8 * i;
        u32x4_t v = *(const u32x4_t*)(Ys + row * YLD + cc * 8);
        const unsigned int w4[4] = {v.x, v.y, v.z, v.w};
        if (FOLD == 2) {
          // forward: z = relu(y * scale + shift + residual) from the ROUNDED y, exactly as bn_apply_kernel forms it; y is stored
          // too (the BatchNorm backward needs it), the bitmask of z > 0 as well
          const unsigned int d4[4] = {f_d2[i].x, f_d2[i].y, f_d2[i].z, f_d2[i].w};
          unsigned int o4[4] = {0u, 0u, 0u, 0u}, bits = 0u;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float a = c1_bf2f((unsigned short)(w4[j >> 1] >> ((j & 1) * 16))) * f_mu[j] + f_is[j];
            a += c1_bf2f((unsigned short)(d4[j >> 1] >> ((j & 1) * 16)));
            a = a > 0.0f ? a : 0.0f;
            const unsigned short zb = c1_f2bf(a);
            bits |= (c1_bf2f(zb) > 0.0f ? 1u : 0u) << j;
            o4[j >> 1] |= (unsigned int)zb << ((j & 1) * 16);
          }
          const int64_t e = (s * 32 + row) * Ntot + (int)blockIdx.y * N + c * 64 + cc * 8;
          u32x4_t z; z.x = o4[0]; z.y = o4[1]; z.z = o4[2]; z.w = o4[3];
          __builtin_nontemporal_store(z, (u32x4_t*)(const_cast<unsigned short*>(fa.xbn) + e));
          const_cast<unsigned char*>(fa.mask)[e >> 3] = (unsigned char)bits;
        } else if (FOLD == 1) {
          const unsigned int d4[4] = {f_d2[i].x, f_d2[i].y, f_d2[i].z, f_d2[i].w};
          const unsigned int x4[4] = {f_x[i].x, f_x[i].y, f_x[i].z, f_x[i].w};
          unsigned int o4[4] = {0u, 0u, 0u, 0u};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float a = c1_bf2f((unsigned short)(w4[j >> 1] >> ((j & 1) * 16)));
            a += c1_bf2f((unsigned short)(d4[j >> 1] >> ((j & 1) * 16)));
            a = (f_m[i] >> j) & 1u ? a : 0.0f;
            const unsigned short gb = c1_f2bf(a);
            a = c1_bf2f(gb);
            st_s[c][j] += a;
            st_q[c][j] += a * ((c1_bf2f((unsigned short)(x4[j >> 1] >> ((j & 1) * 16))) - f_mu[j]) * f_is[j]);
            o4[j >> 1] |= (unsigned int)gb << ((j & 1) * 16);
          }
          v.x = o4[0]; v.y = o4[1]; v.z = o4[2]; v.w = o4[3];
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float f = c1_bf2f((unsigned short)(w4[j >> 1] >> ((j & 1) * 16)));
            if (STATS) { st_s[c][j] += f; st_q[c][j] += f * f; }
          }
        }
        if (FOLD != 3) __builtin_nontemporal_store(v, (u32x4_t*)(Y + (s * 32 + row) * Ntot + c * 64 + cc * 8));
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xb[ks] = xn[ks];
  }

  if (!STATS) return;
  // ---- statistics: lanes with equal (lane & 7) own the same 8 channels of a chunk: fold the 8 row groups, then the 4 waves
#pragma unroll
  for (int c = 0; c < NC; ++c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a = st_s[c][j], b = st_q[c][j];
      a += __shfl_xor(a, 8, 64); b += __shfl_xor(b, 8, 64);
      a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
      a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
      st_s[c][j] = a; st_q[c][j] = b;
    }
  }
  __syncthreads();                                                 // every wave is done with its Ys tile
  if (lane < 8) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        Ss[(wave * 2 + 0) * N + c * 64 + lane * 8 + j] = st_s[c][j];
        Ss[(wave * 2 + 1) * N + c * 64 + lane * 8 + j] = st_q[c][j];
      }
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * N; e += WAVES * 64) {
    const int which = e / N, n = e - which * N;
    float a = 0.0f;
#pragma unroll
    for (int wv = 0; wv < WAVES; ++wv) a += Ss[(wv * 2 + which) * N + n];
    part[(int64_t)blockIdx.x * 2 * Ntot + which * Ntot + blockIdx.y * N + n] = a;
  }
}

template <int K, int N, bool STATS, int FOLD = 0>
static int launch_conv1x1(const void* x, const void* w, int64_t M, int Ntot, void* y, float* part, int* nblk_out, int wtrans, hipStream_t st,
                          FoldArgs fa = FoldArgs{nullptr, nullptr, nullptr, nullptr, nullptr}) {
  // eight waves per workgroup (two per SIMD) for K <= 128 (128 -> 512: 117 -> 105 us); the K = 256 instances, whose
  // fragment sets already fill the registers, measured better with four (256 -> 128: 229 vs 241 us)
  // (the FOLD = 1 epilogue's operand sets need more than the 256 registers per lane of an eight-wave workgroup: four waves)
  constexpr int WAVES = (FOLD != 1 && K <= 128 && ((size_t)N * (K + 8) + 8 * 32 * (64 + 8)) * sizeof(unsigned short) <= 160 * 1024) ? 8 : 4;
  static_assert(WAVES * 2 * N * sizeof(float) <= WAVES * 32 * (64 + 8) * sizeof(unsigned short), "statistics staging must fit the Y tiles");
  const size_t smem = ((size_t)N * (K + 8) + WAVES * 32 * (64 + 8)) * sizeof(unsigned short) + ((FOLD == 1 || FOLD == 2) ? 2 * N * sizeof(float) : 0);
  static std::atomic<bool> attr_set{false};
  if (smem > 64 * 1024 && !attr_set.load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute((const void*)conv1x1_fwd_stats_kernel<K, N, STATS, WAVES, FOLD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv1x1)");
    attr_set.store(true, std::memory_order_release);
  }
  const int64_t nstrips = M / 32;
  int64_t nb = (nstrips + WAVES - 1) / WAVES;
  int cap = kC1MaxBlocks / (Ntot / N) > 0 ? kC1MaxBlocks / (Ntot / N) : 1;
  if (WAVES == 8 && cap > 256) cap = 256;
  const int nblk = (int)(nb > cap ? cap : nb);
  hipLaunchKernelGGL((conv1x1_fwd_stats_kernel<K, N, STATS, WAVES, FOLD>), dim3(nblk, Ntot / N), dim3(WAVES * 64), smem, st, (const unsigned short*)x,
                     (const unsigned short*)w, M, Ntot, (unsigned short*)y, part, wtrans, fa);
  if (nblk_out) *nblk_out = nblk;
  LEC_CHECK_LAUNCH("conv1x1_fwd_stats_kernel");
  return LEC_OK;
}

// Wide-K form (K = 512: the 512 -> 128 convolutions of layer2 and the data gradient of its 128 -> 512 ones).  A strip's
// fragments no longer fit the register file next to their prefetch, so K streams through a ring of R 64-wide chunks:
// chunk ch is consumed by the MFMAs of all N/32 accumulator tiles (kept in registers for the whole strip) and its slot
// is refilled at once with chunk ch + R -- of this strip, or of the next one, so that the loads never drain.
template <int K, int N, bool STATS>
__global__ __launch_bounds__(kC1Threads) void conv1x1_bigk_kernel(const unsigned short* __restrict__ X,
                                                                  const unsigned short* __restrict__ Wt, int64_t M, int Ntot,
                                                                  unsigned short* __restrict__ Y, float* __restrict__ part, int wtrans) {
  constexpr int NCH = K / 64, R = 4, NT = N / 32, NC = N / 64;
  constexpr int WLD = K + 8, YLD = 64 + 8;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  unsigned short* Ws = smem;
  unsigned short* Ys = Ws + N * WLD + (threadIdx.x >> 6) * 32 * YLD;
  float* Ss = (float*)(smem + N * WLD);                         // reuses the Y tiles at the end of the launch
  Y += (int64_t)blockIdx.y * N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  if (!wtrans) {
    const unsigned short* Wb = Wt + (int64_t)blockIdx.y * N * K;
    for (int e = threadIdx.x; e < N * (K / 8); e += kC1Threads) {
      const int n = e / (K / 8), c = e - n * (K / 8);
      *(u32x4_t*)(Ws + n * WLD + c * 8) = *(const u32x4_t*)(Wb + (int64_t)n * K + c * 8);
    }
  } else {                                                       // [K][Ntot] forward weight, transposed on the way in
    const unsigned short* Wb = Wt + (int64_t)blockIdx.y * N;
    for (int e = threadIdx.x; e < K * (N / 8); e += kC1Threads) {
      const int k = e / (N / 8), c = e - k * (N / 8);
      const u32x4_t v = *(const u32x4_t*)(Wb + (int64_t)k * Ntot + c * 8);
      const unsigned int w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) Ws[(c * 8 + j) * WLD + k] = (unsigned short)(w4[j >> 1] >> ((j & 1) * 16));
    }
  }
  __syncthreads();
  float st_s[NC][8], st_q[NC][8];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) { st_s[c][j] = 0.0f; st_q[c][j] = 0.0f; }

  const int64_t nstrips = M / 32;
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t s = (int64_t)blockIdx.x * 4 + wave;
  bf16x8_t ring[R][4];
  if (s < nstrips) {
#pragma unroll
    for (int j = 0; j < R; ++j)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) ring[j][ks] = *(const bf16x8_t*)(X + (s * 32 + r) * K + j * 64 + ks * 16 + h * 8);
  }
  for (; s < nstrips; s += stride) {
    const int64_t sn = s + stride;
    f32x16_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[t][q] = 0.0f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      constexpr int dummy = 0; (void)dummy;
      const int slot = ch % R;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const unsigned short* wrow = Ws + (t * 32 + r) * WLD + ch * 64 + h * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)(wrow + ks * 16), ring[slot][ks], acc[t], 0, 0, 0);
      }
      const int nch = ch + R;
      if (nch < NCH) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ring[slot][ks] = *(const bf16x8_t*)(X + (s * 32 + r) * K + nch * 64 + ks * 16 + h * 8);
      } else if (sn < nstrips) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ring[slot][ks] = *(const bf16x8_t*)(X + (sn * 32 + r) * K + (nch - NCH) * 64 + ks * 16 + h * 8);
      }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          u32x2_t pk;
          pk.x = (unsigned int)c1_f2bf(acc[c * 2 + t][4 * g + 0]) | ((unsigned int)c1_f2bf(acc[c * 2 + t][4 * g + 1]) << 16);
          pk.y = (unsigned int)c1_f2bf(acc[c * 2 + t][4 * g + 2]) | ((unsigned int)c1_f2bf(acc[c * 2 + t][4 * g + 3]) << 16);
          *(u32x2_t*)(Ys + r * YLD + t * 32 + 8 * g + 4 * h) = pk;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int cc = lane & 7, r0 = lane >> 3;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = r0 + 8 * i;
        const u32x4_t v = *(const u32x4_t*)(Ys + row * YLD + cc * 8);
        const unsigned int w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = c1_bf2f((unsigned short)(w4[j >> 1] >> ((j & 1) * 16)));
          if (STATS) { st_s[c][j] += f; st_q[c][j] += f * f; }
        }
        __builtin_nontemporal_store(v, (u32x4_t*)(Y + (s * 32 + row) * Ntot + c * 64 + cc * 8));
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  if (!STATS) return;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a = st_s[c][j], b = st_q[c][j];
      a += __shfl_xor(a, 8, 64); b += __shfl_xor(b, 8, 64);
      a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
      a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
      st_s[c][j] = a; st_q[c][j] = b;
    }
  }
  __syncthreads();
  if (lane < 8) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        Ss[(wave * 2 + 0) * N + c * 64 + lane * 8 + j] = st_s[c][j];
        Ss[(wave * 2 + 1) * N + c * 64 + lane * 8 + j] = st_q[c][j];
      }
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * N; e += kC1Threads) {
    const int which = e / N, n = e - which * N;
    part[(int64_t)blockIdx.x * 2 * Ntot + which * Ntot + blockIdx.y * N + n] =
        Ss[(0 * 2 + which) * N + n] + Ss[(1 * 2 + which) * N + n] + Ss[(2 * 2 + which) * N + n] + Ss[(3 * 2 + which) * N + n];
  }
}

template <int K, int N, bool STATS>
static int launch_conv1x1_bigk(const void* x, const void* w, int64_t M, int Ntot, void* y, float* part, int* nblk_out, int wtrans, hipStream_t st) {
  static_assert(4 * 2 * N * sizeof(float) <= 4 * 32 * (64 + 8) * sizeof(unsigned short), "statistics staging must fit the Y tiles");
  const size_t smem = ((size_t)N * (K + 8) + 4 * 32 * (64 + 8)) * sizeof(unsigned short);
  static std::atomic<bool> attr_set{false};
  if (!attr_set.load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute((const void*)conv1x1_bigk_kernel<K, N, STATS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv1x1_bigk)");
    attr_set.store(true, std::memory_order_release);
  }
  const int64_t nstrips = M / 32;
  int64_t nb = (nstrips + 3) / 4;
  const int cap = kC1MaxBlocks / (Ntot / N) > 0 ? kC1MaxBlocks / (Ntot / N) : 1;
  const int nblk = (int)(nb > cap ? cap : nb);
  hipLaunchKernelGGL((conv1x1_bigk_kernel<K, N, STATS>), dim3(nblk, Ntot / N), dim3(kC1Threads), smem, st, (const unsigned short*)x,
                     (const unsigned short*)w, M, Ntot, (unsigned short*)y, part, wtrans);
  if (nblk_out) *nblk_out = nblk;
  LEC_CHECK_LAUNCH("conv1x1_bigk_kernel");
  return LEC_OK;
}

// 3x3 / stride 1 / pad 1 convolution, 64 -> 64 channels (layer1's conv2 at 56x56: 205 MB in, 205 MB out at the bench batch,
// the one 3x3 layer of ResNet-50 that sits near the HBM ridge rather than deep in MFMA territory).  The same wave-strip
// scheme as the 1x1 kernels with the nine taps as the K loop: tap (dr, ds) contributes X[pixel + (dr, ds)] * W_tap^T, its B
// fragments are the shifted pixels' channel vectors straight from global memory / L2 (zero outside the image), its A
// fragments W_tap from LDS (9 x 64 x 64 bf16 = 83 KB resident).  A strip's nine fragment sets live in registers; each is
// refilled with the next strip's same tap as soon as its MFMAs have issued (prefetch distance = one whole strip).  With W := flipped, transposed weights it is the layer's data gradient.
template <bool STATS>
__global__ __launch_bounds__(kC1Threads) void conv3x3_c64_kernel(const unsigned short* __restrict__ X,
                                                                 const unsigned short* __restrict__ Wt, int64_t M, int H, int W,
                                                                 unsigned short* __restrict__ Y, float* __restrict__ part, int wtrans) {
  constexpr int K = 64, N = 64, R = 9, WLD = K + 8, YLD = 64 + 8;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  unsigned short* Ws = smem;                                     // [9 taps][64 out][WLD]
  unsigned short* Ys = Ws + 9 * N * WLD + (threadIdx.x >> 6) * 32 * YLD;
  float* Ss = (float*)(smem + 9 * N * WLD);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  // weights arrive as [co][3][3][ci] (a channels_last conv weight): LDS wants [tap][co][ci]
  for (int e = threadIdx.x; e < 9 * N * (K / 8); e += kC1Threads) {
    const int c8 = e % (K / 8); int t = e / (K / 8);
    const int co = t % N; const int tap = t / N;
    const u32x4_t v = *(const u32x4_t*)(Wt + ((int64_t)co * 9 + tap) * K + c8 * 8);
    if (!wtrans) *(u32x4_t*)(Ws + (tap * N + co) * WLD + c8 * 8) = v;
    else {       // forward weight [co][r][s][ci] of the layer whose data gradient this is: W'[ci][2-r][2-s][co], flipped and transposed here
      const unsigned int w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) Ws[((8 - tap) * N + c8 * 8 + j) * WLD + co] = (unsigned short)(w4[j >> 1] >> ((j & 1) * 16));
    }
  }
  __syncthreads();
  float st_s[8], st_q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { st_s[j] = 0.0f; st_q[j] = 0.0f; }

  const int nstrips = (int)(M / 32);
  const int stride = (int)gridDim.x * 4;
  const int HW = H * W;
  // XCD-aware order: workgroup ids go round-robin over the 8 XCDs, each with its own L2.  Give every XCD a CONTIGUOUS run of
  // strips, so that the image rows above and below a strip -- read for the dr = -1 / +1 taps -- are rows its own L2 has
  // just served to the neighbouring workgroups (otherwise every XCD fetches each input row about three times).
  const int nx = 8, per = (int)gridDim.x / nx;
  const int lb = (per > 0 && (int)gridDim.x % nx == 0) ? ((int)blockIdx.x % nx) * per + (int)blockIdx.x / nx : (int)blockIdx.x;
  int s = lb * 4 + wave;
  bf16x8_t ring[R][4];
  // the lane's pixel of a strip, decoded once: (row, column) inside its image and the address of its channel vector
  struct Pix { int hh, ww; const unsigned short* p; };
  auto decode = [&](int strip) {
    Pix q; const int m = strip * 32 + r; const int rem = m % HW;
    q.hh = rem / W; q.ww = rem - q.hh * W; q.p = X + (int64_t)m * K + h * 8; return q;
  };
  // Loads are UNCONDITIONAL (a tap outside the image reads the pixel itself and is zeroed when it is consumed): a load
  // under a lane-dependent branch makes the compiler wait for every outstanding load (vmcnt(0)) at each use, which
  // throws away the one-strip prefetch distance and costs ~2 us of memory latency per strip.
  auto tap_ok = [&](const Pix& q, int tap) {
    const int dr = tap / 3 - 1, ds = tap % 3 - 1;
    return (unsigned)(q.hh + dr) < (unsigned)H && (unsigned)(q.ww + ds) < (unsigned)W;
  };
  auto load_tap = [&](const Pix& q, int tap, bf16x8_t (&dst)[4]) {
    const int dr = tap / 3 - 1, ds = tap % 3 - 1;
    const unsigned short* src = q.p + (tap_ok(q, tap) ? (dr * W + ds) * K : 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) dst[ks] = *(const bf16x8_t*)(src + ks * 16);
  };
  // A fragments of one tap (2 n-tiles x 4 k-steps), read a tap ahead of the MFMAs that use them
  auto load_w = [&](int tap, bf16x8_t (&dst)[8]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) dst[t * 4 + ks] = *(const bf16x8_t*)(Ws + (tap * N + t * 32 + r) * WLD + h * 8 + ks * 16);
  };
  Pix qc; qc.hh = 0; qc.ww = 0; qc.p = X + h * 8;
  if (s < nstrips) qc = decode(s);
#pragma unroll
  for (int j = 0; j < R; ++j) load_tap(qc, j, ring[j]);
  bf16x8_t wf[2][8];
  load_w(0, wf[0]);
  for (; s < nstrips; s += stride) {
    const int sn = s + stride;
    Pix qn = qc;                                                  // past the end: harmless re-reads of the current pixel
    if (sn < nstrips) qn = decode(sn);
    f32x16_t acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[t][q] = 0.0f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      load_w((tap + 1) % 9, wf[(tap + 1) & 1]);                   // next tap's weights (tap 0 of the next strip after tap 8)
      __builtin_amdgcn_sched_barrier(0);                          // keep these LDS reads HERE: the scheduler otherwise sinks each one
                                                                  // to just before its MFMA and every MFMA pair eats the LDS latency
      const bool ok = tap_ok(qc, tap);
      bf16x8_t xz[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
        xz[ks] = ok ? ring[tap][ks] : zero;
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t)                               // the two accumulator chains alternate
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[tap & 1][t * 4 + ks], xz[ks], acc[t], 0, 0, 0);
      load_tap(qn, tap, ring[tap]);                               // the slot just consumed takes the next strip's same tap
    }
    qc = qn;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2_t pk;
        pk.x = (unsigned int)c1_f2bf(acc[t][4 * g + 0]) | ((unsigned int)c1_f2bf(acc[t][4 * g + 1]) << 16);
        pk.y = (unsigned int)c1_f2bf(acc[t][4 * g + 2]) | ((unsigned int)c1_f2bf(acc[t][4 * g + 3]) << 16);
        *(u32x2_t*)(Ys + r * YLD + t * 32 + 8 * g + 4 * h) = pk;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int cc = lane & 7, r0 = lane >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = r0 + 8 * i;
      const u32x4_t v = *(const u32x4_t*)(Ys + row * YLD + cc * 8);
      const unsigned int w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = c1_bf2f((unsigned short)(w4[j >> 1] >> ((j & 1) * 16)));
        if (STATS) { st_s[j] += f; st_q[j] += f * f; }
      }
      __builtin_nontemporal_store(v, (u32x4_t*)(Y + ((int64_t)s * 32 + row) * N + cc * 8));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (!STATS) return;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float a = st_s[j], b = st_q[j];
    a += __shfl_xor(a, 8, 64); b += __shfl_xor(b, 8, 64);
    a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
    a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
    st_s[j] = a; st_q[j] = b;
  }
  __syncthreads();
  if (lane < 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { Ss[(wave * 2 + 0) * N + lane * 8 + j] = st_s[j]; Ss[(wave * 2 + 1) * N + lane * 8 + j] = st_q[j]; }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * N; e += kC1Threads) {
    const int which = e / N, n = e - which * N;
    part[(int64_t)blockIdx.x * 2 * N + which * N + n] =
        Ss[(0 * 2 + which) * N + n] + Ss[(1 * 2 + which) * N + n] + Ss[(2 * 2 + which) * N + n] + Ss[(3 * 2 + which) * N + n];
  }
}

// 3x3 / stride 1 / pad 1, 64 -> 64, LDS-halo form.  A wave owns 4 x 8-pixel output tiles (H % 4 == 0, W % 8 == 0).  It loads
// the tile's 6 x 10 input halo ONCE -- 60 pixel lines as coalesced 16-byte-per-lane loads, 8 lanes per line, prefetched a
// tile ahead into registers -- into a per-wave LDS image (zeros outside the image), and the nine taps read their B
// fragments from it with ds_read_b128: 60 L1 line fetches per tile instead of the 1 152 of the strip kernel above.
// A workgroup is EIGHT waves sharing the 83 KB of weights (two waves per SIMD: one wave's halo traffic and epilogue run
// under the other's MFMAs; 182 -> 156 us against four waves; sharing each weight fragment between two sub-tiles of an
// 8 x 8 tile instead, which needs the LDS of the second wave set, gave 169 us).
constexpr int kHaloPix = 60, kHaloLd = 64 + 8;                   // bf16 elements per halo pixel (16-byte pad)
template <bool STATS>
__global__ __launch_bounds__(512) void conv3x3_c64_halo_kernel(const unsigned short* __restrict__ X,
                                                                      const unsigned short* __restrict__ Wt, int Nimg, int H, int W,
                                                                      unsigned short* __restrict__ Y, float* __restrict__ part, int wtrans) {
  constexpr int K = 64, N = 64, WLD = K + 8, YLD = 64 + 8;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  unsigned short* Ws = smem;                                                   // [9][64][WLD]
  unsigned short* Hs = Ws + 9 * N * WLD + (threadIdx.x >> 6) * kHaloPix * kHaloLd;   // per wave [60][kHaloLd]
  unsigned short* Ys = Hs;                                                     // output tile over the (consumed) halo image
  float* Ss = (float*)(Ws + 9 * N * WLD);                                      // end of the launch only
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  for (int e = threadIdx.x; e < 9 * N * (K / 8); e += 512) {
    const int c8 = e % (K / 8); int t = e / (K / 8);
    const int co = t % N; const int tap = t / N;
    const u32x4_t v = *(const u32x4_t*)(Wt + ((int64_t)co * 9 + tap) * K + c8 * 8);
    if (!wtrans) *(u32x4_t*)(Ws + (tap * N + co) * WLD + c8 * 8) = v;
    else {       // forward weight [co][r][s][ci] of the layer whose data gradient this is: W'[ci][2-r][2-s][co], flipped and transposed here
      const unsigned int w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) Ws[((8 - tap) * N + c8 * 8 + j) * WLD + co] = (unsigned short)(w4[j >> 1] >> ((j & 1) * 16));
    }
  }
  __syncthreads();
  float st_s[8], st_q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { st_s[j] = 0.0f; st_q[j] = 0.0f; }

  const int tx_n = W / 8, ty_n = H / 4;
  const int ntiles = Nimg * ty_n * tx_n;
  const int stride = (int)gridDim.x * 8;
  const int nx = 8, per = (int)gridDim.x / nx;                                 // XCD-contiguous tile ranges (see above)
  const int lb = (per > 0 && (int)gridDim.x % nx == 0) ? ((int)blockIdx.x % nx) * per + (int)blockIdx.x / nx : (int)blockIdx.x;
  int t = lb * 8 + wave;
  // halo element e = lane + 64 i (i < 8): pixel e >> 3 (0..59, row-major 6 x 10), 16-byte chunk e & 7
  struct Tile { int n, y0, x0; };
  auto decode = [&](int tt) { Tile q; q.x0 = (tt % tx_n) * 8; const int u = tt / tx_n; q.y0 = (u % ty_n) * 4; q.n = u / ty_n; return q; };
  u32x4_t hreg[8];
  unsigned int hmask = 0;                                                       // bit i: element i of this lane lies inside the image
  auto load_halo = [&](const Tile& q) {
    hmask = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = lane + 64 * i, pix = e >> 3, ch = e & 7;
      const int hy = pix / 10, hx = pix - hy * 10;
      const int yy = q.y0 + hy - 1, xx = q.x0 + hx - 1;
      const bool ok = pix < kHaloPix && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
      const int64_t row = ok ? ((int64_t)q.n * H + yy) * W + xx : ((int64_t)q.n * H + q.y0) * W + q.x0;   // always a valid address
      hreg[i] = *(const u32x4_t*)(X + row * K + ch * 8);
      hmask |= (ok ? 1u : 0u) << i;
    }
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = lane + 64 * i, pix = e >> 3, ch = e & 7;
      if (pix < kHaloPix) {
        u32x4_t v = hreg[i];
        if (!((hmask >> i) & 1u)) { v.x = 0; v.y = 0; v.z = 0; v.w = 0; }
        *(u32x4_t*)(Hs + pix * kHaloLd + ch * 8) = v;
      }
    }
  };
  auto load_w = [&](int tap, bf16x8_t (&dst)[8]) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) dst[tt * 4 + ks] = *(const bf16x8_t*)(Ws + (tap * N + tt * 32 + r) * WLD + h * 8 + ks * 16);
  };
  const int py = r >> 3, px = r & 7;                                            // this lane's output pixel inside the tile
  Tile qc; qc.n = 0; qc.y0 = 0; qc.x0 = 0;
  if (t < ntiles) qc = decode(t);
  load_halo(qc);
  bf16x8_t wf[2][8];
  load_w(0, wf[0]);
  for (; t < ntiles; t += stride) {
    store_halo();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int tn = t + stride;
    Tile qn = qc;
    if (tn < ntiles) qn = decode(tn);
    load_halo(qn);                                                              // next tile's halo, in flight during this tile's taps
    f32x16_t acc[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[tt][q] = 0.0f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      load_w((tap + 1) % 9, wf[(tap + 1) & 1]);
      bf16x8_t xb[4];
      const unsigned short* hp = Hs + ((py + tap / 3) * 10 + px + tap % 3) * kHaloLd + h * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) xb[ks] = *(const bf16x8_t*)(hp + ks * 16);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
          acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[tap & 1][tt * 4 + ks], xb[ks], acc[tt], 0, 0, 0);
    }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2_t pk;
        pk.x = (unsigned int)c1_f2bf(acc[tt][4 * g + 0]) | ((unsigned int)c1_f2bf(acc[tt][4 * g + 1]) << 16);
        pk.y = (unsigned int)c1_f2bf(acc[tt][4 * g + 2]) | ((unsigned int)c1_f2bf(acc[tt][4 * g + 3]) << 16);
        *(u32x2_t*)(Ys + r * YLD + tt * 32 + 8 * g + 4 * h) = pk;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int cc = lane & 7, r0 = lane >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = r0 + 8 * i;                                               // tile pixel (row >> 3, row & 7)
      const u32x4_t v = *(const u32x4_t*)(Ys + row * YLD + cc * 8);
      const unsigned int w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = c1_bf2f((unsigned short)(w4[j >> 1] >> ((j & 1) * 16)));
        if (STATS) { st_s[j] += f; st_q[j] += f * f; }
      }
      const int64_t orow = ((int64_t)qc.n * H + qc.y0 + (row >> 3)) * W + qc.x0 + (row & 7);
      __builtin_nontemporal_store(v, (u32x4_t*)(Y + orow * N + cc * 8));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    qc = qn;
  }
  if (!STATS) return;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float a = st_s[j], b = st_q[j];
    a += __shfl_xor(a, 8, 64); b += __shfl_xor(b, 8, 64);
    a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
    a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
    st_s[j] = a; st_q[j] = b;
  }
  __syncthreads();
  if (lane < 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { Ss[(wave * 2 + 0) * N + lane * 8 + j] = st_s[j]; Ss[(wave * 2 + 1) * N + lane * 8 + j] = st_q[j]; }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * N; e += 512) {
    const int which = e / N, n = e - which * N;
    float a = 0.0f;
#pragma unroll
    for (int wv = 0; wv < 8; ++wv) a += Ss[(wv * 2 + which) * N + n];
    part[(int64_t)blockIdx.x * 2 * N + which * N + n] = a;
  }
}


}  // namespace lec

extern "C" int lec_conv3x3_c64_fwd(const void* x, const void* w, int w_transposed, int Nimg, int H, int W, void* y, float* partials,
                                   int64_t partials_bytes, int* n_partials, lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(x && w && y, "conv3x3_c64_fwd: null pointer");
  const int64_t M = (int64_t)Nimg * H * W;
  LEC_CHECK_ARG(Nimg > 0 && H > 0 && W > 0 && M % 32 == 0, "conv3x3_c64_fwd: N*H*W must be a positive multiple of 32");
  LEC_CHECK_ARG((partials == nullptr) == (n_partials == nullptr), "conv3x3_c64_fwd: pass partials and n_partials together");
  LEC_CHECK_ARG(!partials || partials_bytes >= (int64_t)kC1MaxBlocks * 2 * 64 * (int64_t)sizeof(float), "conv3x3_c64_fwd: partials buffer too small");
  const size_t smem = ((size_t)9 * 64 * (64 + 8) + 4 * 32 * (64 + 8)) * sizeof(unsigned short);
  static std::atomic<bool> attr_set{false};
  if (!attr_set.load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3_c64_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_c64_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv3x3_c64)");
    attr_set.store(true, std::memory_order_release);
  }
  const int64_t nstrips = M / 32;
  int64_t nb = (nstrips + 3) / 4;
  const int nblk = (int)(nb > kC1MaxBlocks ? kC1MaxBlocks : nb);
  hipStream_t st = (hipStream_t)stream;
  if (H % 4 == 0 && W % 8 == 0 && !tuning().c3_strip) {             // 4 x 8-pixel tiles: the LDS-halo kernel
    const size_t hsm = ((size_t)9 * 64 * (64 + 8) + 8 * kHaloPix * kHaloLd) * sizeof(unsigned short);
    static std::atomic<bool> hattr8{false};
    if (!hattr8.load(std::memory_order_acquire)) {
      hipError_t e = hipFuncSetAttribute((const void*)conv3x3_c64_halo_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hsm);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_c64_halo_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hsm);
      if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv3x3_c64_halo_w8)");
      hattr8.store(true, std::memory_order_release);
    }
    const int64_t nt8 = (int64_t)Nimg * (H / 4) * (W / 8);
    int64_t nb8 = (nt8 + 7) / 8;
    const int nblk8 = (int)(nb8 > 256 ? 256 : nb8);
    if (partials) hipLaunchKernelGGL((conv3x3_c64_halo_kernel<true>), dim3(nblk8), dim3(512), hsm, st, (const unsigned short*)x, (const unsigned short*)w, Nimg, H, W, (unsigned short*)y, partials, w_transposed);
    else hipLaunchKernelGGL((conv3x3_c64_halo_kernel<false>), dim3(nblk8), dim3(512), hsm, st, (const unsigned short*)x, (const unsigned short*)w, Nimg, H, W, (unsigned short*)y, (float*)nullptr, w_transposed);
    if (n_partials) *n_partials = nblk8;
    LEC_CHECK_LAUNCH("conv3x3_c64_halo_kernel");
    return LEC_OK;
  }
  if (partials) hipLaunchKernelGGL((conv3x3_c64_kernel<true>), dim3(nblk), dim3(kC1Threads), smem, st, (const unsigned short*)x, (const unsigned short*)w, M, H, W, (unsigned short*)y, partials, w_transposed);
  else hipLaunchKernelGGL((conv3x3_c64_kernel<false>), dim3(nblk), dim3(kC1Threads), smem, st, (const unsigned short*)x, (const unsigned short*)w, M, H, W, (unsigned short*)y, (float*)nullptr, w_transposed);
  if (n_partials) *n_partials = nblk;
  LEC_CHECK_LAUNCH("conv3x3_c64_kernel");
  return LEC_OK;
}

namespace lec {
// 3x3 / stride 1 / pad 1, 128 -> 128 (layer2's conv2 at 28x28; ResNet-18's layer2).  295 KB of weights do not fit LDS, so
// they stream through it one tap at a time: the workgroup's 4 waves walk the nine taps in lockstep, each tap's 128 x 128
// slice (32 KB) double-buffered in LDS -- fetched from L2 into registers two taps ahead, written one tap ahead, one
// workgroup barrier per tap -- while every wave keeps its own 4 x 8-pixel tile's halo (60 pixels x 128 channels) in LDS
// and all four 32-channel accumulator tiles in registers, so that a B fragment read serves four MFMAs.
// Tiles may hang over the right / bottom edge (28 = 3.5 x 8): such pixels read zeros, store nothing and stay out of the
// statistics.
template <bool STATS>
__global__ __launch_bounds__(kC1Threads) void conv3x3_c128_kernel(const unsigned short* __restrict__ X,
                                                                  const unsigned short* __restrict__ Wt, int Nimg, int H, int W,
                                                                  unsigned short* __restrict__ Y, float* __restrict__ part) {
  constexpr int C = 128, LD = C + 8, NT = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  unsigned short* Wb = smem;                                                   // [2][128][LD]
  unsigned short* Hs = Wb + 2 * C * LD + (threadIdx.x >> 6) * kHaloPix * LD;   // per wave [60][LD]; reused as the [32][LD] output tile
  float* Ss = (float*)(Wb + 2 * C * LD);                                       // end of the launch only
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int py = r >> 3, px = r & 7;
  float st_s[8], st_q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { st_s[j] = 0.0f; st_q[j] = 0.0f; }

  const int tx_n = (W + 7) / 8, ty_n = (H + 3) / 4;
  const int ntiles = Nimg * ty_n * tx_n;
  const int nrounds = (ntiles + 3) / 4;                                         // a round = one tile per wave
  const int nx = 8, per = (int)gridDim.x / nx;
  const int lb = (per > 0 && (int)gridDim.x % nx == 0) ? ((int)blockIdx.x % nx) * per + (int)blockIdx.x / nx : (int)blockIdx.x;
  struct Tile { int n, y0, x0; };
  auto decode = [&](int tt) { Tile q; q.x0 = (tt % tx_n) * 8; const int u = tt / tx_n; q.y0 = (u % ty_n) * 4; q.n = u / ty_n; return q; };
  // halo: 60 pixels x 16 chunks of 16 B = 960 elements = 15 per lane
  u32x4_t hreg[15];
  unsigned int hmask = 0;
  auto load_halo = [&](const Tile& q, bool live) {
    hmask = 0;
#pragma unroll
    for (int i = 0; i < 15; ++i) {
      const int e = lane + 64 * i, pix = e >> 4, ch = e & 15;
      const int hy = pix / 10, hx = pix - hy * 10;
      const int yy = q.y0 + hy - 1, xx = q.x0 + hx - 1;
      const bool ok = live && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
      const int64_t row = ok ? ((int64_t)q.n * H + yy) * W + xx : 0;
      hreg[i] = *(const u32x4_t*)(X + row * C + ch * 8);
      hmask |= (ok ? 1u : 0u) << i;
    }
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int i = 0; i < 15; ++i) {
      const int e = lane + 64 * i, pix = e >> 4, ch = e & 15;
      u32x4_t v = hreg[i];
      if (!((hmask >> i) & 1u)) { v.x = 0; v.y = 0; v.z = 0; v.w = 0; }
      *(u32x4_t*)(Hs + pix * LD + ch * 8) = v;
    }
  };
  // one tap's weights: 128 rows x 16 chunks = 2048 elements of 16 B = 8 per thread
  u32x4_t wreg[8];
  auto load_wtap = [&](int tap) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = threadIdx.x + kC1Threads * i, co = e >> 4, c8 = e & 15;
      wreg[i] = *(const u32x4_t*)(Wt + ((int64_t)co * 9 + tap) * C + c8 * 8);
    }
  };
  auto store_wtap = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = threadIdx.x + kC1Threads * i, co = e >> 4, c8 = e & 15;
      *(u32x4_t*)(Wb + (buf * C + co) * LD + c8 * 8) = wreg[i];
    }
  };

  int rd = lb;                                                                  // this workgroup's round
  const int rstride = (int)gridDim.x;
  Tile qc; qc.n = 0; qc.y0 = 0; qc.x0 = 0;
  bool live = rd < nrounds && rd * 4 + wave < ntiles;
  if (live) qc = decode(rd * 4 + wave);
  load_halo(qc, live);
  load_wtap(0); store_wtap(0);                                                  // tap 0 -> buffer 0
  load_wtap(1);                                                                 // tap 1 in registers
  int cur = 0;                                                                  // buffer holding the tap being consumed (9 taps: the parity flips every round)
  for (; rd < nrounds; rd += rstride) {
    store_halo();
    const int rn = rd + rstride;
    Tile qn = qc; bool live_n = rn < nrounds && rn * 4 + wave < ntiles;
    if (live_n) qn = decode(rn * 4 + wave);
    load_halo(qn, live_n);
    f32x16_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[t][q] = 0.0f;
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {      // not unrolled: nine unrolled taps push the kernel past 512 VGPRs
      __syncthreads();                       // every wave is past tap - 1: buffer (tap+1)&1 is free, buffer tap&1 is complete
      store_wtap(cur ^ 1);                   // weights of tap + 1 (tap 0 of the next round after tap 8)
      load_wtap(tap + 2 >= 9 ? tap + 2 - 9 : tap + 2);
      const unsigned short* wbase = Wb + (cur * C + r) * LD + h * 8;
      const int tr = tap >= 6 ? 2 : (tap >= 3 ? 1 : 0);
      const unsigned short* hp = Hs + ((py + tr) * 10 + px + tap - 3 * tr) * LD + h * 8;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const bf16x8_t b = *(const bf16x8_t*)(hp + ks * 16);
        bf16x8_t a[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) a[t] = *(const bf16x8_t*)(wbase + t * 32 * LD + ks * 16);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t], b, acc[t], 0, 0, 0);
      }
      cur ^= 1;
    }
    // ---- epilogue: D'[n][m] -> the wave's LDS tile (over its halo, which the taps are done with) -> rows out
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2_t pk;
        pk.x = (unsigned int)c1_f2bf(acc[t][4 * g + 0]) | ((unsigned int)c1_f2bf(acc[t][4 * g + 1]) << 16);
        pk.y = (unsigned int)c1_f2bf(acc[t][4 * g + 2]) | ((unsigned int)c1_f2bf(acc[t][4 * g + 3]) << 16);
        *(u32x2_t*)(Hs + r * LD + t * 32 + 8 * g + 4 * h) = pk;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int cc = lane & 15, r0 = lane >> 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = r0 + 4 * i;                                               // tile pixel (row >> 3, row & 7)
      const int oy = qc.y0 + (row >> 3), ox = qc.x0 + (row & 7);
      const bool ok = live && oy < H && ox < W;
      const u32x4_t v = *(const u32x4_t*)(Hs + row * LD + cc * 8);
      if (ok) {
        const unsigned int w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = c1_bf2f((unsigned short)(w4[j >> 1] >> ((j & 1) * 16)));
          if (STATS) { st_s[j] += f; st_q[j] += f * f; }
        }
        __builtin_nontemporal_store(v, (u32x4_t*)(Y + (((int64_t)qc.n * H + oy) * W + ox) * C + cc * 8));
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    qc = qn; live = live_n;
  }
  if (!STATS) return;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float a = st_s[j], b = st_q[j];
    a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
    a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
    st_s[j] = a; st_q[j] = b;
  }
  __syncthreads();
  if (lane < 16) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { Ss[(wave * 2 + 0) * C + lane * 8 + j] = st_s[j]; Ss[(wave * 2 + 1) * C + lane * 8 + j] = st_q[j]; }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * C; e += kC1Threads) {
    const int which = e / C, n = e - which * C;
    part[(int64_t)blockIdx.x * 2 * C + which * C + n] =
        Ss[(0 * 2 + which) * C + n] + Ss[(1 * 2 + which) * C + n] + Ss[(2 * 2 + which) * C + n] + Ss[(3 * 2 + which) * C + n];
  }
}

}  // namespace lec

extern "C" int lec_conv3x3_c128_fwd(const void* x, const void* w, int Nimg, int H, int W, void* y, float* partials,
                                    int64_t partials_bytes, int* n_partials, lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(x && w && y, "conv3x3_c128_fwd: null pointer");
  LEC_CHECK_ARG(Nimg > 0 && H > 0 && W > 0, "conv3x3_c128_fwd: bad sizes");
  LEC_CHECK_ARG((partials == nullptr) == (n_partials == nullptr), "conv3x3_c128_fwd: pass partials and n_partials together");
  LEC_CHECK_ARG(!partials || partials_bytes >= (int64_t)kC1MaxBlocks * 2 * 128 * (int64_t)sizeof(float), "conv3x3_c128_fwd: partials buffer too small");
  const size_t smem = ((size_t)2 * 128 * 136 + 4 * kHaloPix * 136) * sizeof(unsigned short);
  static std::atomic<bool> attr_set{false};
  if (!attr_set.load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3_c128_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_c128_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv3x3_c128)");
    attr_set.store(true, std::memory_order_release);
  }
  const int64_t ntiles = (int64_t)Nimg * ((H + 3) / 4) * ((W + 7) / 8);
  int64_t nb = (ntiles + 3) / 4;
  const int nblk = (int)(nb > kC1MaxBlocks ? kC1MaxBlocks : nb);
  hipStream_t st = (hipStream_t)stream;
  if (partials) hipLaunchKernelGGL((conv3x3_c128_kernel<true>), dim3(nblk), dim3(kC1Threads), smem, st, (const unsigned short*)x, (const unsigned short*)w, Nimg, H, W, (unsigned short*)y, partials);
  else hipLaunchKernelGGL((conv3x3_c128_kernel<false>), dim3(nblk), dim3(kC1Threads), smem, st, (const unsigned short*)x, (const unsigned short*)w, Nimg, H, W, (unsigned short*)y, (float*)nullptr);
  if (n_partials) *n_partials = nblk;
  LEC_CHECK_LAUNCH("conv3x3_c128_kernel");
  return LEC_OK;
}

namespace lec {
// ---------------------------------------------------------------------------------------------------------------
// Weight gradient of the wide 1x1 convolutions:  dW[co][ci] += sum_m dY[m][co] * X[m][ci]   (fp32, accumulated straight into
// the parameter's gradient slot of the flat arena: no zero-fill, no fp32 -> bf16 cast, no copy kernel around it).
// HBM-bound like the forward (dY and X are read once, the output is a few KB).  The reduction index m is the slow memory
// index of BOTH operands, while an MFMA fragment wants 8 consecutive m per lane: the workgroup stages 64-row chunks of dY
// and X through LDS TRANSPOSED -- coalesced 16-byte global loads, eight 2-byte LDS stores each into [channel][m] images
// whose 16-byte groups are XOR-swizzled by the channel-octet index (4-way instead of 32-way bank conflicts on the stores)
// -- double-buffered, one barrier per chunk.  A wave owns a TCO x TCI block of 32 x 32 accumulator tiles for the whole
// launch and adds it into dW with float atomics at the end (rows of 32 consecutive floats per instruction).
// KO / KI: output / input channels of this workgroup (blockIdx.y selects the KO- or KI-wide window of a wider layer).
// XF (conv3 behind bn3): dY does not exist yet.  The kernel is handed g (the masked gradient, pass 1's output) and the
// BatchNorm's input instead, forms pass 2 -- dx = gamma invstd (g - c1 - xhat c2), rounded to bf16, bn_bwd_apply_kernel's
// arithmetic -- on the 16-byte pieces while they sit in registers between the coalesced load and the transposed LDS store,
// writes dx out for the data-gradient kernel (coalesced, the same addresses) and uses it as its own dY operand.  The separate
// pass 2 (read g, x; write dx) and this kernel's read of dx go away: one read of a block-output-sized tensor less per layer.
struct XfArgs {
  const unsigned short* xbn;      // [M][CoutTot] BatchNorm input
  const float *gamma, *mean, *invstd, *c1, *c2;   // [CoutTot]
  unsigned short* dx;             // [M][CoutTot] out
};

template <int KO, int KI, int WCO, int NW, bool SPLIT_CO, bool XF>
__global__ __launch_bounds__(NW * 64) void wgrad1x1_kernel(const unsigned short* __restrict__ dY, int CoutTot,
                                                           const unsigned short* __restrict__ X, int CinTot, int64_t M,
                                                           float* __restrict__ dW, XfArgs xf) {
  constexpr int MC = 64, NT = NW * 64;                 // rows per chunk, threads
  constexpr int WCI = NW / WCO, TCO = KO / 32 / WCO, TCI = KI / 32 / WCI;
  constexpr int NC8 = (KO + KI) / 8;                   // 16-byte pieces per row
  constexpr int PAIRS = NC8 * (MC / 2) / NT;           // (piece of row m, piece of row m + 1) pairs per thread and chunk
  static_assert(NC8 % 8 == 0 && NC8 * (MC / 2) % NT == 0 && TCO >= 1 && TCI >= 1 && WCO * WCI == NW, "tile split");
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];   // [2][(KO + KI)][MC]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int co0 = SPLIT_CO ? (int)blockIdx.y * KO : 0, ci0 = SPLIT_CO ? 0 : (int)blockIdx.y * KI;
  const int wco = wave / WCI, wci = wave % WCI;        // this wave's block of tiles
  f32x16_t acc[TCO][TCI];
#pragma unroll
  for (int a = 0; a < TCO; ++a)
#pragma unroll
    for (int b = 0; b < TCI; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.0f;

  // Pair q of a chunk: channel octet c8 = 8 * c8hi + (q & 7), rows m = 8 * mhi + 2 * ((q >> 3) & 3) + {0, 1}.  The 32 lanes
  // of a half-wave hold 8 octets x 4 row pairs: four 128-byte row segments per global load, and in the LDS image
  // (element (n, m) at n * MC + (((m >> 3) ^ ((n >> 3) & 7)) << 3) + (m & 7)) their 4-byte stores fall on 32 different banks.
  const int64_t nchunks = M / MC;
  u32x4_t regs[PAIRS][2];
  u32x4_t regx[XF ? PAIRS : 1][2];                       // XF: the BatchNorm input at the dY pieces' addresses
  float* Cs = (float*)(smem + 2 * (KO + KI) * MC);       // XF: [5][KO] gamma*invstd, mean, invstd, c1, c2 of this workgroup's channels
  if (XF) {
    for (int e = threadIdx.x; e < KO; e += NT) {
      const float is = xf.invstd[co0 + e];
      Cs[e] = xf.gamma[co0 + e] * is; Cs[KO + e] = xf.mean[co0 + e]; Cs[2 * KO + e] = is;
      Cs[3 * KO + e] = xf.c1[co0 + e]; Cs[4 * KO + e] = xf.c2[co0 + e];
    }
    __syncthreads();
  }
  auto load_chunk = [&](int64_t c) {
#pragma unroll
    for (int i = 0; i < PAIRS; ++i) {
      const int q = threadIdx.x + NT * i, rest = q >> 5;
      const int c8 = (rest % (NC8 / 8)) * 8 + (q & 7), m = (rest / (NC8 / 8)) * 8 + ((q >> 3) & 3) * 2;
      const int64_t row = c * MC + m;
      const unsigned short* src = c8 < KO / 8 ? dY + row * CoutTot + co0 + c8 * 8 : X + row * CinTot + ci0 + (c8 - KO / 8) * 8;
      const int ld = c8 < KO / 8 ? CoutTot : CinTot;
      regs[i][0] = *(const u32x4_t*)src;
      regs[i][1] = *(const u32x4_t*)(src + ld);
      if (XF) {                                          // (the X pieces re-read their own address: every load stays unconditional)
        const unsigned short* sx = c8 < KO / 8 ? xf.xbn + row * CoutTot + co0 + c8 * 8 : src;
        regx[i][0] = *(const u32x4_t*)sx;
        regx[i][1] = *(const u32x4_t*)(sx + ld);
      }
    }
  };
  auto store_chunk = [&](int buf, int64_t c) {
    unsigned short* T = smem + buf * (KO + KI) * MC;
#pragma unroll
    for (int i = 0; i < PAIRS; ++i) {
      const int q = threadIdx.x + NT * i, rest = q >> 5;
      const int c8 = (rest % (NC8 / 8)) * 8 + (q & 7), m = (rest / (NC8 / 8)) * 8 + ((q >> 3) & 3) * 2;
      unsigned int a4[4] = {regs[i][0].x, regs[i][0].y, regs[i][0].z, regs[i][0].w};
      unsigned int b4[4] = {regs[i][1].x, regs[i][1].y, regs[i][1].z, regs[i][1].w};
      if (XF && c8 < KO / 8) {
        const unsigned int xa[4] = {regx[i][0].x, regx[i][0].y, regx[i][0].z, regx[i][0].w};
        const unsigned int xb[4] = {regx[i][1].x, regx[i][1].y, regx[i][1].z, regx[i][1].w};
        unsigned int oa[4] = {0u, 0u, 0u, 0u}, ob[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int ch = c8 * 8 + j, sh = (j & 1) * 16;
          const float gs = Cs[ch], mu = Cs[KO + ch], is = Cs[2 * KO + ch], k1 = Cs[3 * KO + ch], k2 = Cs[4 * KO + ch];
          const float ga = c1_bf2f((unsigned short)(a4[j >> 1] >> sh)), gb = c1_bf2f((unsigned short)(b4[j >> 1] >> sh));
          const float ha = (c1_bf2f((unsigned short)(xa[j >> 1] >> sh)) - mu) * is, hb = (c1_bf2f((unsigned short)(xb[j >> 1] >> sh)) - mu) * is;
          oa[j >> 1] |= (unsigned int)c1_f2bf(gs * (ga - k1 - ha * k2)) << sh;
          ob[j >> 1] |= (unsigned int)c1_f2bf(gs * (gb - k1 - hb * k2)) << sh;
        }
        unsigned short* dst = xf.dx + (c * MC + m) * CoutTot + co0 + c8 * 8;
        u32x4_t va, vb;
        va.x = oa[0]; va.y = oa[1]; va.z = oa[2]; va.w = oa[3]; vb.x = ob[0]; vb.y = ob[1]; vb.z = ob[2]; vb.w = ob[3];
        __builtin_nontemporal_store(va, (u32x4_t*)dst);
        __builtin_nontemporal_store(vb, (u32x4_t*)(dst + CoutTot));
#pragma unroll
        for (int j = 0; j < 4; ++j) { a4[j] = oa[j]; b4[j] = ob[j]; }
      }
      unsigned int* base = (unsigned int*)(T + (c8 * 8) * MC + (((m >> 3) ^ (c8 & 7)) << 3) + (m & 7));
#pragma unroll
      for (int j = 0; j < 4; ++j) {                    // channels 2j, 2j + 1 of the octet: (row m, row m + 1) in one dword
        base[(2 * j) * (MC / 2)] = __builtin_amdgcn_perm(b4[j], a4[j], 0x05040100u);
        base[(2 * j + 1) * (MC / 2)] = __builtin_amdgcn_perm(b4[j], a4[j], 0x07060302u);
      }
    }
  };
  int64_t c = blockIdx.x;
  if (c < nchunks) { load_chunk(c); store_chunk(0, c); }
  __syncthreads();
  int buf = 0;
  for (; c < nchunks; c += gridDim.x) {
    const int64_t cn = c + gridDim.x;
    if (cn < nchunks) load_chunk(cn);
    const unsigned short* T = smem + buf * (KO + KI) * MC;
#pragma unroll
    for (int ks = 0; ks < MC / 16; ++ks) {
      const int g = 2 * ks + h;                        // 16-byte group of this lane's 8 consecutive m
      bf16x8_t af[TCO], bfr[TCI];
#pragma unroll
      for (int a = 0; a < TCO; ++a) {
        const int n = (wco * TCO + a) * 32 + r;
        af[a] = *(const bf16x8_t*)(T + n * MC + ((g ^ ((n >> 3) & 7)) << 3));
      }
#pragma unroll
      for (int b = 0; b < TCI; ++b) {
        const int n = KO + (wci * TCI + b) * 32 + r;
        bfr[b] = *(const bf16x8_t*)(T + n * MC + ((g ^ ((n >> 3) & 7)) << 3));
      }
#pragma unroll
      for (int a = 0; a < TCO; ++a)
#pragma unroll
        for (int b = 0; b < TCI; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
    }
    if (cn < nchunks) store_chunk(buf ^ 1, cn);
    __syncthreads();
    buf ^= 1;
  }
  // D[co][ci]: lane holds ci = col, co rows (reg & 3) + 8 (reg >> 2) + 4 h
#pragma unroll
  for (int a = 0; a < TCO; ++a)
#pragma unroll
    for (int b = 0; b < TCI; ++b) {
      float* base = dW + (int64_t)(co0 + (wco * TCO + a) * 32) * CinTot + ci0 + (wci * TCI + b) * 32 + r;
#pragma unroll
      for (int q = 0; q < 16; ++q) atomicAdd(base + (int64_t)((q & 3) + 8 * (q >> 2) + 4 * h) * CinTot, acc[a][b][q]);
    }
}

template <int KO, int KI, int WCO, int NW, bool SPLIT_CO, bool XF = false>
static int launch_wgrad1x1(const void* dy, int CoutTot, const void* x, int CinTot, int64_t M, float* dW, hipStream_t st,
                           XfArgs xf = XfArgs{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}) {
  const size_t smem = (size_t)2 * (KO + KI) * 64 * sizeof(unsigned short) + (XF ? 5 * KO * sizeof(float) : 0);
  static std::atomic<bool> attr_set{false};
  if (smem > 64 * 1024 && !attr_set.load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute((const void*)wgrad1x1_kernel<KO, KI, WCO, NW, SPLIT_CO, XF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(wgrad1x1)");
    attr_set.store(true, std::memory_order_release);
  }
  const int ny = SPLIT_CO ? CoutTot / KO : CinTot / KI;
  const int64_t nchunks = M / 64;
  // two workgroups per CU where their LDS images fit (measured: 5.8 against 5.7 TB/s with one; three: 5.4), one otherwise
  const int want = 256 * (smem <= 80 * 1024 ? 2 : 1) / ny;
  const int nblk = (int)(nchunks < want ? nchunks : want);
  hipLaunchKernelGGL((wgrad1x1_kernel<KO, KI, WCO, NW, SPLIT_CO, XF>), dim3(nblk, ny), dim3(NW * 64), smem, st, (const unsigned short*)dy, CoutTot,
                     (const unsigned short*)x, CinTot, M, dW, xf);
  LEC_CHECK_LAUNCH("wgrad1x1_kernel");
  return LEC_OK;
}

}  // namespace lec

extern "C" int lec_conv1x1_wgrad_bnapply_supported(int Cin, int Cout, int64_t M) {
  return ((Cin == 64 && Cout == 256) || (Cin == 128 && Cout == 512)) && M > 0 && M % 64 == 0;
}

extern "C" int lec_conv1x1_wgrad_bnapply(const void* g, const void* bn_x, const void* x, int64_t M, int Cin, int Cout, const float* gamma,
                                         const float* save_mean, const float* save_invstd, const float* c1, const float* c2, void* dx, float* dw,
                                         lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(g && bn_x && x && gamma && save_mean && save_invstd && c1 && c2 && dx && dw, "conv1x1_wgrad_bnapply: null pointer");
  LEC_CHECK_ARG(lec_conv1x1_wgrad_bnapply_supported(Cin, Cout, M), "conv1x1_wgrad_bnapply: unsupported shape Cin=%d Cout=%d M=%lld", Cin, Cout,
                (long long)M);
  const XfArgs xf{(const unsigned short*)bn_x, gamma, save_mean, save_invstd, c1, c2, (unsigned short*)dx};
  hipStream_t st = (hipStream_t)stream;
  // 64 -> 256: two 128-channel windows (x, a quarter of the traffic, is read twice): 51 KB of LDS and 204 registers instead of
  // the one-window form's 85 KB and 300 -- two workgroups per CU instead of one (step: 44.1 against 44.6 ms).  128 -> 512: two
  // 256-channel windows, eight waves (four 128-channel windows measured worse: 44.5 against 43.9 ms)
  if (Cin == 64) return launch_wgrad1x1<128, 64, 2, 4, true, true>(g, Cout, x, Cin, M, dw, st, xf);
  return launch_wgrad1x1<256, 128, 4, 8, true, true>(g, Cout, x, Cin, M, dw, st, xf);
}

extern "C" int lec_conv1x1_wgrad_supported(int Cin, int Cout, int64_t M) {
  const bool shape = (Cin == 64 && (Cout == 64 || Cout == 256)) || (Cin == 128 && Cout == 512) || (Cin == 256 && (Cout == 64 || Cout == 128)) ||
                     (Cin == 512 && Cout == 128) || (Cin == 256 && Cout == 1024) || (Cin == 1024 && Cout == 256);
  return shape && M > 0 && M % 64 == 0;
}

extern "C" int lec_conv1x1_wgrad(const void* dy, const void* x, int64_t M, int Cin, int Cout, float* dw, lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(dy && x && dw, "conv1x1_wgrad: null pointer");
  LEC_CHECK_ARG(lec_conv1x1_wgrad_supported(Cin, Cout, M), "conv1x1_wgrad: unsupported shape Cin=%d Cout=%d M=%lld", Cin, Cout, (long long)M);
  hipStream_t st = (hipStream_t)stream;
  if (Cin == 64 && Cout == 64) return launch_wgrad1x1<64, 64, 2, 4, true>(dy, Cout, x, Cin, M, dw, st);
  if (Cin == 64 && Cout == 256) return launch_wgrad1x1<256, 64, 4, 4, true>(dy, Cout, x, Cin, M, dw, st);
  if (Cin == 256 && Cout == 64) return launch_wgrad1x1<64, 256, 1, 4, true>(dy, Cout, x, Cin, M, dw, st);
  if (Cin == 256 && Cout == 128) return launch_wgrad1x1<128, 256, 4, 8, true>(dy, Cout, x, Cin, M, dw, st);
  if (Cin == 128 && Cout == 512) return launch_wgrad1x1<256, 128, 4, 8, true>(dy, Cout, x, Cin, M, dw, st);  // two 256-row halves of dW
  // layer3's 1x1 pairs (14 x 14): 256 x 256 windows of dW, four of them; the narrow operand is read once per window
  if (Cin == 256 && Cout == 1024) return launch_wgrad1x1<256, 256, 4, 8, true>(dy, Cout, x, Cin, M, dw, st);
  if (Cin == 1024 && Cout == 256) return launch_wgrad1x1<256, 256, 4, 8, false>(dy, Cout, x, Cin, M, dw, st);
  return launch_wgrad1x1<128, 256, 2, 8, false>(dy, Cout, x, Cin, M, dw, st);                                // 512 -> 128: two 256-column halves
}

namespace lec {
// ---------------------------------------------------------------------------------------------------------------
// Weight gradient of layer1's 3x3 convolution (64 -> 64, stride 1, pad 1, NHWC bf16):
//     dW[co][ky][kx][ci] += sum over images and pixels (y, x) of dY[y][x][co] * X[y + ky - 1][x + kx - 1][ci]        (fp32)
// i.e. nine [64 x 64] products whose reduction index is the pixel -- the slow memory index of both operands, as in the 1x1
// weight gradient above -- and whose X operand is the input shifted by the tap.  A workgroup (4 waves) walks 8 x 8-pixel tiles:
// the tile of dY and the tile's 10 x 10 input halo arrive by coalesced 16-byte loads (a tile ahead, in registers) and are
// stored TRANSPOSED into LDS, [channel][pixel], two horizontally adjacent pixels per dword store (16-byte groups XOR-swizzled by
// the channel octet: conflict-free stores and reads).  An MFMA fragment is 8 consecutive pixels of one tile row shifted by
// the tap: one aligned 16-byte read + one dword of the halo row serve kx = 0, 1, 2 (kx = 1 through v_alignbyte).  Eight
// waves: a wave owns a 32 x 32 block of (co, ci) for five or four of the nine taps -- its accumulator tiles for the whole
// launch -- and adds them into the parameter's gradient slot with float atomics at the end (memory [co][ky][kx][ci], the
// channels_last weight layout).
// Bytes: dY and X read once (2 x 128 B per pixel); the library kernel it replaces takes 577 us at the bench batch.
constexpr int kW3CiLd = 256;                                     // input image: 32 sixteen-byte groups per channel (10 halo rows x 2, swizzled)
constexpr int kW3BufElems = 64 * 64 + 64 * kW3CiLd;              // dY tile + input halo, one buffer

// Swizzles of the 16-byte groups of the two transposed images.  Stores come from lanes that differ in the channel OCTET (bits
// 5:3 of the channel; 8 lanes x 4 pixel pairs per half-wave), reads from lanes that hold 16 CONSECUTIVE channels (bits 3:0):
// both must land on distinct groups.
__device__ __forceinline__ int w3_dswz(int co) { return ((((co >> 3) & 1) << 2) | ((co >> 1) & 3)) ^ ((co >> 4) & 3); }   // 8 groups, rows of 128 B
__device__ __forceinline__ int w3_xswz(int ci) { return (ci & 15) ^ ((ci >> 4) & 3); }                                   // 32 groups, rows of 512 B

__global__ __launch_bounds__(512) void wgrad3x3_c64_kernel(const unsigned short* __restrict__ dY, const unsigned short* __restrict__ X,
                                                           int Nimg, int H, int W, float* __restrict__ dW) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];   // [2][ Dt[64][64] | Xt[64][kW3CiLd] ]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int coh = (wave >> 1) & 1, cih = wave & 1, tg = wave >> 2;   // 32 x 32 (co, ci) block; taps 0..4 (tg 0) or 5..8 (tg 1)
  const unsigned int tw = W / 8, tpi = (H / 8) * tw;             // tiles per row, per image
  const unsigned int ntiles = (unsigned int)Nimg * tpi;
  f32x16_t acc[5];
#pragma unroll
  for (int t = 0; t < 5; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[t][q] = 0.0f;

  // a thread stages one PAIR of horizontally adjacent pixels (8 channels each): the transposed images take them as one dword
  // per channel.  dY: 64 px = 32 pairs x 8 channel octets = 256 threads; halo: 10 rows x 5 pairs x 8 octets = 400 threads.
  const int tid = threadIdx.x, c8 = tid & 7;
  const int dpr = (tid >> 3) & 31;                               // dY pair: tile row dpr >> 2, first pixel (dpr & 3) * 2 (threads 256.. repeat)
  const int hq = tid >> 3, hr = hq / 5, hc = (hq - hr * 5) * 2;  // halo pair: row hr, first column hc (threads 400.. load clamped rows, store nothing)
  const int dlane = ((dpr >> 2) * W + (dpr & 3) * 2) * 64 + c8 * 8;   // element offset of the dY pair inside its tile
  constexpr int NP = 4;                                          // tiles in flight in registers: a tile's compute phase is far shorter than the memory latency
  u32x4_t rd[NP][2], rx[NP][2];
  unsigned int okm[NP];                                          // halo pair: bit 0 / 1 = first / second pixel inside the image
  // every load is unconditional, from a clamped address (a load under a lane-dependent branch makes the compiler drain the whole
  // load queue at the join: no prefetch left); pixels outside the image are zeroed when the set is stored
  auto load_tile = [&](unsigned int t, int set) {
    const unsigned int n = t / tpi, rem = t - n * tpi, trow = rem / tw;      // wave-uniform: scalar unit
    const int y0 = (int)trow * 8, x0 = (int)(rem - trow * tw) * 8;
    const int64_t img = (int64_t)n * H * W * 64;
    const unsigned short* src = dY + img + (int64_t)(y0 * W + x0) * 64 + dlane;
    rd[set][0] = *(const u32x4_t*)src;
    rd[set][1] = *(const u32x4_t*)(src + 64);
    const int yy = y0 - 1 + hr, xx = x0 - 1 + hc;
    const int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
    const int xa = xx < 0 ? 0 : xx, xb = xx + 1 >= W ? W - 1 : xx + 1;
    const unsigned short* rowp = X + img + (yc * W) * 64 + c8 * 8;
    rx[set][0] = *(const u32x4_t*)(rowp + xa * 64);
    rx[set][1] = *(const u32x4_t*)(rowp + xb * 64);
    const bool yok = yy >= 0 && yy < H;
    okm[set] = (yok && xx >= 0 ? 1u : 0u) | (yok && xx + 1 < W ? 2u : 0u);
  };
  auto store_tile = [&](int buf, int set) {
    unsigned int* Dt = (unsigned int*)(smem + buf * kW3BufElems);
    unsigned int* Xt = Dt + 64 * 64 / 2;
    if (tid < 256) {
      const unsigned int a4[4] = {rd[set][0].x, rd[set][0].y, rd[set][0].z, rd[set][0].w}, b4[4] = {rd[set][1].x, rd[set][1].y, rd[set][1].z, rd[set][1].w};
      // Dt element (co, px) at co * 64 + (((px >> 3) ^ w3_dswz(co)) << 3) + (px & 7); px = 8 (dpr >> 2) + 2 (dpr & 3)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int co = c8 * 8 + j;
        Dt[co * 32 + ((((dpr >> 2) ^ w3_dswz(co)) << 2)) + (dpr & 3)] = __builtin_amdgcn_perm(b4[j >> 1], a4[j >> 1], (j & 1) ? 0x07060302u : 0x05040100u);
      }
    }
    if (tid < 400) {
      const unsigned int m0 = (okm[set] & 1u) ? 0xffffffffu : 0u, m1 = (okm[set] & 2u) ? 0xffffffffu : 0u;
      const unsigned int a4[4] = {rx[set][0].x & m0, rx[set][0].y & m0, rx[set][0].z & m0, rx[set][0].w & m0};
      const unsigned int b4[4] = {rx[set][1].x & m1, rx[set][1].y & m1, rx[set][1].z & m1, rx[set][1].w & m1};
      // Xt element (ci, hr, hc) at ci * 256 + (((2 hr + (hc >> 3)) ^ w3_xswz(ci)) << 3) + (hc & 7)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ci = c8 * 8 + j;
        Xt[ci * (kW3CiLd / 2) + (((2 * hr + (hc >> 3)) ^ w3_xswz(ci)) << 2) + ((hc & 7) >> 1)] =
            __builtin_amdgcn_perm(b4[j >> 1], a4[j >> 1], (j & 1) ? 0x07060302u : 0x05040100u);
      }
    }
  };

  const unsigned int G = gridDim.x, t0 = blockIdx.x;
#pragma unroll
  for (int u = 0; u < NP; ++u)
    if (t0 + u * G < ntiles) load_tile(t0 + u * G, u);
  if (t0 < ntiles) store_tile(0, 0);
  __syncthreads();
  int buf = 0;
  const int co = coh * 32 + r, ci = cih * 32 + r;
  const int dswz = w3_dswz(co), xswz = w3_xswz(ci);
  bool more = t0 < ntiles;
  for (unsigned int tb = t0; more; tb += NP * G) {
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const unsigned int t = tb + u * G;                         // this tile sits in LDS buffer `buf`; its register set u is free again
      if (t >= ntiles) { more = false; break; }
      if (t + NP * G < ntiles) load_tile(t + NP * G, u);
      const unsigned short* Dt = smem + buf * kW3BufElems + co * 64;
      const unsigned short* Xt = smem + buf * kW3BufElems + 64 * 64 + ci * kW3CiLd;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int ty = 2 * ks + h;                               // tile row of this lane's 8 consecutive pixels
        const bf16x8_t a = *(const bf16x8_t*)(Dt + ((ty ^ dswz) << 3));
        // halo row ty + ky, columns kx .. kx + 7: dwords d0..d4 of the row (one 16-byte read + the first dword of the next group)
        auto row = [&](int ky, u32x4_t& w0, unsigned int& d4) {
          const int g = 2 * (ty + ky);
          w0 = *(const u32x4_t*)(Xt + ((g ^ xswz) << 3));
          d4 = *(const unsigned int*)(Xt + (((g + 1) ^ xswz) << 3));
        };
        auto frag = [&](const u32x4_t& w0, unsigned int d4, int kx) -> bf16x8_t {
          u32x4_t f;
          if (kx == 0) f = w0;
          else if (kx == 2) { f.x = w0.y; f.y = w0.z; f.z = w0.w; f.w = d4; }
          else {
            f.x = __builtin_amdgcn_alignbyte(w0.y, w0.x, 2); f.y = __builtin_amdgcn_alignbyte(w0.z, w0.y, 2);
            f.z = __builtin_amdgcn_alignbyte(w0.w, w0.z, 2); f.w = __builtin_amdgcn_alignbyte(d4, w0.w, 2);
          }
          return __builtin_bit_cast(bf16x8_t, f);
        };
        u32x4_t w0; unsigned int d4;
        if (tg == 0) {                                           // taps (0,0) (0,1) (0,2) (1,0) (1,1)
          row(0, w0, d4);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, frag(w0, d4, 0), acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, frag(w0, d4, 1), acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, frag(w0, d4, 2), acc[2], 0, 0, 0);
          row(1, w0, d4);
          acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, frag(w0, d4, 0), acc[3], 0, 0, 0);
          acc[4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, frag(w0, d4, 1), acc[4], 0, 0, 0);
        } else {                                                 // taps (1,2) (2,0) (2,1) (2,2)
          row(1, w0, d4);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, frag(w0, d4, 2), acc[0], 0, 0, 0);
          row(2, w0, d4);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, frag(w0, d4, 0), acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, frag(w0, d4, 1), acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, frag(w0, d4, 2), acc[3], 0, 0, 0);
        }
      }
      if (t + G < ntiles) store_tile(buf ^ 1, (u + 1) % NP);
      __syncthreads();
      buf ^= 1;
    }
  }
  // D[co][ci]: lane holds ci = cih*32 + r, co rows coh*32 + (q & 3) + 8 (q >> 2) + 4 h; memory [co][tap][ci]
  const int tap0 = tg == 0 ? 0 : 5, ntap = tg == 0 ? 5 : 4;
#pragma unroll
  for (int tp = 0; tp < 5; ++tp) {
    if (tp >= ntap) break;
    float* base = dW + (int64_t)(coh * 32) * 576 + (tap0 + tp) * 64 + cih * 32 + r;
#pragma unroll
    for (int q = 0; q < 16; ++q) atomicAdd(base + (int64_t)((q & 3) + 8 * (q >> 2) + 4 * h) * 576, acc[tp][q]);
  }
}

}  // namespace lec

extern "C" int lec_conv3x3_c64_wgrad_supported(int N, int H, int W) {
  return N > 0 && H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0 && (int64_t)H * W * 64 < (1ll << 31) && (int64_t)N * (H / 8) * (W / 8) < (1ll << 31);
}

extern "C" int lec_conv3x3_c64_wgrad(const void* dy, const void* x, int N, int H, int W, float* dw, lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(dy && x && dw, "conv3x3_c64_wgrad: null pointer");
  LEC_CHECK_ARG(lec_conv3x3_c64_wgrad_supported(N, H, W), "conv3x3_c64_wgrad: H and W must be multiples of 8 (N=%d H=%d W=%d)", N, H, W);
  const size_t smem = (size_t)2 * kW3BufElems * sizeof(unsigned short);
  static std::atomic<bool> attr_set{false};
  if (!attr_set.load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute((const void*)wgrad3x3_c64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(wgrad3x3_c64)");
    attr_set.store(true, std::memory_order_release);
  }
  const int64_t ntiles = (int64_t)N * (H / 8) * (W / 8);
  const int nblk = (int)(ntiles < 256 ? ntiles : 256);
  hipLaunchKernelGGL(wgrad3x3_c64_kernel, dim3(nblk), dim3(512), smem, (hipStream_t)stream, (const unsigned short*)dy, (const unsigned short*)x, N, H, W, dw);
  LEC_CHECK_LAUNCH("wgrad3x3_c64_kernel");
  return LEC_OK;
}

namespace lec {
}  // namespace lec

// (Cin, Cout) pairs with a kernel instance; M = N*H*W must be a multiple of 32
extern "C" int lec_conv1x1_supported(int Cin, int Cout, int64_t M) {
  const bool shape = (Cin == 64 && (Cout == 64 || Cout == 256)) || (Cin == 128 && (Cout == 256 || Cout == 512)) || (Cin == 256 && (Cout == 64 || Cout == 128)) || (Cin == 512 && Cout == 128);
  return shape && M > 0 && M % 32 == 0;
}

extern "C" int lec_conv1x1_dgrad_bnfold_supported(int Cin, int Cout, int64_t M) {
  return ((Cin == 64 && Cout == 256) || (Cin == 128 && (Cout == 512 || Cout == 256))) && M > 0 && M % 32 == 0;
}

extern "C" int lec_conv1x1_dgrad_bnfold(const void* dy, const void* w, int w_transposed, int64_t M, int Cin, int Cout, const void* dy2,
                                        const void* bn_x, const uint8_t* relu_mask, const float* save_mean, const float* save_invstd, void* g,
                                        float* partials, int64_t partials_bytes, int* n_partials, lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(dy && w && dy2 && bn_x && relu_mask && save_mean && save_invstd && g && partials && n_partials, "conv1x1_dgrad_bnfold: null pointer");
  LEC_CHECK_ARG(lec_conv1x1_dgrad_bnfold_supported(Cin, Cout, M), "conv1x1_dgrad_bnfold: unsupported shape Cin=%d Cout=%d M=%lld", Cin, Cout,
                (long long)M);
  LEC_CHECK_ARG(partials_bytes >= (int64_t)kC1MaxBlocks * 2 * Cout * (int64_t)sizeof(float), "conv1x1_dgrad_bnfold: partials buffer too small");
  const FoldArgs fa{(const unsigned short*)dy2, (const unsigned short*)bn_x, (const unsigned char*)relu_mask, save_mean, save_invstd};
  hipStream_t st = (hipStream_t)stream;
  if (Cin == 64) return launch_conv1x1<64, 256, true, 1>(dy, w, M, Cout, g, partials, n_partials, w_transposed, st, fa);
  return launch_conv1x1<128, 256, true, 1>(dy, w, M, Cout, g, partials, n_partials, w_transposed, st, fa);
}

extern "C" int lec_conv1x1_bnapply_supported(int Cin, int Cout, int64_t M) {
  return ((Cin == 64 && Cout == 256) || (Cin == 128 && Cout == 512)) && M > 0 && M % 32 == 0;
}

extern "C" int lec_conv1x1_stats(const void* x, const void* w, int64_t M, int Cin, int Cout, float* partials, int64_t partials_bytes,
                                 int* n_partials, lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(x && w && partials && n_partials, "conv1x1_stats: null pointer");
  LEC_CHECK_ARG(lec_conv1x1_bnapply_supported(Cin, Cout, M), "conv1x1_stats: unsupported shape Cin=%d Cout=%d M=%lld", Cin, Cout, (long long)M);
  LEC_CHECK_ARG(partials_bytes >= (int64_t)kC1MaxBlocks * 2 * Cout * (int64_t)sizeof(float), "conv1x1_stats: partials buffer too small");
  hipStream_t st = (hipStream_t)stream;
  if (Cin == 64) return launch_conv1x1<64, 256, true, 3>(x, w, M, Cout, nullptr, partials, n_partials, 0, st);
  return launch_conv1x1<128, 256, true, 3>(x, w, M, Cout, nullptr, partials, n_partials, 0, st);
}

extern "C" int lec_conv1x1_fwd_bnapply(const void* x, const void* w, int64_t M, int Cin, int Cout, const float* scale, const float* shift,
                                       const void* residual, void* y, void* z, uint8_t* relu_mask, lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(x && w && scale && shift && residual && y && z && relu_mask, "conv1x1_fwd_bnapply: null pointer");
  LEC_CHECK_ARG(lec_conv1x1_bnapply_supported(Cin, Cout, M), "conv1x1_fwd_bnapply: unsupported shape Cin=%d Cout=%d M=%lld", Cin, Cout, (long long)M);
  const FoldArgs fa{(const unsigned short*)residual, (const unsigned short*)z, (const unsigned char*)relu_mask, scale, shift};
  hipStream_t st = (hipStream_t)stream;
  if (Cin == 64) return launch_conv1x1<64, 256, false, 2>(x, w, M, Cout, y, nullptr, nullptr, 0, st, fa);
  return launch_conv1x1<128, 256, false, 2>(x, w, M, Cout, y, nullptr, nullptr, 0, st, fa);
}

extern "C" int lec_conv1x1_fwd(const void* x, const void* w, int w_transposed, int64_t M, int Cin, int Cout, void* y, float* partials,
                               int64_t partials_bytes, int* n_partials, lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(x && w && y, "conv1x1_fwd: null pointer");
  LEC_CHECK_ARG(lec_conv1x1_supported(Cin, Cout, M), "conv1x1_fwd: unsupported shape Cin=%d Cout=%d M=%lld", Cin, Cout, (long long)M);
  LEC_CHECK_ARG((partials == nullptr) == (n_partials == nullptr), "conv1x1_fwd: pass partials and n_partials together");
  LEC_CHECK_ARG(!partials || partials_bytes >= (int64_t)kC1MaxBlocks * 2 * Cout * (int64_t)sizeof(float), "conv1x1_fwd: partials buffer too small");
  hipStream_t st = (hipStream_t)stream;
#define LEC_C1(K_, NB_) (partials ? launch_conv1x1<K_, NB_, true>(x, w, M, Cout, y, partials, n_partials, w_transposed, st) \
                                  : launch_conv1x1<K_, NB_, false>(x, w, M, Cout, y, nullptr, nullptr, w_transposed, st))
  if (Cin == 64 && Cout == 256) return LEC_C1(64, 256);
  if (Cin == 64 && Cout == 64) return LEC_C1(64, 64);
  if (Cin == 128) return LEC_C1(128, 256);                          // Cout = 512: two column blocks, X is read twice
  if (Cin == 512) return partials ? launch_conv1x1_bigk<512, 128, true>(x, w, M, Cout, y, partials, n_partials, w_transposed, st)
                                  : launch_conv1x1_bigk<512, 128, false>(x, w, M, Cout, y, nullptr, nullptr, w_transposed, st);
  if (Cin == 256 && Cout == 64) return LEC_C1(256, 64);
  return LEC_C1(256, 128);
#undef LEC_C1
}
