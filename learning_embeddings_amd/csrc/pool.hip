// 3x3 / stride 2 / pad 1 max pooling for NHWC bf16 activations (the ResNet stem: torchvision `self.maxpool`, reached from
// FeatCNN18's backbone, oe_h.py:311,317), forward + backward.
// The framework kernel keeps int64 argmax indices (8 B per pooled element = as many bytes as the whole input) and its
// backward scatters; here the argmax is ONE byte per pooled element (window position 0..8) and backward is a gather:
// each input position looks at the <= 4 windows that cover it -- no atomics, deterministic.
// HBM-bound.  Algorithmic bytes: fwd 2*in + 2*out + 1*out ; bwd 2*out + 1*out + 2*in   (in = 4 * out elements).
#include <hip/hip_bf16.h>
#include "lec_common.h"

namespace lec {

struct alignas(16) pbf16x8 { unsigned short v[8]; };
struct alignas(16) pf32x4 { float v[4]; };
struct alignas(8) pu8x8 { unsigned char v[8]; };

__device__ __forceinline__ float pbf2f(unsigned short u) { return __uint_as_float(((unsigned int)u) << 16); }

// element types: a "vector" is 8 consecutive channels, bf16 (16 bytes) or fp32 (32 bytes: the reference's precision)
struct PBf16 {
  static __device__ __forceinline__ void ld(const void* p, int64_t i, float (&f)[8], int = 0, int = 0) {
    const pbf16x8 v = ((const pbf16x8*)p)[i];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = pbf2f(v.v[j]);
  }
  static __device__ __forceinline__ void st(void* p, int64_t i, const float (&f)[8], int = 0, int = 0) {
    pbf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) { __hip_bfloat16 hb = __float2bfloat16(f[j]); o.v[j] = *reinterpret_cast<unsigned short*>(&hb); }
    ((pbf16x8*)p)[i] = o;
  }
};
// fp32: a thread's 8 channels are two runs of 4 (channels 4 cv .. + 3 and C / 2 + 4 cv .. + 3), as in bn.hip's EF32: each 16-byte access of
// the CV lanes of a pixel is then one contiguous half row instead of every other 16 bytes of the whole row.  i = pixel * CV + cv.
struct PF32 {
  static __device__ __forceinline__ void ld(const void* p, int64_t i, float (&f)[8], int cv, int CV) {
    const char* q0 = (const char*)p + i * 32 - cv * 16;
    const pf32x4 a = *(const pf32x4*)q0, b = *(const pf32x4*)(q0 + CV * 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[j] = a.v[j]; f[4 + j] = b.v[j]; }
  }
  static __device__ __forceinline__ void st(void* p, int64_t i, const float (&f)[8], int cv, int CV) {
    pf32x4 a, b;
#pragma unroll
    for (int j = 0; j < 4; ++j) { a.v[j] = f[j]; b.v[j] = f[4 + j]; }
    char* q0 = (char*)p + i * 32 - cv * 16;
    *(pf32x4*)q0 = a; *(pf32x4*)(q0 + CV * 16) = b;
  }
};

template <typename E>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const void* __restrict__ x, int N, int H, int W, int CV,
                                                          void* __restrict__ y, pu8x8* __restrict__ idx) {
  const int Ho = H / 2, Wo = W / 2;
  const int64_t total = (int64_t)N * Ho * Wo * CV;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV); int64_t r = i / CV;
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
        E::ld(x, (((int64_t)n * H + h) * W + w) * CV + cv, v, cv, CV);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = v[j];
          if (first || f > best[j] || f != f) { best[j] = f; arg[j] = (unsigned char)(kh * 3 + kw); }   // first max wins; NaN propagates
        }
        first = false;
      }
    }
    pu8x8 a;
#pragma unroll
    for (int j = 0; j < 8; ++j) a.v[j] = arg[j];
    E::st(y, i, best, cv, CV);                                                   // exact: the max is one of the inputs
    idx[i] = a;
  }
}

// One thread owns a 2x2 patch of input positions (h = 2i, 2i+1; w = 2j, 2j+1) of one channel vector.  The patch is
// covered by the four windows (ho, wo) in {i, i+1} x {j, j+1}, each loaded ONCE for all four positions (a thread per
// input position re-loaded them 9 times per patch and ran at 2 TB/s, L1-bound):
//   position (2i,   2j  ) <- window (i,   j  ) code 1*3+1
//   position (2i,   2j+1) <- windows (i, j) code 1*3+2, (i, j+1) code 1*3+0
//   position (2i+1, 2j  ) <- windows (i, j) code 2*3+1, (i+1, j) code 0*3+1
//   position (2i+1, 2j+1) <- windows (i, j) 2*3+2, (i, j+1) 2*3+0, (i+1, j) 0*3+2, (i+1, j+1) 0*3+0
template <typename E>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const void* __restrict__ dy, const pu8x8* __restrict__ idx,
                                                          int N, int H, int W, int CV, void* __restrict__ dx) {
  const int Ho = H / 2, Wo = W / 2;
  const int64_t total = (int64_t)N * Ho * Wo * CV;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int cv = (int)(t % CV); int64_t r = t / CV;
    const int j = (int)(r % Wo); r /= Wo;
    const int i = (int)(r % Ho); const int n = (int)(r / Ho);
    float acc[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[q][c] = 0.0f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int ho = i + a, wo = j + b;
        if (ho >= Ho || wo >= Wo) continue;
        const int64_t o = (((int64_t)n * Ho + ho) * Wo + wo) * CV + cv;
        const pu8x8 am = idx[o];
        float g[8];
        E::ld(dy, o, g, cv, CV);
        // window (ho, wo) covers rows 2ho-1..2ho+1: patch row p (h = 2i+p) sits at kh = 2i + p - (2ho - 1) = p + 1 - 2a
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int kh = p + 1 - 2 * a;
          if (kh < 0 || kh > 2) continue;
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int kw = q + 1 - 2 * b;
            if (kw < 0 || kw > 2) continue;
            const int code = kh * 3 + kw;
#pragma unroll
            for (int c = 0; c < 8; ++c) if (am.v[c] == code) acc[p * 2 + q][c] += g[c];
          }
        }
      }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        E::st(dx, (((int64_t)n * H + 2 * i + p) * W + 2 * j + q) * CV + cv, acc[p * 2 + q], cv, CV);
      }
    }
  }
}

}  // namespace lec

template <typename E>
static int maxpool_fwd_impl(const void* x, int N, int H, int W, int C, void* y, uint8_t* argmax, lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(x && y && argmax, "maxpool_fwd: null pointer");
  LEC_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && H % 2 == 0 && W % 2 == 0, "maxpool_fwd: need even H, W and C %% 8 == 0");
  const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 8);
  int64_t nb = (total + 255) / 256; const int nblk = (int)(nb > 8192 ? 8192 : nb);
  hipLaunchKernelGGL((maxpool_fwd_kernel<E>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, N, H, W, C / 8, y, (pu8x8*)argmax);
  LEC_CHECK_LAUNCH("maxpool_fwd_kernel");
  return LEC_OK;
}

template <typename E>
static int maxpool_bwd_impl(const void* dy, const uint8_t* argmax, int N, int H, int W, int C, void* dx, lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(dy && dx && argmax, "maxpool_bwd: null pointer");
  LEC_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && H % 2 == 0 && W % 2 == 0, "maxpool_bwd: need even H, W and C %% 8 == 0");
  const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 8);            // one thread per 2x2 input patch and channel vector
  int64_t nb = (total + 255) / 256; const int nblk = (int)(nb > 16384 ? 16384 : nb);
  hipLaunchKernelGGL((maxpool_bwd_kernel<E>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, dy, (const pu8x8*)argmax, N, H, W, C / 8, dx);
  LEC_CHECK_LAUNCH("maxpool_bwd_kernel");
  return LEC_OK;
}

extern "C" int lec_maxpool3x3s2_fwd(const void* x, int N, int H, int W, int C, void* y, uint8_t* argmax, lec_stream_t stream) {
  return maxpool_fwd_impl<lec::PBf16>(x, N, H, W, C, y, argmax, stream);
}
extern "C" int lec_maxpool3x3s2_fwd_f32(const void* x, int N, int H, int W, int C, void* y, uint8_t* argmax, lec_stream_t stream) {
  return maxpool_fwd_impl<lec::PF32>(x, N, H, W, C, y, argmax, stream);
}
extern "C" int lec_maxpool3x3s2_bwd(const void* dy, const uint8_t* argmax, int N, int H, int W, int C, void* dx, lec_stream_t stream) {
  return maxpool_bwd_impl<lec::PBf16>(dy, argmax, N, H, W, C, dx, stream);
}
extern "C" int lec_maxpool3x3s2_bwd_f32(const void* dy, const uint8_t* argmax, int N, int H, int W, int C, void* dx, lec_stream_t stream) {
  return maxpool_bwd_impl<lec::PF32>(dy, argmax, N, H, W, C, dx, stream);
}
