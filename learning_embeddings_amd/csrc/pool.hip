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
struct alignas(8) pu8x8 { unsigned char v[8]; };

__device__ __forceinline__ float pbf2f(unsigned short u) { return __uint_as_float(((unsigned int)u) << 16); }

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const pbf16x8* __restrict__ x, int N, int H, int W, int CV,
                                                          pbf16x8* __restrict__ y, pu8x8* __restrict__ idx) {
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
        const pbf16x8 v = x[(((int64_t)n * H + h) * W + w) * CV + cv];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = pbf2f(v.v[j]);
          if (first || f > best[j] || f != f) { best[j] = f; arg[j] = (unsigned char)(kh * 3 + kw); }   // first max wins; NaN propagates
        }
        first = false;
      }
    }
    pbf16x8 o; pu8x8 a;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      o.v[j] = (unsigned short)(__float_as_uint(best[j]) >> 16);          // exact: the max is one of the bf16 inputs
      a.v[j] = arg[j];
    }
    y[i] = o; idx[i] = a;
  }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const pbf16x8* __restrict__ dy, const pu8x8* __restrict__ idx,
                                                          int N, int H, int W, int CV, pbf16x8* __restrict__ dx) {
  const int Ho = H / 2, Wo = W / 2;
  const int64_t total = (int64_t)N * H * W * CV;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV); int64_t r = i / CV;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H); const int n = (int)(r / H);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
    // windows ho with 2ho-1 <= h <= 2ho+1: ho = h/2 for even h; (h-1)/2 and (h+1)/2 for odd h
    const int ho0 = (h & 1) ? (h - 1) / 2 : h / 2, nho = (h & 1) ? 2 : 1;
    const int wo0 = (w & 1) ? (w - 1) / 2 : w / 2, nwo = (w & 1) ? 2 : 1;
    for (int a = 0; a < nho; ++a) {
      const int ho = ho0 + a;
      if (ho >= Ho) continue;
      const int kh = h - (2 * ho - 1);
      for (int b = 0; b < nwo; ++b) {
        const int wo = wo0 + b;
        if (wo >= Wo) continue;
        const int code = kh * 3 + (w - (2 * wo - 1));
        const int64_t o = (((int64_t)n * Ho + ho) * Wo + wo) * CV + cv;
        const pu8x8 am = idx[o];
        const pbf16x8 g = dy[o];
#pragma unroll
        for (int j = 0; j < 8; ++j) if (am.v[j] == code) acc[j] += pbf2f(g.v[j]);
      }
    }
    pbf16x8 out;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      __hip_bfloat16 hb = __float2bfloat16(acc[j]);
      out.v[j] = *reinterpret_cast<unsigned short*>(&hb);
    }
    dx[i] = out;
  }
}

}  // namespace lec

extern "C" int lec_maxpool3x3s2_fwd(const void* x, int N, int H, int W, int C, void* y, uint8_t* argmax, lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(x && y && argmax, "maxpool_fwd: null pointer");
  LEC_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && H % 2 == 0 && W % 2 == 0, "maxpool_fwd: need even H, W and C %% 8 == 0");
  const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 8);
  int64_t nb = (total + 255) / 256; const int nblk = (int)(nb > 8192 ? 8192 : nb);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const pbf16x8*)x, N, H, W, C / 8, (pbf16x8*)y, (pu8x8*)argmax);
  LEC_CHECK_LAUNCH("maxpool_fwd_kernel");
  return LEC_OK;
}

extern "C" int lec_maxpool3x3s2_bwd(const void* dy, const uint8_t* argmax, int N, int H, int W, int C, void* dx, lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(dy && dx && argmax, "maxpool_bwd: null pointer");
  LEC_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && H % 2 == 0 && W % 2 == 0, "maxpool_bwd: need even H, W and C %% 8 == 0");
  const int64_t total = (int64_t)N * H * W * (C / 8);
  int64_t nb = (total + 255) / 256; const int nblk = (int)(nb > 16384 ? 16384 : nb);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const pbf16x8*)dy, (const pu8x8*)argmax, N, H, W, C / 8, (pbf16x8*)dx);
  LEC_CHECK_LAUNCH("maxpool_bwd_kernel");
  return LEC_OK;
}
