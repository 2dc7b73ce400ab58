// HBM-resident image store: the input side of the training step.
// The reference turns every image of a step into a float tensor on the HOST, one image at a time: cv2.imread -> ToPILImage ->
// Resize((224, 224)) -> [RandomHorizontalFlip] -> ToTensor (uint8 HWC -> float CHW / 255), positives in DataLoader workers
// (oe_h.py:700-712, 1463-1471) and every image drawn as a negative synchronously in the training thread (oe_h.py:668-677, 980-983,
// 1003-1007), then torch.stack + .to(device).  What the file determines is the RESIZED uint8 image (150 528 bytes at 224 x 224);
// everything after it is arithmetic.  Here that uint8 image lives in HBM ([slots, H, W, 3], all of ETHEC = 7 GB of 288) and ONE launch
// per step does the rest for all rows of the CNN batch: gather by slot, optional mirror along W, uint8 -> fp32 / 255 (ToTensor's
// `img.to(float32).div(255)`: an IEEE division, taken from a 256-entry table built with __fdiv_rn), NHWC with the stem's zero 4th
// channel already in place.
// HBM-bound.  Algorithmic bytes per pixel: 3 read + 4 * c_out written (c_out = 4: 19 B; a 512-row step = 488 MB).
#include "lec_common.h"

namespace lec {

struct alignas(16) if32x4 { float v[4]; };

// W % 4 == 0: one thread = 4 consecutive output pixels of a row = 12 source bytes (three aligned dwords, mirrored or not).
// Measured (512 rows of 224 x 224, one MI355X): 111 us = 3.4 TB/s of algorithmic bytes at c_out = 3.  Tried in round 4 without gain: rows walked with
// 32-bit indices (no 64-bit divisions per item): 118 us; one pixel per thread with ONE contiguous 12- / 16-byte store per lane: 162 / 174 us (the byte loads cost more
// than the coalesced stores save).  0.1 ms of a 131 ms step: left here.
template <int CO>
__global__ __launch_bounds__(256) void image_gather4_kernel(const uint8_t* __restrict__ store, const int32_t* __restrict__ slots,
                                                            const uint8_t* __restrict__ flip, int n, int H, int W,
                                                            int64_t n_slots, float* __restrict__ out) {
  __shared__ float lut[256];
  lut[threadIdx.x] = __fdiv_rn((float)threadIdx.x, 255.0f);
  __syncthreads();
  const int Q = W >> 2;
  const int64_t total = (int64_t)n * H * Q;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int q = (int)(t % Q); int64_t r = t / Q;
    const int h = (int)(r % H); const int i = (int)(r / H);
    const int64_t slot = slots[i];
    const bool fl = flip != nullptr && flip[i] != 0;
    const int src = fl ? W - 4 - 4 * q : 4 * q;
    uint32_t w0 = 0u, w1 = 0u, w2 = 0u;                      // a slot outside the store reads as a black image
    if (slot >= 0 && slot < n_slots) {
      const uint32_t* p = (const uint32_t*)(store + ((slot * H + h) * (int64_t)W + src) * 3);
      w0 = p[0]; w1 = p[1]; w2 = p[2];
    }
    uint8_t b[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) { b[k] = (w0 >> (8 * k)) & 255u; b[4 + k] = (w1 >> (8 * k)) & 255u; b[8 + k] = (w2 >> (8 * k)) & 255u; }
    float f[4 * CO];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int c = 0; c < 3; ++c) f[j * CO + c] = lut[fl ? b[3 * (3 - j) + c] : b[3 * j + c]];   // constant indices + a select: registers
      if (CO == 4) f[j * CO + 3] = 0.0f;
    }
    if32x4* o = (if32x4*)(out + (((int64_t)i * H + h) * W + 4 * q) * CO);
#pragma unroll
    for (int k = 0; k < CO; ++k) {
      if32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v.v[e] = f[4 * k + e];
      o[k] = v;
    }
  }
}

// any W: one thread per output pixel
template <int CO>
__global__ __launch_bounds__(256) void image_gather1_kernel(const uint8_t* __restrict__ store, const int32_t* __restrict__ slots,
                                                            const uint8_t* __restrict__ flip, int n, int H, int W,
                                                            int64_t n_slots, float* __restrict__ out) {
  __shared__ float lut[256];
  lut[threadIdx.x] = __fdiv_rn((float)threadIdx.x, 255.0f);
  __syncthreads();
  const int64_t total = (int64_t)n * H * W;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int w = (int)(t % W); int64_t r = t / W;
    const int h = (int)(r % H); const int i = (int)(r / H);
    const int64_t slot = slots[i];
    const bool fl = flip != nullptr && flip[i] != 0;
    float* o = out + t * CO;
    if (slot >= 0 && slot < n_slots) {
      const uint8_t* p = store + ((slot * H + h) * (int64_t)W + (fl ? W - 1 - w : w)) * 3;
      o[0] = lut[p[0]]; o[1] = lut[p[1]]; o[2] = lut[p[2]];
    } else {
      o[0] = o[1] = o[2] = 0.0f;
    }
    if (CO == 4) o[3] = 0.0f;
  }
}

}  // namespace lec

extern "C" int lec_image_gather_u8(const uint8_t* store, int64_t n_slots, const int32_t* slots, const uint8_t* flip, int n, int H, int W,
                                   int c_out, float* out, lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(store && slots && out, "image_gather_u8: null pointer");
  LEC_CHECK_ARG(n > 0 && H > 0 && W > 0 && n_slots > 0, "image_gather_u8: need n, H, W, n_slots > 0");
  LEC_CHECK_ARG(c_out == 3 || c_out == 4, "image_gather_u8: c_out must be 3 (NHWC rgb) or 4 (zero 4th channel)");
  LEC_CHECK_ARG(((uintptr_t)store & 3) == 0 && ((uintptr_t)out & 15) == 0, "image_gather_u8: store must be 4-byte and out 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const bool quad = (W % 4) == 0;
  const int64_t total = (int64_t)n * H * (quad ? W / 4 : W);
  int64_t nb = (total + 255) / 256; const int nblk = (int)(nb > 16384 ? 16384 : nb);
  if (quad) {
    if (c_out == 4) hipLaunchKernelGGL((image_gather4_kernel<4>), dim3(nblk), dim3(256), 0, st, store, slots, flip, n, H, W, n_slots, out);
    else            hipLaunchKernelGGL((image_gather4_kernel<3>), dim3(nblk), dim3(256), 0, st, store, slots, flip, n, H, W, n_slots, out);
  } else {
    if (c_out == 4) hipLaunchKernelGGL((image_gather1_kernel<4>), dim3(nblk), dim3(256), 0, st, store, slots, flip, n, H, W, n_slots, out);
    else            hipLaunchKernelGGL((image_gather1_kernel<3>), dim3(nblk), dim3(256), 0, st, store, slots, flip, n, H, W, n_slots, out);
  }
  LEC_CHECK_LAUNCH("image_gather_kernel");
  return LEC_OK;
}
