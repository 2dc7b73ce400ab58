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

// W % 4 == 0: one thread = 4 consecutive output pixels of a row = 12 source bytes (three aligned dwords, mirrored or not).  A block walks image rows:
// its 256 threads are RPB = 256 / Q rows of Q = W / 4 quads (224 wide: 4 rows of 56), 32-bit indices, one division per thread per row.
// History (512 rows of 224 x 224, c_out = 3, one MI355X): the first version indexed byte / float ARRAYS under `fl ? x[a] : x[b]`; the compiler selected the
// INDEX and then indexed the register array dynamically -- a 130-deep compare / select chain, 883 instructions per 4 pixels, bound by the vector ALU:
// 111 us = 3.4 TB/s.  One pixel per thread with one contiguous store per lane: 162 us (byte loads).  Named scalars + bit-select + this row walk: 115 us (so it
// was not the ALU alone); + ONE 12-byte load per lane and a grid of exactly one pass per block: 97.5 us = 3.95 TB/s (c_out = 4: 149.5 us).  A plain fill
// of the same tensor runs at 6.9 TB/s on this chip and a copy at 5.3.  Two more forms measured slower and were dropped: source rows staged through LDS with
// one contiguous 12- / 16-byte pixel store per lane (128 / 146 us: the stores' lane stride is NOT what limits the quad form), and four rows per thread with their
// loads issued together (111 / 161 us: nor is it load latency).  What binds the remaining factor of two is not identified.
template <int CO>
__global__ __launch_bounds__(256) void image_gather4_kernel(const uint8_t* __restrict__ store, const int32_t* __restrict__ slots,
                                                            const uint8_t* __restrict__ flip, int n, int H, int W,
                                                            int64_t n_slots, float* __restrict__ out) {
  __shared__ float lut[256];
  lut[threadIdx.x] = __fdiv_rn((float)threadIdx.x, 255.0f);
  __syncthreads();
  const int Q = W >> 2;
  const int RPB = Q >= 256 ? 1 : 256 / Q;                     // rows per block pass
  const int rib = (int)threadIdx.x / Q, q0 = (int)threadIdx.x - rib * Q;
  const int rows = n * H;
  if (rib >= RPB) return;                                     // (after the barrier: the idle tail of the block)
  for (int row = blockIdx.x * RPB + rib; row < rows; row += gridDim.x * RPB) {
    const int i = row / H, h = row - i * H;
    const int64_t slot = slots[i];
    const bool fl = flip != nullptr && flip[i] != 0;
    const bool ok = slot >= 0 && slot < n_slots;              // a slot outside the store reads as a black image
    const uint8_t* srow = store + (ok ? (slot * H + h) * (int64_t)W * 3 : 0);
    float* orow = out + (int64_t)row * W * CO;
    const unsigned m = fl ? 0xffffffffu : 0u;
    for (int q = q0; q < Q; q += 256) {                       // (Q > 256: images wider than 1024 pixels)
      uint3 w = make_uint3(0u, 0u, 0u);
      if (ok) w = *(const uint3*)(srow + (fl ? W - 4 - 4 * q : 4 * q) * 3);        // one 12-byte load per lane: a wave reads 768 contiguous bytes
      const uint32_t w0 = w.x, w1 = w.y, w2 = w.z;
      const float a0 = lut[w0 & 255u], a1 = lut[(w0 >> 8) & 255u], a2 = lut[(w0 >> 16) & 255u];       // source pixel 0
      const float b0 = lut[w0 >> 24], b1 = lut[w1 & 255u], b2 = lut[(w1 >> 8) & 255u];                // source pixel 1
      const float c0 = lut[(w1 >> 16) & 255u], c1 = lut[w1 >> 24], c2 = lut[w2 & 255u];               // source pixel 2
      const float d0 = lut[(w2 >> 8) & 255u], d1 = lut[(w2 >> 16) & 255u], d2 = lut[w2 >> 24];        // source pixel 3
      // mirrored: output pixel j is source pixel 3 - j (one v_bfi per value)
#define LEC_SEL(x, y) __uint_as_float((__float_as_uint(x) & m) | (__float_as_uint(y) & ~m))
      float f[4 * CO];
      f[0 * CO + 0] = LEC_SEL(d0, a0); f[0 * CO + 1] = LEC_SEL(d1, a1); f[0 * CO + 2] = LEC_SEL(d2, a2);
      f[1 * CO + 0] = LEC_SEL(c0, b0); f[1 * CO + 1] = LEC_SEL(c1, b1); f[1 * CO + 2] = LEC_SEL(c2, b2);
      f[2 * CO + 0] = LEC_SEL(b0, c0); f[2 * CO + 1] = LEC_SEL(b1, c1); f[2 * CO + 2] = LEC_SEL(b2, c2);
      f[3 * CO + 0] = LEC_SEL(a0, d0); f[3 * CO + 1] = LEC_SEL(a1, d1); f[3 * CO + 2] = LEC_SEL(a2, d2);
#undef LEC_SEL
      if (CO == 4) { f[3] = 0.0f; f[7] = 0.0f; f[11] = 0.0f; f[15] = 0.0f; }
      if32x4* o = (if32x4*)(orow + 4 * q * CO);
#pragma unroll
      for (int k = 0; k < CO; ++k) {
        if32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v.v[e] = f[4 * k + e];
        o[k] = v;
      }
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
  int64_t nb;
  if (quad) { const int Q = W / 4, RPB = Q >= 256 ? 1 : 256 / Q; nb = ((int64_t)n * H + RPB - 1) / RPB; }   // one pass of a block = RPB image rows
  else nb = ((int64_t)n * H * W + 255) / 256;
  LEC_CHECK_ARG((int64_t)n * H < (1ll << 31), "image_gather_u8: n * H must stay below 2^31");
  const int nblk = (int)(nb > (1 << 20) ? (1 << 20) : nb);          // one pass per block whenever possible: no ragged last round
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
