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
// of the same tensor runs at 6.9 TB/s on this chip and a copy at 5.3: the 48- / 64-byte lane stride of the stores is what is left.
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

// W % 4 == 0, RPB * W * 3 <= 12 KB: the source rows of a block pass go through LDS.  Loads: consecutive lanes fetch consecutive dwords of a source row
// (672 bytes at W = 224); stores: consecutive lanes write consecutive PIXELS (12 or 16 bytes each), so every store instruction of a wave covers
// contiguous memory -- the quad kernel above gives a lane 4 pixels, i.e. three or four 16-byte stores at a 48- / 64-byte lane stride.  The byte -> float
// table and the mirror are as above (three ds_read_u8 + three table reads per pixel).
struct if32x3 { float v[3]; };
template <int CO>
__global__ __launch_bounds__(256) void image_gather_lds_kernel(const uint8_t* __restrict__ store, const int32_t* __restrict__ slots,
                                                               const uint8_t* __restrict__ flip, int n, int H, int W, int RPB,
                                                               int64_t n_slots, float* __restrict__ out) {
  __shared__ float lut[256];
  __shared__ uint32_t srow[3072];                              // RPB source rows, W * 3 bytes each, packed
  __shared__ int rowflip[16];
  const int tid = threadIdx.x;
  lut[tid] = __fdiv_rn((float)tid, 255.0f);
  const int rows = n * H, DW = (W * 3) >> 2;                   // dwords per source row
  for (int row0 = blockIdx.x * RPB; row0 < rows; row0 += gridDim.x * RPB) {
    const int nr = min(RPB, rows - row0);
    __syncthreads();                                            // (the table; the previous pass's readers)
    for (int d = tid; d < nr * DW; d += 256) {
      const int r = d / DW, dw = d - r * DW;
      const int row = row0 + r, i = row / H, h = row - i * H;
      const int64_t slot = slots[i];
      uint32_t v = 0u;                                          // a slot outside the store reads as a black image
      if (slot >= 0 && slot < n_slots) v = ((const uint32_t*)(store + (slot * H + h) * (int64_t)W * 3))[dw];
      srow[d] = v;
      if (dw == 0) rowflip[r] = (flip != nullptr && flip[i] != 0) ? 1 : 0;
    }
    __syncthreads();
    const uint8_t* sb = (const uint8_t*)srow;
    float* obase = out + (int64_t)row0 * W * CO;
    int r = 0, w = tid;
    while (w >= W) { w -= W; ++r; }
    for (int pix = tid; pix < nr * W; pix += 256) {
      const int sw = rowflip[r] ? W - 1 - w : w;
      const uint8_t* p = sb + (r * W + sw) * 3;
      const float c0 = lut[p[0]], c1 = lut[p[1]], c2 = lut[p[2]];
      if (CO == 4) { if32x4 v; v.v[0] = c0; v.v[1] = c1; v.v[2] = c2; v.v[3] = 0.f; *(if32x4*)(obase + (int64_t)pix * 4) = v; }
      else { if32x3 v; v.v[0] = c0; v.v[1] = c1; v.v[2] = c2; *(if32x3*)(obase + (int64_t)pix * 3) = v; }
      w += 256;
      while (w >= W) { w -= W; ++r; }
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
  static const int use_lds = [] { const char* e = getenv("LEC_IMAGE_GATHER_LDS"); return e ? atoi(e) : 1; }();
  if (quad && use_lds && W * 3 <= 12288) {
    int RPB = 12288 / (W * 3); if (RPB > 16) RPB = 16;
    while (RPB > 1 && RPB * W > 1024) --RPB;                   // ~4 pixels per thread per pass
    const int64_t nb2 = ((int64_t)n * H + RPB - 1) / RPB;
    const int nblk2 = (int)(nb2 > (1 << 20) ? (1 << 20) : nb2);
    if (c_out == 4) hipLaunchKernelGGL((image_gather_lds_kernel<4>), dim3(nblk2), dim3(256), 0, st, store, slots, flip, n, H, W, RPB, n_slots, out);
    else            hipLaunchKernelGGL((image_gather_lds_kernel<3>), dim3(nblk2), dim3(256), 0, st, store, slots, flip, n, H, W, RPB, n_slots, out);
  } else if (quad) {
    if (c_out == 4) hipLaunchKernelGGL((image_gather4_kernel<4>), dim3(nblk), dim3(256), 0, st, store, slots, flip, n, H, W, n_slots, out);
    else            hipLaunchKernelGGL((image_gather4_kernel<3>), dim3(nblk), dim3(256), 0, st, store, slots, flip, n, H, W, n_slots, out);
  } else {
    if (c_out == 4) hipLaunchKernelGGL((image_gather1_kernel<4>), dim3(nblk), dim3(256), 0, st, store, slots, flip, n, H, W, n_slots, out);
    else            hipLaunchKernelGGL((image_gather1_kernel<3>), dim3(nblk), dim3(256), 0, st, store, slots, flip, n, H, W, n_slots, out);
  }
  LEC_CHECK_LAUNCH("image_gather_kernel");
  return LEC_OK;
}
