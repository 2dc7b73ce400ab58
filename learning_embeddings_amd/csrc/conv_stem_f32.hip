// The ResNet stem (7x7 / stride 2 / pad 3, 3 -> 64 channels) at the reference's precision (fp32 on v_mfma_f32_16x16x4_f32: exact fp32), forward.
//
// The generic family (conv_f32.hip) runs this layer through its per-piece-tap loader: K = 49 taps x 4 stored channels padded to 7 chunks of 32 = 224,
// a quarter of it the zero 4th channel, 966 us per 256 rows against a matrix-pipe floor of 586 for that K.  Here (the bf16 stem kernel's scheme,
// conv_bf16.hip) a tile is ONE output row: the 7 input rows it needs are staged once into an LDS patch of 16-byte pixels, a 16-byte LDS read hands a lane
// the 4 channels of ITS tap, and only the three real channels are multiplied: K = 14 tap groups x 4 taps x 3 channels = 168 (42 MFMAs per 16 x 16 block).
//   * patch [7 rows][W + 8 positions] of 16 bytes, pixel column c at position c + 3; the lane of output pixel ow and tap s reads position 2 ow + s.  A K group
//     is 4 taps (r, 4 h + kq) of one filter row; the 16 lanes a ds_read_b128 serves per cycle -- pixels {0..3, 12..15} of tap kq = 0 and pixels 4..11 of tap 1 --
//     land on 16 distinct 16-byte columns of the 256-byte bank row (stride 32 bytes, the odd tap filling the gaps);
//   * MFMA m of a group multiplies channel m of those four taps: lane (pixel, kq) supplies v[m] of its 16 bytes, lane (channel, kq) the weight of tap
//     4 h + kq, channel m, which it keeps in a register for the whole launch (42 registers);
//   * wave v owns output channels 16 v .. 16 v + 15 and all the row's pixels; D'[channel][pixel] -> an LDS row image [Wo][64] fp32 (rows padded to 272 bytes)
//     -> 256-byte rows to memory, the BatchNorm statistics (sum y, sum y^2, fp32) of the workgroup's rows as ONE partial row (lec_bn_fwd_prestat's layout);
//   * the pixels of the NEXT tile are requested into registers before this tile's products; a tile waits for them with a counted vmcnt that leaves its
//     predecessor's output stores in flight.  Forward results are deterministic (fixed summation order); they differ from the generic kernel's in the last bit.
#include "conv_geo.h"
#include "tuning.h"

namespace lec {

typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));

struct StemGeoF {
  int N, H, W, Ho, Wo, tiles, npix;         // npix = 7 * (W + 6) pixels a tile requests
  uint32_t x_bytes, w_bytes, y_bytes;
  FastDiv dRow, dHo;                        // / (W + 6), / Ho
};

template <int N> __device__ __forceinline__ void stem_wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else static_assert(N < 0, "add the count");
}

template <bool STATS, int NPB>              // NPB = Wo / 16: 16-pixel blocks of an output row (7 at 224 x 224 images)
__global__ __launch_bounds__(256, 2) void conv_f32_stem_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, StemGeoF g,
                                                               float* __restrict__ part) {
  constexpr int WO = 16 * NPB, PC = 32 * NPB + 8, ROWB = PC * 16, PATCH = 7 * ROWB;
  constexpr int NLD = (7 * (32 * NPB + 6) + 255) / 256;        // pixel loads per thread and tile
  constexpr int OROW = 272;                                    // bytes per row of the output image (256 + 16: the 16 lanes of a store land on 16 columns)
  extern __shared__ __attribute__((aligned(16))) float smemf[];
  char* const patch = (char*)smemf;
  char* const outimg = patch + PATCH;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kq = lane >> 4;
  const rsrc_t rs_x = make_rsrc(x, g.x_bytes), rs_w = make_rsrc(w, g.w_bytes), rs_y = make_rsrc(y, g.y_bytes);
  // weights of group gi = (filter row gi >> 1, half gi & 1): tap s = 4 (gi & 1) + kq of output channel 16 wave + l15, channels 0..2 (w [64][49][4])
  float wr[14][3];
#pragma unroll
  for (int gi = 0; gi < 14; ++gi) {
    const int s = 4 * (gi & 1) + kq;
    const f32x4v v = bload4(rs_w, s < 7 ? (unsigned)(((16 * wave + l15) * 49 + (gi >> 1) * 7 + s) * 16) : kOob);
    wr[gi][0] = v[0]; wr[gi][1] = v[1]; wr[gi][2] = v[2];
  }
  int rel[NLD]; int rr[NLD]; unsigned pofs[NLD];
#pragma unroll
  for (int u = 0; u < NLD; ++u) {
    const int idx = tid + 256 * u;
    const int r = fdiv(idx, g.dRow); const int cpos = idx - r * (g.W + 6);     // position = column + 3
    const int col = cpos - 3;
    const bool ok = idx < g.npix && (unsigned)col < (unsigned)g.W;
    rel[u] = ok ? (r * g.W + col) * 16 : -1;
    rr[u] = r;
    pofs[u] = idx < g.npix ? (unsigned)(r * ROWB + cpos * 16) : 0xffffffffu;
  }
  f32x4v stg[NLD];
  auto request = [&](int t) {
    const int n = fdiv(t, g.dHo); const int oh = t - n * g.Ho;
    const int row0 = 2 * oh - 3;
    const int base = (n * g.H + row0) * g.W * 16;              // (may point before the image: the row test below covers it)
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const bool ok = rel[u] >= 0 && (unsigned)(row0 + rr[u]) < (unsigned)g.H;
      stg[u] = bload4(rs_x, ok ? (unsigned)(base + rel[u]) : kOob);
    }
  };
  const int cc = tid & 15, r0 = tid >> 4;                      // output stores: 4 channels 4 cc .., pixels r0 + 16 i
  float st_s[4], st_q[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { st_s[j] = 0.f; st_q[j] = 0.f; }
  // zero the patch once: the last two positions of every row are never written (never multiplied by a non-zero weight, but must not hold a NaN)
  for (int i = tid; i < PATCH / 16; i += 256) { f32x4v z; z[0] = 0.f; z[1] = 0.f; z[2] = 0.f; z[3] = 0.f; *(f32x4v*)(patch + 16 * i) = z; }

  // Tile walk: workgroup ids go round-robin over the 8 XCDs (one L2 each); XCD x owns the CONTIGUOUS run of output rows [x per, (x + 1) per)
  const int per = (g.tiles + 7) >> 3;
  auto tile_of = [&](int slot) { return (slot & 7) * per + (slot >> 3); };
  const int nslots = 8 * per;
  bool first = true;
  { const int t0 = tile_of(blockIdx.x); if ((int)blockIdx.x < nslots && t0 < g.tiles) request(t0); }
  for (int slot = blockIdx.x; slot < nslots; slot += gridDim.x) {
    const int t = tile_of(slot);
    const int tn = slot + (int)gridDim.x < nslots ? tile_of(slot + gridDim.x) : g.tiles;
    if (t >= g.tiles) continue;                                 // (only the last run can be short: its tail slots have no tile, and neither have their successors)
    if (!first) stem_wait_vmcnt<NPB == 7 ? 7 : NPB == 4 ? 4 : 2>(); else stem_wait_vmcnt<0>();      // the pixels are there; the previous tile's stores stay in flight
    first = false;
    __syncthreads();                                            // the previous tile's patch and row image have been read by every wave
#pragma unroll
    for (int u = 0; u < NLD; ++u) if (pofs[u] != 0xffffffffu) *(f32x4v*)(patch + pofs[u]) = stg[u];
    __syncthreads();
    if (tn < g.tiles) request(tn);
    const char* pb = patch + (2 * l15 + kq) * 16;
    f32x4v acc[NPB];
#pragma unroll
    for (int ib = 0; ib < NPB; ++ib)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[ib][r] = 0.f;
#pragma unroll
    for (int gi = 0; gi < 14; ++gi) {
      f32x4v v[NPB];
#pragma unroll
      for (int ib = 0; ib < NPB; ++ib) v[ib] = *(const f32x4v*)(pb + (gi >> 1) * ROWB + (gi & 1) * 64 + ib * 512);
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int ib = 0; ib < NPB; ++ib) acc[ib] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[gi][m], v[ib][m], acc[ib], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);                        // (keep the fragment reads of one group in flight at a time)
    }
    // D'[channel 4 kq + r][pixel l15] of block ib -> the row image
#pragma unroll
    for (int ib = 0; ib < NPB; ++ib) *(f32x4v*)(outimg + (16 * ib + l15) * OROW + (16 * wave + 4 * kq) * 4) = acc[ib];
    __syncthreads();
    const int n = fdiv(t, g.dHo); const int oh = t - n * g.Ho;
    const unsigned rowbase = (unsigned)((n * g.Ho + oh) * WO) * 256u;
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
      const int px = r0 + 16 * i;
      const f32x4v v = *(const f32x4v*)(outimg + px * OROW + cc * 16);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, v), rs_y, (int)(rowbase + (unsigned)px * 256u + (unsigned)cc * 16u), 0, kCfStoreAux);
      if (STATS) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { st_s[j] += v[j]; st_q[j] += v[j] * v[j]; }
      }
    }
  }
  if (STATS) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = st_s[j], b = st_q[j];
      a += __shfl_xor(a, 16, kWave); b += __shfl_xor(b, 16, kWave);
      a += __shfl_xor(a, 32, kWave); b += __shfl_xor(b, 32, kWave);
      st_s[j] = a; st_q[j] = b;
    }
    __syncthreads();
    float* red = smemf;                                         // [4 waves][2][64]
    if (lane < 16) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { red[(wave * 2 + 0) * 64 + lane * 4 + j] = st_s[j]; red[(wave * 2 + 1) * 64 + lane * 4 + j] = st_q[j]; }
    }
    __syncthreads();
    if (tid < 128) {
      const int sidx = tid >> 6, cidx = tid & 63;
      part[((int64_t)blockIdx.x * 2 + sidx) * 64 + cidx] = ((red[(0 * 2 + sidx) * 64 + cidx] + red[(1 * 2 + sidx) * 64 + cidx]) + red[(2 * 2 + sidx) * 64 + cidx]) + red[(3 * 2 + sidx) * 64 + cidx];
    }
  }
}

// The stem's weight gradient on the same patch, exact fp32: dw[co][(r, s)][ci] = sum over output pixels of dy[p][co] * x[2 oh + r - 3][2 ow + s - 3][ci].
// A tile is one output row, K = its pixels (4 per v_mfma_f32_16x16x4_f32).  Only the 147 real (tap, channel) columns are multiplied: column f = 21 r + 3 s + ci,
// ten blocks of 16 (the last 13 columns of the tenth are idle), each lane keeping the patch offset of ITS column of every block; the operand of pixel p is then
// the float at patch[offset + 32 p].  dY rows are stored 320 bytes apart, so that the four pixels of a K step (one per quarter-wave) read disjoint banks.
// Wave v owns output channels 16 v .. 16 v + 15 and all ten column blocks (40 accumulator registers, the same work for every wave); sums stay in registers over
// the workgroup's rows and leave as float atomics into the layer's [64][7][7][3] slot, like every other weight gradient.
template <int NPB>
__global__ __launch_bounds__(256, 2) void conv_f32_stem_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dw, StemGeoF g) {
  constexpr int WO = 16 * NPB, PC = 32 * NPB + 8, ROWB = PC * 16, PATCH = 7 * ROWB, DROW = 320;
  constexpr int NLD = (7 * (32 * NPB + 6) + 255) / 256, NDY = WO * 16 / 256;
  extern __shared__ __attribute__((aligned(16))) float smemf[];
  char* const patch = (char*)smemf;
  char* const dyt = patch + PATCH;                              // [WO] rows of 64 floats, 320 bytes apart
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kq = lane >> 4;
  const rsrc_t rs_x = make_rsrc(x, g.x_bytes), rs_dy = make_rsrc(dy, g.y_bytes);
  int rel[NLD]; int rr[NLD]; unsigned pofs[NLD];
#pragma unroll
  for (int u = 0; u < NLD; ++u) {
    const int idx = tid + 256 * u;
    const int r = fdiv(idx, g.dRow); const int cpos = idx - r * (g.W + 6);
    const int col = cpos - 3;
    const bool ok = idx < g.npix && (unsigned)col < (unsigned)g.W;
    rel[u] = ok ? (r * g.W + col) * 16 : -1;
    rr[u] = r;
    pofs[u] = idx < g.npix ? (unsigned)(r * ROWB + cpos * 16) : 0xffffffffu;
  }
  // this lane's column of block b: f = 16 b + l15 = 21 r + 3 s + ci -> byte offset of (row r, position s, channel ci) in the patch; idle columns read offset 0
  unsigned boff[10];
#pragma unroll
  for (int b = 0; b < 10; ++b) {
    const int f = 16 * b + l15;
    const int r = f / 21, rem = f - 21 * r; const int s = rem / 3, ci = rem - 3 * s;
    boff[b] = f < 147 ? (unsigned)(r * ROWB + s * 16 + ci * 4) : 0u;
  }
  f32x4v stx[NLD], sty[NDY];
  auto request = [&](int t) {
    const int n = fdiv(t, g.dHo); const int oh = t - n * g.Ho;
    const int row0 = 2 * oh - 3;
    const int base = (n * g.H + row0) * g.W * 16;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const bool ok = rel[u] >= 0 && (unsigned)(row0 + rr[u]) < (unsigned)g.H;
      stx[u] = bload4(rs_x, ok ? (unsigned)(base + rel[u]) : kOob);
    }
    const unsigned dbase = (unsigned)(t * WO) * 256u;
#pragma unroll
    for (int u = 0; u < NDY; ++u) sty[u] = bload4(rs_dy, dbase + (unsigned)(tid + 256 * u) * 16u);
  };
  for (int i = tid; i < PATCH / 16; i += 256) { f32x4v z; z[0] = 0.f; z[1] = 0.f; z[2] = 0.f; z[3] = 0.f; *(f32x4v*)(patch + 16 * i) = z; }
  f32x4v acc[10];
#pragma unroll
  for (int b = 0; b < 10; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[b][r] = 0.f;

  const int per = (g.tiles + 7) >> 3;
  auto tile_of = [&](int slot) { return (slot & 7) * per + (slot >> 3); };
  const int nslots = 8 * per;
  { const int t0 = tile_of(blockIdx.x); if ((int)blockIdx.x < nslots && t0 < g.tiles) request(t0); }
  for (int slot = blockIdx.x; slot < nslots; slot += gridDim.x) {
    const int t = tile_of(slot);
    const int tn = slot + (int)gridDim.x < nslots ? tile_of(slot + gridDim.x) : g.tiles;
    if (t >= g.tiles) continue;
    stem_wait_vmcnt<0>();
    __syncthreads();                                            // the previous tile's operands have been read by every wave
#pragma unroll
    for (int u = 0; u < NLD; ++u) if (pofs[u] != 0xffffffffu) *(f32x4v*)(patch + pofs[u]) = stx[u];
#pragma unroll
    for (int u = 0; u < NDY; ++u) { const int idx = tid + 256 * u; *(f32x4v*)(dyt + (idx >> 4) * DROW + (idx & 15) * 16) = sty[u]; }
    __syncthreads();
    if (tn < g.tiles) request(tn);
    const char* pa = dyt + kq * DROW + (16 * wave + l15) * 4;   // dY[pixel 4 j + kq][channel 16 wave + l15]
    const char* pbx = patch + kq * 32;                           // the pixel's operand: 32 bytes (two positions) per output pixel
#pragma unroll 2
    for (int j = 0; j < WO / 4; ++j) {
      const float a = *(const float*)(pa + j * 4 * DROW);
      float bv[10];
#pragma unroll
      for (int b = 0; b < 10; ++b) bv[b] = *(const float*)(pbx + boff[b] + j * 128);
#pragma unroll
      for (int b = 0; b < 10; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[b], acc[b], 0, 0, 0);
    }
  }
  // D[channel 16 wave + 4 kq + r][column l15 of block b]
#pragma unroll
  for (int b = 0; b < 10; ++b) {
    const int f = 16 * b + l15;
    if (f < 147) {
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(dw + (int64_t)(16 * wave + 4 * kq + r) * 147 + f, acc[b][r]);
    }
  }
}

int conv_f32_stem_wgrad_launch(const float* dy, const float* x, int N, int H, int W, float* dw3, void* stream) {
  StemGeoF g;
  g.N = N; g.H = H; g.W = W; g.Ho = H / 2; g.Wo = W / 2; g.tiles = N * g.Ho; g.npix = 7 * (W + 6);
  g.x_bytes = (uint32_t)((int64_t)N * H * W * 16); g.w_bytes = 0; g.y_bytes = (uint32_t)((int64_t)N * g.Ho * g.Wo * 256);
  g.dRow = make_fastdiv(W + 6); g.dHo = make_fastdiv(g.Ho);
  int gx = 512; if (gx > g.tiles) gx = g.tiles;
  const size_t lds = (size_t)7 * (W + 8) * 16 + (size_t)g.Wo * 320;
  hipStream_t st = (hipStream_t)stream;
  if (W == 224) hipLaunchKernelGGL((conv_f32_stem_wgrad_kernel<7>), dim3(gx), dim3(256), lds, st, dy, x, dw3, g);
  else if (W == 128) hipLaunchKernelGGL((conv_f32_stem_wgrad_kernel<4>), dim3(gx), dim3(256), lds, st, dy, x, dw3, g);
  else hipLaunchKernelGGL((conv_f32_stem_wgrad_kernel<2>), dim3(gx), dim3(256), lds, st, dy, x, dw3, g);
  LEC_CHECK_LAUNCH("conv_f32_stem_wgrad_kernel");
  return LEC_OK;
}

}  // namespace lec

// torchvision's stem convolution (resnet.py conv1) on a 3-channel image stored with 4 channels per pixel: channels 0..2 of x [N, H, W, 4] and w [64][7][7][4]
// enter the product, channel 3 is never multiplied.  Even H, W in {64, 128, 224}, tensors below 2 GiB.
extern "C" int lec_conv_f32_stem_supported(int N, int H, int W) {
  return lec::tuning().cf_stem && N > 0 && H > 0 && H % 2 == 0 && (W == 224 || W == 128 || W == 64)
         && (int64_t)N * (H / 2) * (W / 2) * 256 < (1ll << 31) && (int64_t)N * H * W * 16 < (1ll << 31);
}

extern "C" int lec_conv_f32_stem_fwd(const float* x, const float* w, int N, int H, int W, float* y, float* partials, int64_t partials_bytes, int* n_partials,
                                     lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(x && w && y, "conv_f32_stem_fwd: null pointer");
  LEC_CHECK_ARG(lec_conv_f32_stem_supported(N, H, W), "conv_f32_stem_fwd: %d images of %d x %d are not served by the stem kernel (even height; width 64, 128 or 224; tensors < 2 GiB)", N, H, W);
  LEC_CHECK_ARG(!partials || (n_partials && partials_bytes >= (int64_t)kCfMaxPart * 2 * 64 * (int64_t)sizeof(float)), "conv_f32_stem_fwd: partials buffer too small");
  StemGeoF g;
  g.N = N; g.H = H; g.W = W; g.Ho = H / 2; g.Wo = W / 2; g.tiles = N * g.Ho; g.npix = 7 * (W + 6);
  g.x_bytes = (uint32_t)((int64_t)N * H * W * 16); g.w_bytes = (uint32_t)(64 * 49 * 16); g.y_bytes = (uint32_t)((int64_t)N * g.Ho * g.Wo * 256);
  g.dRow = make_fastdiv(W + 6); g.dHo = make_fastdiv(g.Ho);
  int gx = kCfMaxPart; if (gx > g.tiles) gx = g.tiles;          // two workgroups per CU; one partial row each
  const size_t lds = (size_t)7 * (W + 8) * 16 + (size_t)g.Wo * 272;
  hipStream_t st = (hipStream_t)stream;
#define LEC_STEMF_LAUNCH(NPB_) do { if (partials) hipLaunchKernelGGL((conv_f32_stem_kernel<true, NPB_>), dim3(gx), dim3(256), lds, st, x, w, y, g, partials); \
                                    else hipLaunchKernelGGL((conv_f32_stem_kernel<false, NPB_>), dim3(gx), dim3(256), lds, st, x, w, y, g, partials); } while (0)
  if (W == 224) LEC_STEMF_LAUNCH(7); else if (W == 128) LEC_STEMF_LAUNCH(4); else LEC_STEMF_LAUNCH(2);
#undef LEC_STEMF_LAUNCH
  if (n_partials) *n_partials = gx;
  LEC_CHECK_LAUNCH("conv_f32_stem_kernel");
  return LEC_OK;
}
