// Shared pieces of the fp32 convolution families (conv_f32.hip: f32-input MFMA; conv_f32x3.hip: the same fp32 products on the bf16
// matrix pipe): exact fast division, raw buffer resources, GEMM geometries, argument checks.
#pragma once
#include <cstdlib>
#include "lec_common.h"

namespace lec {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef int i32x4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4r __attribute__((__vector_size__(4 * sizeof(unsigned int))));

constexpr int kCfBK = 32;                 // K chunk (floats)
constexpr int kCfLdk = kCfBK + 4;         // row stride of a k-contiguous LDS tile: 144 B, conflict-free ds_read_b128
constexpr int kCfKQ = kCfBK / 4;          // 16-byte pieces per k-contiguous row
constexpr int kCfRP = 256 / kCfKQ;        // rows staged per pass of the 256 threads
constexpr int kCfThreads = 256;
constexpr int kCfMaxPart = 512;           // statistics partial rows of the tile-walk kernels (the buffer a caller passes holds this many)
constexpr int kCfMaxRows = 2048;          // = kBnMaxRows: rows the BatchNorm workspace holds; the balanced kernel leaves one row per m-tile
constexpr int kWgBK = 16;                 // K chunk of the weight gradient (pixels): 16 -> 40 KB of LDS per workgroup
constexpr unsigned kOob = 0x80000000u;    // a byte offset no tensor reaches (num_records < 2^31): the load returns zeros, the store is dropped

// exact unsigned division by a launch-invariant divisor (Granlund-Montgomery round-up form: exact for every 32-bit dividend)
struct FastDiv { uint32_t mul, sh1, sh2, d; };
static inline FastDiv make_fastdiv(int d_) {
  FastDiv f; const uint32_t d = (uint32_t)(d_ < 1 ? 1 : d_); f.d = d;
  uint32_t l = 0; while ((1ull << l) < d) ++l;
  f.mul = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << l) - d)) / d + 1);
  f.sh1 = l < 1 ? l : 1; f.sh2 = l > 0 ? l - 1 : 0;
  return f;
}
__device__ __forceinline__ int fdiv(int n, const FastDiv& f) {
  const uint32_t t = __umulhi(f.mul, (uint32_t)n);
  return (int)((t + (((uint32_t)n - t) >> f.sh1)) >> f.sh2);
}

// raw buffer resource over [p, p + bytes): stride 0, hardware range check on the byte offset
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4v bload4(rsrc_t rsrc, unsigned voff) {
  const u32x4r v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, 0, 0);
  return __builtin_bit_cast(f32x4v, v);
}
constexpr int kCfStoreAux = 0;      // cache policy of the convolutions' output stores (gfx942+ aux bits: 1 = sc0, 2 = nt, 16 = sc1; none changed the finalize waits: EXPERIMENTS.md)
__device__ __forceinline__ void bstore1(float v, rsrc_t rsrc, unsigned voff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, (int)voff, 0, kCfStoreAux);
}

// geometry of an "activation-gather" GEMM (forward, or one parity class of a data gradient)
struct ActGeo {
  int Mg;                                  // GEMM rows = pixels of the m-space
  int Hm, Wm;                              // m-space grid per image: m = (n * Hm + mh) * Wm + mw
  int Hs, Ws, Cs, lgCs;                    // gathered source tensor [N, Hs, Ws, Cs]; Cs a power of two
  int sst;                                 // source position of tap (a, b): (mh * sst + oh0 + sg * a, mw * sst + ow0 + sg * b)
  int oh0, ow0, sg;
  int na, nb;                              // taps per dimension of this launch (K = na * nb * Cs)
  int r0, rstep, s0, sstep, S, RS;         // weight tap of (a, b): (r0 + rstep * a) * S + s0 + sstep * b
  int Cd;                                  // GEMM columns = channels of the destination
  int Cin;                                 // the layer's input channels (innermost weight dimension)
  int Hd, Wd, dst_st, dph, dpw;            // destination pixel of m: (n, mh * dst_st + dph, mw * dst_st + dpw) of [N, Hd, Wd, Cd]
  int Kg;                                  // na * nb * Cs
  int zfill;                               // strided destination of a 1x1 layer: the three other pixels of every 2 x 2 block are written as zeros by this launch
  int xcd_per;                             // > 0: m-tiles are dealt to the 8 XCDs in contiguous runs of this many (see launch_act); 0: round-robin
  uint32_t src_bytes, wgt_bytes, dst_bytes;
  FastDiv dWm, dHm, dnb;                   // divisions by Wm, Hm, nb
};

struct ActGeoSet { ActGeo g[4]; };      // the parity classes of a strided data gradient (one launch, blockIdx.z = class)

struct WgGeo {
  int Mpix;                                // N * Ho * Wo
  int Ho, Wo, H, W, Cin, lgCin, Cout;
  int S, RS, stride, pad;
  int Ng;                                  // RS * Cin
  int dCin;                                // channels of the DESTINATION dw [Cout][RS][dCin]: Cin, or 3 for the stem (x carries a zero 4th channel)
  int chunks_per_split, tiles, split;
  uint32_t dy_bytes, x_bytes;
  FastDiv dWo, dHo, dS;
  int HoWo;                                // pixels per image (the shifted-dense form's validity table has one entry each)
  FastDiv dHW;
};

static inline int ilog2_exact(int v) {
  int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1;
}

static inline int conv_check(const char* who, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad) {
  LEC_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && R > 0 && S > 0 && (stride == 1 || stride == 2) && pad >= 0 && pad < R && pad < S,
                "%s: bad geometry N=%d H=%d W=%d Cin=%d Cout=%d R=%d S=%d stride=%d pad=%d", who, N, H, W, Cin, Cout, R, S, stride, pad);
  LEC_CHECK_ARG(Cin % 4 == 0 && ilog2_exact(Cin) >= 2, "%s: Cin must be a power of two >= 4 (pad the stem's 3 channels to 4), got %d", who, Cin);
  LEC_CHECK_ARG(Cout % 4 == 0 && ilog2_exact(Cout) >= 2, "%s: Cout must be a power of two >= 4, got %d", who, Cout);
  const int64_t Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
  LEC_CHECK_ARG(Ho > 0 && Wo > 0, "%s: empty output", who);
  // 32-bit byte offsets with the top bit reserved for "out of range"
  LEC_CHECK_ARG((int64_t)N * H * W * Cin * 4 < (1ll << 31) && (int64_t)N * Ho * Wo * Cout * 4 < (1ll << 31) && (int64_t)Cout * R * S * Cin * 4 < (1ll << 30),
                "%s: a tensor of this layer reaches 2 GiB (N=%d H=%d W=%d Cin=%d Cout=%d): split the batch", who, N, H, W, Cin, Cout);
  return LEC_OK;
}


// csrc/conv_stem_f32.hip: the fp32 stem's weight gradient on its own kernel (lec_conv_f32_wgrad_c3 hands it the sizes lec_conv_f32_stem_supported accepts)
int conv_f32_stem_wgrad_launch(const float* dy, const float* x, int N, int H, int W, float* dw3, void* stream);

}  // namespace lec
