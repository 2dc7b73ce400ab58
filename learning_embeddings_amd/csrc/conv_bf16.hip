// Convolutions of the ResNet backbone on bf16 activations (BASELINE config 5's 16-bit conv stack): ONE implicit-GEMM family for every layer
// shape, stride and direction, on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
//
// The reference reaches these layers as torchvision's ResNet inside FeatCNN (oe_h.py:331-351 -> resnet50, finetuner.py:122); at 16 bits the
// step used to hand every 3x3 layer above 64 channels, every strided layer, the stem and the 1x1 layers with >= 1024 input channels to
// MIOpen / CK / hipBLASLt (a third of config 5's kernel time).  This file is the bf16 sibling of conv_f32.hip: the same three GEMMs
//
//     forward        Y[m, co]  = sum_{tap, ci} X[pixel(m) + tap, ci] * W[co, tap, ci]           M = N*Ho*Wo, N = Cout, K = R*S*Cin
//     data gradient  dX[m, ci] = sum_{tap, co} dY[src(m, tap), co]  * Wt[ci, tap, co]           M = N*H*W,   N = Cin,  K = taps*Cout
//     weight grad.   dW[co, (tap, ci)] += sum_m dY[m, co] * X[pixel(m) + tap, ci]                M = Cout, N = R*S*Cin, K = N*Ho*Wo
//
// the same geometry structs (conv_geo.h: ActGeo / WgGeo), the same raw-buffer addressing (a 32-bit byte offset per 16-byte piece, the
// hardware range check returns the zeros of padding, row tails and K tails), the same scalar tap decode.  What differs, and why:
//
//   * At the bf16 rate (16x the f32 MFMA's) NO ResNet layer is bound by the matrix pipe at a 128 x 128 tile: a tile consumes 32 KB of operands
//     per 512 matrix cycles.  These kernels are bound by bytes -- HBM for the activations, L2 for the re-read operand -- so the design
//     goals are one HBM read of the input and one write of the output per layer, 16-byte accesses everywhere, and BatchNorm work in the epilogues
//     (the statistics of the forward's output; pass 1 of the BatchNorm backward in the data gradient's) so that those passes never re-read
//     what the convolution still has in registers.
//   * A K chunk is 64 bf16 = 128 bytes per row: the byte geometry of conv_f32's 32-float chunk.  Both operands of the forward are
//     k-contiguous; the data gradient gets a k-contiguous B operand too by reading TRANSPOSED weights Wt[ci][tap][co]
//     (lec_conv_bf16_wt_transpose, tens of KB to 4.7 MB per layer, once per optimizer step) -- so forward and data gradient are ONE loader:
//     16-byte buffer loads -> registers (a chunk ahead) -> LDS rows of 144 bytes -> ds_read_b128 = exactly one MFMA fragment (lane half h
//     holds k = 8h .. 8h + 7).
//   * The product is formed as D'[channel][pixel] (A = weights, B = activations): a lane then holds FOUR consecutive channels of one pixel per
//     register group, packs them to 8 bytes and the 64 x 64 wave tile goes through a per-wave LDS image once, coming back as 16-byte row
//     segments -- coalesced 128-byte stores, and the row-segment form is also the BatchNorm kernels' vector (8 consecutive channels), so the
//     statistics, the ReLU bitmask, the residual gradient and the BatchNorm input are addressed exactly as bn.hip addresses them.
//   * The weight gradient's operands are both k-slow in memory (k = pixel).  They are staged AS THEY LIE (16-byte pieces, ds_write_b128) and
//     read back through gfx950's transposing LDS read ds_read_b64_tr_b16 (two per fragment); row strides of 2 BM + 64 bytes make both the
//     stores and the transposed reads conflict-free.  fp32 accumulation, float atomics into the arena's gradient slot (K split over ~1 024
//     work items, dealt to the XCDs in contiguous runs like conv_f32's).
//
// Roofline: HBM.  Algorithmic bytes per launch: forward / data gradient 2 (M Cs' + M Cd) with Cs' the distinct source elements per row
// (Cs at stride 1), weight gradient 2 M (Cout + Cin'), weights and partials are noise.
#include "conv_geo.h"
#include "tuning.h"

namespace lec {

typedef short bfrag __attribute__((ext_vector_type(8)));          // one MFMA A/B fragment: 8 bf16 = 4 VGPRs
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4q __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2q __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int kBfBK = 64;                 // K chunk (bf16 elements): 128 bytes per row
constexpr int kBfLdk = kBfBK + 8;         // LDS row stride (elements): 144 B, conflict-free ds_read_b128
constexpr int kBfKQ = kBfBK / 8;          // 16-byte pieces per row
constexpr int kBfThreads = 256;
constexpr int kBfWBK = 32;                // K chunk of the weight gradient (pixels)

__device__ __forceinline__ u32x4q bload16(rsrc_t rsrc, unsigned voff) {
  return __builtin_bit_cast(u32x4q, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, 0, 0));
}
__device__ __forceinline__ void bstore16(u32x4q v, rsrc_t rsrc, unsigned voff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4r, v), rsrc, (int)voff, 0, 0);
}
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
// two fp32 -> packed bf16 pair, round to nearest even, NaN stays NaN (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 v; v[0] = lo; v[1] = hi;
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
}
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// FOLD (epilogue of a stride-1 data gradient): instead of dx the kernel writes g = mask * (dx + dres) rounded to bf16 -- pass 1 of the backward of
// the BatchNorm IN FRONT of this layer (whose output this layer consumed; dres = the gradient of that output's other consumer, or null) -- and
// leaves the per-channel partials (sum g, sum g * xhat), xhat from that BatchNorm's input xbn, in lec_bn_bwd's workspace layout.  The mask is
// bn.hip's EBf16 bitmask: byte row * (C / 8) + c / 8, bit c % 8.
struct BfFuse {
  const unsigned short* dres; const unsigned short* xbn; const unsigned char* mask; const float* mean; const float* invstd;
  uint32_t mask_bytes;
};

// ---------------------------------------------------------------------------------------------------------------
// forward / data gradient.  Workgroup tile BM pixels x BN channels, 4 waves as WM x WN, a wave owns 32 TM pixels x 64 channels
// (acc[jt][it]: channel tile jt on the MFMA rows, pixel tile it on the lanes).  Instances: 128 x 128 (WM 2, WN 2, TM 2) and 128 x 64 for
// 64-channel destinations (WM 4, WN 1, TM 1).  TAPV: source channels narrower than a chunk (the stem's 8): one tap per 16-byte piece.
template <int WM, int WN, int TM, bool STATS, bool TAPV, bool FOLD>
__device__ __forceinline__ void bf16_act_body(const unsigned short* __restrict__ src, const unsigned short* __restrict__ wgt,
                                              unsigned short* __restrict__ dst, const ActGeo& g, float* __restrict__ part, const BfFuse& fz,
                                              const int bx, const int gdx, const int by) {
  static_assert((WM * WN == 4 || WM * WN == 8) && !(FOLD && STATS) && !(FOLD && TAPV), "four or eight waves; one statistics epilogue at a time");
  constexpr int TN = 2;
  constexpr int NT = 64 * WM * WN;                            // threads of the workgroup
  constexpr int RP = NT / kBfKQ;                              // rows staged per pass
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  constexpr int NA = BM * kBfKQ / NT, NB = BN * kBfKQ / NT;
  static_assert(NA >= 1 && NB >= 1, "tile too small for the workgroup");
  constexpr int SA = BM * kBfLdk, SB = BN * kBfLdk;           // elements
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * 32 * TM, wn0 = (wave % WN) * 32 * TN;
  const int n0 = by * BN;
  const int nchunks = (g.Kg + kBfBK - 1) / kBfBK;
  const int mtiles = (g.Mg + BM - 1) / BM;
  const int kqA = tid & (kBfKQ - 1), rowA = tid / kBfKQ;
  const rsrc_t rs_src = make_rsrc(src, g.src_bytes), rs_wgt = make_rsrc(wgt, g.wgt_bytes), rs_dst = make_rsrc(dst, g.dst_bytes);
  const int rsc = g.RS * g.Cs;                                 // weight row (one destination channel): RS taps x Cs source channels, k-contiguous
  unsigned wB[NB];
#pragma unroll
  for (int u = 0; u < NB; ++u) { const int co = n0 + rowA + RP * u; wB[u] = co < g.Cd ? (unsigned)(co * rsc + 8 * kqA) * 2u : kOob; }
  const unsigned ldsA = (unsigned)((rowA * kBfLdk + 8 * kqA) * 2);
  const unsigned ldsB = (unsigned)((SA + rowA * kBfLdk + 8 * kqA) * 2);
  const bool dense_dst = g.dst_st == 1;
  const int ntaps = g.na * g.nb;
  const int l31 = lane & 31, h = lane >> 5;
  const int cc = lane & 7, r0 = lane >> 3;                     // epilogue: this lane's channel octet of the wave's 64 and its first row
  float st_s[8], st_q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { st_s[j] = 0.f; st_q[j] = 0.f; }
  float f_mu[8], f_is[8];
  if (FOLD) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = n0 + wn0 + cc * 8 + j; const bool okc = c < g.Cd;
      f_mu[j] = okc ? fz.mean[c] : 0.f; f_is[j] = okc ? fz.invstd[c] : 0.f;
    }
  }

  // Per-tile loader state, set up by tile_setup(mt): source byte offset of each staged row at tap (0, 0) and a bit per tap (the row exists and the
  // tap's source pixel lies inside the image).
  int rowoff[NA]; unsigned tapmask[NA]; int hb[TAPV ? NA : 1], wb[TAPV ? NA : 1], pixn[TAPV ? NA : 1];
  unsigned cur[NA];
  int cur_tap = -1;
  auto tile_setup = [&](int mt_) {
    const int m0_ = mt_ * BM;
    cur_tap = -1;
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int m = m0_ + rowA + RP * u;
      const bool live = m < g.Mg;
      const int mm = live ? m : 0;
      const int t2 = fdiv(mm, g.dWm); const int mw = mm - t2 * g.Wm; const int n = fdiv(t2, g.dHm); const int mh = t2 - n * g.Hm;
      const int hb_ = mh * g.sst + g.oh0, wb_ = mw * g.sst + g.ow0;
      if constexpr (TAPV) { hb[u] = hb_; wb[u] = wb_; pixn[u] = live ? n * g.Hs * g.Ws : -1; }
      rowoff[u] = (((n * g.Hs + hb_) * g.Ws + wb_) << g.lgCs) * 2 + 16 * kqA;
      unsigned msk = 0;
      if (!TAPV) {
        for (int t = 0; t < ntaps; ++t) {
          const int ta = fdiv(t, g.dnb), tb = t - ta * g.nb;
          const int hs = hb_ + g.sg * ta, ws = wb_ + g.sg * tb;
          msk |= ((unsigned)hs < (unsigned)g.Hs && (unsigned)ws < (unsigned)g.Ws ? 1u : 0u) << t;
        }
      }
      tapmask[u] = live ? msk : 0u;
    }
  };
  // Operands travel global -> registers (one chunk ahead) -> LDS (two buffers, one barrier per chunk).
  // MEASURED and removed (round 6, same box, 512 rows, every ResNet-50 layer; profiles/EXPERIMENTS.md "bf16 loader pipeline"): two chunks in flight in two
  // register sets (3x3 layers -1 .. -5 %, every other data gradient +20 .. +60 %: the second set spills), the next tile's first chunk requested under
  // this tile's epilogue (+-1 %), the requests pinned above the MFMA block (+-2 %).  None moves the compute-bound layers off ~150 us: at a
  // 128 x 128 tile the matrix pipe, the LDS array (64 KB read + 32 KB written per chunk) and the vector-memory path (32 KB per chunk) each need
  // ~512 cycles per chunk and overlap imperfectly -- a third of the bf16 matrix peak is this tile's ceiling, memory latency is not what binds.
  u32x4q ra0[NA], rb0[NB];
  auto load_chunk = [&](int ch, auto& ra, auto& rb) {
    const int k0 = ch * kBfBK;
    if constexpr (!TAPV) {
      const int tap = k0 >> g.lgCs, c0 = k0 & (g.Cs - 1);
      const int ta = fdiv(tap, g.dnb), tb = tap - ta * g.nb;
      if (tap != cur_tap) {
        cur_tap = tap;
        const int toff = (((g.sg * ta) * g.Ws + g.sg * tb) << g.lgCs) * 2;
        const unsigned tapbit = tap < 32 ? 1u << tap : 0u;
#pragma unroll
        for (int u = 0; u < NA; ++u) cur[u] = (tapmask[u] & tapbit) ? (unsigned)(rowoff[u] + toff) : kOob;
      }
      const int tw = (g.r0 + g.rstep * ta) * g.S + g.s0 + g.sstep * tb;
      const unsigned wsc = (unsigned)(tw * g.Cs + c0) * 2u;
      const unsigned c0b = (unsigned)c0 * 2u;
#pragma unroll
      for (int u = 0; u < NA; ++u) ra[u] = bload16(rs_src, cur[u] + c0b);
#pragma unroll
      for (int u = 0; u < NB; ++u) rb[u] = bload16(rs_wgt, wB[u] + wsc);
      return;
    }
    // one tap / channel position per 16-byte piece
    const int kA = k0 + 8 * kqA;
    const int tapA = kA >> g.lgCs, cA = kA & (g.Cs - 1);
    const int ta = fdiv(tapA, g.dnb), tb = tapA - ta * g.nb;
    const int dh = g.sg * ta, dw = g.sg * tb;
    const bool tap_ok = tapA < ntaps;
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int hs = hb[u] + dh, ws = wb[u] + dw;
      const bool ok = tap_ok && pixn[u] >= 0 && (unsigned)hs < (unsigned)g.Hs && (unsigned)ws < (unsigned)g.Ws;
      ra[u] = bload16(rs_src, ok ? (unsigned)(((pixn[u] + hs * g.Ws + ws) << g.lgCs) + cA) * 2u : kOob);
    }
    const int tw = (g.r0 + g.rstep * ta) * g.S + g.s0 + g.sstep * tb;
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int co = n0 + rowA + RP * u;
      rb[u] = bload16(rs_wgt, (tap_ok && co < g.Cd) ? (unsigned)((co * g.RS + tw) * g.Cs + cA) * 2u : kOob);
    }
  };
  auto store_chunk = [&](int buf, const auto& ra, const auto& rb) {
    char* base = (char*)smem + buf * (SA + SB) * 2;
#pragma unroll
    for (int u = 0; u < NA; ++u) *(u32x4q*)(base + ldsA + u * RP * kBfLdk * 2) = ra[u];
#pragma unroll
    for (int u = 0; u < NB; ++u) *(u32x4q*)(base + ldsB + u * RP * kBfLdk * 2) = rb[u];
  };
  for (int mt = bx; mt < mtiles; mt += gdx) {
    const int m0 = mt * BM;
    f32x16 acc[TN][TM];
#pragma unroll
    for (int jt = 0; jt < TN; ++jt)
#pragma unroll
      for (int it = 0; it < TM; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[jt][it][r] = 0.f;

    auto mma_chunk = [&](int buf) {
      const unsigned short* sA = smem + buf * (SA + SB);
      const unsigned short* sB = sA + SA;
#pragma unroll
      for (int q = 0; q < kBfBK / 16; ++q) {
        bfrag xa[TM], wb_[TN];
#pragma unroll
        for (int it = 0; it < TM; ++it) xa[it] = *(const bfrag*)(sA + (wm0 + it * 32 + l31) * kBfLdk + 16 * q + 8 * h);
#pragma unroll
        for (int jt = 0; jt < TN; ++jt) wb_[jt] = *(const bfrag*)(sB + (wn0 + jt * 32 + l31) * kBfLdk + 16 * q + 8 * h);
#pragma unroll
        for (int jt = 0; jt < TN; ++jt)
#pragma unroll
          for (int it = 0; it < TM; ++it) acc[jt][it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb_[jt], xa[it], acc[jt][it], 0, 0, 0);
      }
    };

    // FOLD: the epilogue's operands of this tile (rows r0 + 8 i of the wave's 32 TM) are requested in front of the LAST chunk's MFMAs: by then the
    // staging registers are free (requested ahead of the K loop they made this a 256-register kernel) and a chunk still covers their latency
    u32x4q f_d[FOLD ? 4 * TM : 1], f_x[FOLD ? 4 * TM : 1]; unsigned f_m[FOLD ? 4 * TM : 1];
    auto fold_loads = [&]() {
      if constexpr (FOLD) {
        const rsrc_t rs_dres = make_rsrc(fz.dres ? fz.dres : dst, fz.dres ? g.dst_bytes : 0u), rs_xbn = make_rsrc(fz.xbn, g.dst_bytes);
        const rsrc_t rs_mask = make_rsrc(fz.mask ? (const void*)fz.mask : (const void*)dst, fz.mask ? fz.mask_bytes : 0u);
        const int c = n0 + wn0 + cc * 8;
#pragma unroll
        for (int i = 0; i < 4 * TM; ++i) {
          const int m = m0 + wm0 + r0 + 8 * i;
          const bool ok = m < g.Mg && c < g.Cd;
          const unsigned off = ok ? (unsigned)(m * g.Cd + c) * 2u : kOob;
          f_d[i] = bload16(rs_dres, off); f_x[i] = bload16(rs_xbn, off);
          f_m[i] = fz.mask ? (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs_mask, (int)(ok ? (unsigned)(m * (g.Cd >> 3) + (c >> 3)) : kOob), 0, 0) : (ok ? 0xffu : 0u);
        }
      }
    };

    tile_setup(mt);
    if (nchunks > 0) {
      load_chunk(0, ra0, rb0);
      __syncthreads();                                          // the previous m-tile's LDS reads (K loop and epilogue image) are done
      store_chunk(0, ra0, rb0);
      __syncthreads();
      for (int ch = 0; ch + 1 < nchunks; ++ch) {
        load_chunk(ch + 1, ra0, rb0);
        mma_chunk(ch & 1);
        store_chunk((ch & 1) ^ 1, ra0, rb0);
        __syncthreads();
      }
      fold_loads();                                             // (the staging registers are free from here on)
      mma_chunk((nchunks - 1) & 1);
      __syncthreads();
    } else {
      fold_loads();
      __syncthreads();                                          // (a class without taps: zeros are stored; keep the image hand-off ordered)
    }

    // ---- epilogue.  D'[channel][pixel]: a lane holds pixel l31 of tile it, channels jt * 32 + 8 gq + 4 h + (0..3) in registers 4 gq .. 4 gq + 3
    unsigned short* ep = smem + wave * (32 * TM) * kBfLdk;      // per-wave image [32 TM pixels][64 channels], 144-byte rows
#pragma unroll
    for (int it = 0; it < TM; ++it)
#pragma unroll
      for (int jt = 0; jt < TN; ++jt)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          u32x2q pk;
          pk[0] = pk_bf16(acc[jt][it][4 * gq + 0], acc[jt][it][4 * gq + 1]);
          pk[1] = pk_bf16(acc[jt][it][4 * gq + 2], acc[jt][it][4 * gq + 3]);
          *(u32x2q*)(ep + (it * 32 + l31) * kBfLdk + jt * 32 + 8 * gq + 4 * h) = pk;
        }
    wave_lds_sync();
    const int c = n0 + wn0 + cc * 8;
    const unsigned coff = c < g.Cd ? (unsigned)c * 2u : kOob;
#pragma unroll
    for (int i = 0; i < 4 * TM; ++i) {
      const int row = r0 + 8 * i;
      const int m = m0 + wm0 + row;
      u32x4q v = *(const u32x4q*)(ep + row * kBfLdk + cc * 8);
      if (FOLD) {
        u32x4q o;
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) {
          float a0 = bf_lo(v[j2]) + bf_lo(f_d[i][j2]), a1 = bf_hi(v[j2]) + bf_hi(f_d[i][j2]);
          a0 = (f_m[i] >> (2 * j2)) & 1u ? a0 : 0.f; a1 = (f_m[i] >> (2 * j2 + 1)) & 1u ? a1 : 0.f;
          const unsigned w = pk_bf16(a0, a1);
          a0 = bf_lo(w); a1 = bf_hi(w);
          st_s[2 * j2] += a0; st_s[2 * j2 + 1] += a1;
          st_q[2 * j2] += a0 * ((bf_lo(f_x[i][j2]) - f_mu[2 * j2]) * f_is[2 * j2]);
          st_q[2 * j2 + 1] += a1 * ((bf_hi(f_x[i][j2]) - f_mu[2 * j2 + 1]) * f_is[2 * j2 + 1]);
          o[j2] = w;
        }
        v = o;
      } else if (STATS) {
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) {                        // rows past Mg are zeros (their source rows were out of range)
          const float a0 = bf_lo(v[j2]), a1 = bf_hi(v[j2]);
          st_s[2 * j2] += a0; st_q[2 * j2] += a0 * a0; st_s[2 * j2 + 1] += a1; st_q[2 * j2 + 1] += a1 * a1;
        }
      }
      unsigned poff;
      if (dense_dst) {
        poff = m < g.Mg ? (unsigned)(m * g.Cd) * 2u : kOob;
      } else {
        const int mm = m < g.Mg ? m : 0;
        const int t2 = fdiv(mm, g.dWm); const int mw = mm - t2 * g.Wm; const int n = fdiv(t2, g.dHm); const int mh = t2 - n * g.Hm;
        const int pix = (n * g.Hd + mh * g.dst_st + g.dph) * g.Wd + mw * g.dst_st + g.dpw;
        poff = m < g.Mg ? (unsigned)(pix * g.Cd) * 2u : kOob;
      }
      const unsigned off = (poff + coff) | ((poff | coff) & kOob);
      bstore16(v, rs_dst, off);
      if (g.zfill) {                                            // stride-2 1x1 data gradient: the pixels no output pixel reaches are zero
        const unsigned rowb = (unsigned)g.Cd * 2u, lineb = (unsigned)g.Wd * rowb;
        u32x4q z; z[0] = 0u; z[1] = 0u; z[2] = 0u; z[3] = 0u;
        bstore16(z, rs_dst, off + rowb); bstore16(z, rs_dst, off + lineb); bstore16(z, rs_dst, off + lineb + rowb);
      }
    }
    wave_lds_sync();
  }

  if (STATS || FOLD) {
    // lanes with equal (lane & 7) hold the same 8 channels: fold the 8 row groups, then the WM waves of a column block: part[bx][2][Cd]
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a = st_s[j], b = st_q[j];
      a += __shfl_xor(a, 8, kWave); b += __shfl_xor(b, 8, kWave);
      a += __shfl_xor(a, 16, kWave); b += __shfl_xor(b, 16, kWave);
      a += __shfl_xor(a, 32, kWave); b += __shfl_xor(b, 32, kWave);
      st_s[j] = a; st_q[j] = b;
    }
    __syncthreads();
    float* red = (float*)smem;                                  // [WM][2][BN]
    if (lane < 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        red[((wave / WN) * 2 + 0) * BN + wn0 + lane * 8 + j] = st_s[j];
        red[((wave / WN) * 2 + 1) * BN + wn0 + lane * 8 + j] = st_q[j];
      }
    }
    __syncthreads();
    for (int i = tid; i < 2 * BN; i += NT) {
      const int s = i / BN, cidx = i - s * BN;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) v += red[(w * 2 + s) * BN + cidx];
      if (n0 + cidx < g.Cd) part[((int64_t)bx * 2 + s) * g.Cd + n0 + cidx] = v;
    }
  }
}

template <int WM, int WN, int TM, bool STATS, bool TAPV, bool FOLD>
__global__ __launch_bounds__(64 * WM * WN, WM * WN == 4 ? 2 : 1) void conv_bf16_act_kernel(const unsigned short* __restrict__ src, const unsigned short* __restrict__ wgt,
                                                                      unsigned short* __restrict__ dst, ActGeo g, float* __restrict__ part, BfFuse fz) {
  bf16_act_body<WM, WN, TM, STATS, TAPV, FOLD>(src, wgt, dst, g, part, fz, blockIdx.x, gridDim.x, blockIdx.y);
}

// the parity classes of a strided data gradient as ONE launch (blockIdx.z = class, longest K first): see conv_f32_act_classes_kernel
template <int WM, int WN, int TM>
__global__ __launch_bounds__(kBfThreads, 2) void conv_bf16_act_classes_kernel(const unsigned short* __restrict__ src, const unsigned short* __restrict__ wgt,
                                                                              unsigned short* __restrict__ dst, ActGeoSet gs) {
  const ActGeo g = gs.g[blockIdx.z];
  bf16_act_body<WM, WN, TM, false, false, false>(src, wgt, dst, g, nullptr, BfFuse{}, blockIdx.x, gridDim.x, blockIdx.y);
}

// ---------------------------------------------------------------------------------------------------------------
// The same forward / data gradient with the operands travelling global -> LDS DIRECTLY (buffer_load_dwordx4 ... lds, "LDS-DMA") through a ring of
// NS stages of kLnBK = 64 k each, NS - 1 of them in flight, multiplied on v_mfma_f32_16x16x32_bf16.
//
// Why (profiles/r06_conv_bf16_pmc.md, profiles/EXPERIMENTS.md round 6 (3)): in the register-staged kernel above a wave spends 52 % of its cycles
// parked in s_waitcnt / s_barrier, and a second register set for a deeper look-ahead spills.  LDS-DMA needs no staging registers.  What it needs is
// WHOLE CACHE LINES per request: a first ring of four 32-k stages (64-byte rows) asked the L2 for every 128-byte line twice, a stage apart, and ran
// no faster than the register-staged kernel; 128-byte rows in two stages (64 KB at 128 x 128, two workgroups per CU) are 13 - 16 % faster on the
// layer3 / layer4 shapes, and the 16 x 16 x 32 MFMA shape -- same cycles per FLOP, higher clock held (MI355X_MICROARCH.md 'DVFS give-back' (7)) --
// another 5 - 8 % on the 3x3 ones.
//   * a DMA wave-instruction writes 64 lanes x 16 bytes = 1 KiB of LDS LINEARLY (M0 base + 16 lane) from PER-LANE source offsets: lane i lands in
//     row i >> 3, 16-byte slot i & 7 of an 8-row piece.  The rows cannot be padded, so the bank spread comes from the SOURCE: slot p of row R holds
//     the row's logical piece p ^ ((R >> 1) & 7), and the fragment reads (row = lane & 15, logical piece 4 q + (lane >> 4)) apply the same XOR: the
//     16 lanes a ds_read_b128 serves per LDS cycle ({0-3, 12-15, 20-27}, ...) land on 16 distinct 16-byte columns of the 256-byte bank row;
//   * out-of-range offsets (padding taps, row tails, channel tails: bit 31) make the DMA write ZEROS (measured: tools/microbench/lds_dma_oob.hip), so
//     the 32-bit offset scheme of the register-staged loader carries over unchanged;
//   * a stage is ready when every wave has waited for ITS pieces (counted s_waitcnt vmcnt: the younger stages stay in flight) and the workgroup has
//     met at ONE raw s_barrier, which also retires the stage read an iteration ago -- its buffer is the one the next request targets.
#ifndef LEC_BF_DBG
#define LEC_BF_DBG 0                      // 1 / 2: what-if builds for tools/whatif_conv_bf16.sh (wrong results by design); the product build is 0
#endif
constexpr int kLnBK = 64;                 // k per stage: 128-byte rows, one cache line per request
typedef __attribute__((address_space(3))) void lds_void;

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else static_assert(N < 0, "add the count");
}
__device__ __forceinline__ void lds_barrier() {                // LDS traffic of this wave is done; meet the workgroup (vmcnt untouched)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int WM, int WN, int TM, bool STATS, bool FOLD, int NS>
__device__ __forceinline__ void bf16_act_dma_body(const unsigned short* __restrict__ src, const unsigned short* __restrict__ wgt,
                                                  unsigned short* __restrict__ dst, const ActGeo& g, float* __restrict__ part, const BfFuse& fz,
                                                  const int bx, const int gdx, const int by) {
  static_assert((WM * WN == 4 || WM * WN == 8 || WM * WN == 16) && !(FOLD && STATS), "four, eight or sixteen waves; one statistics epilogue at a time");
  static_assert(NS == 2 || NS == 3, "two or three stages");
  constexpr int NW = WM * WN;
  constexpr bool PER_TILE = NW == 16;                          // sixteen waves: one partial row per m-TILE (row = the tile's index), no sums carried through the K loop
  constexpr bool LATE_FOLD = FOLD && NW == 16;               // 128 VGPRs per wave at four waves per SIMD: the fold's operands are fetched after the K loop
  constexpr int TN = 2;
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  constexpr int BKS = kLnBK, RB = 2 * BKS;                    // a stage's row: 64 k = 128 bytes
  constexpr int RPP = 1024 / RB;                              // rows per DMA piece
  constexpr int PA = BM / RPP / NW, PB = BN / RPP / NW;       // DMA pieces (1 KiB) of A / B per wave and stage
  constexpr int IPW = PA + PB;                                // DMA instructions per wave and stage
  constexpr int SA = BM * RB, SB = BN * RB;                   // bytes of a stage's A / B image
  constexpr int SS = SA + SB;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  char* const lds = (char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave / WN) * 32 * TM, wn0 = (wave % WN) * 32 * TN;
  const int n0 = by * BN;
  const int nst = g.Kg / BKS;                                  // stages per tile (Cs % BKS == 0: a stage lies inside one tap)
  const int mtiles = (g.Mg + BM - 1) / BM;
  const rsrc_t rs_src = make_rsrc(src, g.src_bytes), rs_wgt = make_rsrc(wgt, g.wgt_bytes), rs_dst = make_rsrc(dst, g.dst_bytes);
  const int rsc = g.RS * g.Cs;
  // DMA side: this lane's row inside a piece and its logical 16-byte slot (the XOR of the image, applied to the SOURCE): row R = 8 piece + (lane >> 3)
  const int prow = lane >> 3;
  auto pslot_of = [&](int piece) { return (lane & 7) ^ (((8 * piece + (lane >> 3)) >> 1) & 7); };
  unsigned wB[PB];
#pragma unroll
  for (int u = 0; u < PB; ++u) { const int co = n0 + RPP * (wave + NW * u) + prow; wB[u] = co < g.Cd ? (unsigned)(co * rsc + 8 * pslot_of(wave + NW * u)) * 2u : kOob; }
  // fragment side (v_mfma_f32_16x16x32_bf16: row lane & 15, k = 8 (lane >> 4) ... + 7 of a 32-k step): byte offset of (row, logical slot 4 q + (lane >> 4))
  // inside an image; the blocks start at multiples of 16 rows, which leave the XOR term alone
  const int l15 = lane & 15, h4 = lane >> 4;
  unsigned fo16[BKS / 32];
#pragma unroll
  for (int q = 0; q < BKS / 32; ++q) fo16[q] = (unsigned)(l15 * RB + 16 * ((4 * q + h4) ^ ((l15 >> 1) & 7)));
  const bool dense_dst = g.dst_st == 1;
  const int ntaps = g.na * g.nb;
  const int cc = lane & 7, r0 = lane >> 3;
  float st_s[8], st_q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { st_s[j] = 0.f; st_q[j] = 0.f; }
  float f_mu[8], f_is[8];
  auto load_moments = [&]() {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = n0 + wn0 + cc * 8 + j; const bool okc = c < g.Cd;
      f_mu[j] = okc ? fz.mean[c] : 0.f; f_is[j] = okc ? fz.invstd[c] : 0.f;
    }
  };
  if (FOLD && !LATE_FOLD) load_moments();
  auto flush_stats = [&](const int prow_) {                     // the workgroup's sums -> partial row prow_ (fixed order: lanes, then the WM wave rows)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a = st_s[j], b = st_q[j];
      a += __shfl_xor(a, 8, kWave); b += __shfl_xor(b, 8, kWave);
      a += __shfl_xor(a, 16, kWave); b += __shfl_xor(b, 16, kWave);
      a += __shfl_xor(a, 32, kWave); b += __shfl_xor(b, 32, kWave);
      st_s[j] = a; st_q[j] = b;
    }
    lds_barrier();
    float* red = (float*)smem;                                  // [WM][2][BN]
    if (lane < 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        red[((wave / WN) * 2 + 0) * BN + wn0 + lane * 8 + j] = st_s[j];
        red[((wave / WN) * 2 + 1) * BN + wn0 + lane * 8 + j] = st_q[j];
      }
    }
    lds_barrier();
    for (int i = tid; i < 2 * BN; i += 64 * NW) {
      const int sidx = i / BN, cidx = i - sidx * BN;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) v += red[(w * 2 + sidx) * BN + cidx];
      if (n0 + cidx < g.Cd) part[((int64_t)prow_ * 2 + sidx) * g.Cd + n0 + cidx] = v;
    }
  };

  for (int mt = bx; mt < mtiles; mt += gdx) {
    const int m0 = mt * BM;
    if (PER_TILE && (STATS || FOLD)) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { st_s[j] = 0.f; st_q[j] = 0.f; }
    }
    // rows this lane requests: piece wave + 4 u of the A image, row prow
    int rowoff[PA]; unsigned tapmask[PA];
#pragma unroll
    for (int u = 0; u < PA; ++u) {
      const int m = m0 + RPP * (wave + NW * u) + prow;
      const bool live = m < g.Mg;
      const int mm = live ? m : 0;
      const int t2 = fdiv(mm, g.dWm); const int mw = mm - t2 * g.Wm; const int n = fdiv(t2, g.dHm); const int mh = t2 - n * g.Hm;
      const int hb_ = mh * g.sst + g.oh0, wb_ = mw * g.sst + g.ow0;
      rowoff[u] = (((n * g.Hs + hb_) * g.Ws + wb_) << g.lgCs) * 2 + 16 * pslot_of(wave + NW * u);
      unsigned msk = 0;
      for (int t = 0; t < ntaps; ++t) {
        const int ta = fdiv(t, g.dnb), tb = t - ta * g.nb;
        const int hs = hb_ + g.sg * ta, ws = wb_ + g.sg * tb;
        msk |= ((unsigned)hs < (unsigned)g.Hs && (unsigned)ws < (unsigned)g.Ws ? 1u : 0u) << t;
      }
      tapmask[u] = live ? msk : 0u;
    }
    f32x4v acc16[2 * TN][2 * TM];                               // the wave's 64-channel x 32 TM-pixel tile as 16 x 16 blocks
#pragma unroll
    for (int jb = 0; jb < 2 * TN; ++jb)
#pragma unroll
      for (int ib = 0; ib < 2 * TM; ++ib)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc16[jb][ib][r] = 0.f;

    unsigned cur[PA];
    int cur_tap = -1;
    auto issue = [&](int st) {                                  // request stage st into ring slot st % NS
      const int k0 = st * BKS;
      const int tap = k0 >> g.lgCs, c0 = k0 & (g.Cs - 1);
      const int ta = fdiv(tap, g.dnb), tb = tap - ta * g.nb;
      if (tap != cur_tap) {
        cur_tap = tap;
        const int toff = (((g.sg * ta) * g.Ws + g.sg * tb) << g.lgCs) * 2;
        const unsigned tapbit = tap < 32 ? 1u << tap : 0u;
#pragma unroll
        for (int u = 0; u < PA; ++u) cur[u] = (tapmask[u] & tapbit) ? (unsigned)(rowoff[u] + toff) : kOob;
      }
      const int tw = (g.r0 + g.rstep * ta) * g.S + g.s0 + g.sstep * tb;
      const unsigned wsc = (unsigned)(tw * g.Cs + c0) * 2u;
      const unsigned c0b = (unsigned)c0 * 2u;
      char* base = lds + (st % NS) * SS;
#pragma unroll
      for (int u = 0; u < PA; ++u)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (lds_void*)(base + (wave + NW * u) * 1024), 16, (int)(cur[u] + c0b), 0, 0, 0);
#pragma unroll
      for (int u = 0; u < PB; ++u)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, (lds_void*)(base + SA + (wave + NW * u) * 1024), 16, (int)(wB[u] + wsc), 0, 0, 0);
    };
    auto compute = [&](int st) {
      const char* sA = lds + (st % NS) * SS;
      const char* sB = sA + SA;
#pragma unroll
      for (int q = 0; q < BKS / 32; ++q) {
        bfrag xa[2 * TM], wb_[2 * TN];
#pragma unroll
        for (int ib = 0; ib < 2 * TM; ++ib) xa[ib] = *(const bfrag*)(sA + (wm0 + ib * 16) * RB + fo16[q]);
#pragma unroll
        for (int jb = 0; jb < 2 * TN; ++jb) wb_[jb] = *(const bfrag*)(sB + (wn0 + jb * 16) * RB + fo16[q]);
#pragma unroll
        for (int jb = 0; jb < 2 * TN; ++jb)                     // D'[channel][pixel]: the weights are the A operand
#pragma unroll
          for (int ib = 0; ib < 2 * TM; ++ib) acc16[jb][ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb_[jb], xa[ib], acc16[jb][ib], 0, 0, 0);
      }
    };
    u32x4q f_d[FOLD ? 4 * TM : 1], f_x[FOLD ? 4 * TM : 1]; unsigned f_m[FOLD ? 4 * TM : 1];
    auto fold_loads = [&](const int i_lo = 0, const int i_hi = 4 * TM) {
      if constexpr (FOLD) {
        const rsrc_t rs_dres = make_rsrc(fz.dres ? fz.dres : dst, fz.dres ? g.dst_bytes : 0u), rs_xbn = make_rsrc(fz.xbn, g.dst_bytes);
        const rsrc_t rs_mask = make_rsrc(fz.mask ? (const void*)fz.mask : (const void*)dst, fz.mask ? fz.mask_bytes : 0u);
        const int c = n0 + wn0 + cc * 8;
#pragma unroll
        for (int i = 0; i < 4 * TM; ++i) {
          if (i < i_lo || i >= i_hi) continue;
          const int m = m0 + wm0 + r0 + 8 * i;
          const bool ok = m < g.Mg && c < g.Cd;
          const unsigned off = ok ? (unsigned)(m * g.Cd + c) * 2u : kOob;
          f_d[i] = bload16(rs_dres, off); f_x[i] = bload16(rs_xbn, off);
          f_m[i] = fz.mask ? (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs_mask, (int)(ok ? (unsigned)(m * (g.Cd >> 3) + (c >> 3)) : kOob), 0, 0) : (ok ? 0xffu : 0u);
        }
      }
    };

    lds_barrier();                                              // the previous tile's epilogue image has been read by every wave: the ring is free
#pragma unroll
    for (int st = 0; st < NS - 1; ++st) if (st < nst) issue(st);
    for (int c = 0; c < nst; ++c) {
      // stage c has landed for THIS wave once at most the younger stages' requests are outstanding
#if LEC_BF_DBG == 1                                              // what-if build (tools/whatif_conv_bf16.sh): no operand traffic after the ring's first fill
      if (c == 0) wait_vmcnt<0>();
      lds_barrier();
#else
      if constexpr (NS == 3) { if (c + 1 < nst) wait_vmcnt<IPW>(); else wait_vmcnt<0>(); }
      else wait_vmcnt<0>();
      lds_barrier();                                            // ... for every wave; and everyone is done reading stage c - 1
      if (c + NS - 1 < nst) issue(c + NS - 1);                   // into the slot stage c - 1 occupied
#endif
      if (!LATE_FOLD && c + 1 == nst) fold_loads();
#if LEC_BF_DBG != 2                                              // what-if build 2: the operand traffic alone, no LDS reads, no MFMAs
      compute(c);
#endif
    }
    if (!LATE_FOLD && nst == 0) fold_loads();
    lds_barrier();                                              // every wave is done with the last stage: the epilogue image may overwrite the ring

    // ---- epilogue (as in the register-staged kernel): D'[channel][pixel] -> per-wave LDS image [32 TM pixels][64 channels] -> 16-byte row segments
    unsigned short* ep = smem + wave * (32 * TM) * kBfLdk;
#pragma unroll
    for (int ib = 0; ib < 2 * TM; ++ib)                         // a block's lane holds D'[channel 4 (lane >> 4) + reg][pixel lane & 15]
#pragma unroll
      for (int jb = 0; jb < 2 * TN; ++jb) {
        u32x2q pk;
        pk[0] = pk_bf16(acc16[jb][ib][0], acc16[jb][ib][1]);
        pk[1] = pk_bf16(acc16[jb][ib][2], acc16[jb][ib][3]);
        *(u32x2q*)(ep + (ib * 16 + l15) * kBfLdk + jb * 16 + 4 * h4) = pk;
      }
    wave_lds_sync();
    if (LATE_FOLD) load_moments();
    const int c = n0 + wn0 + cc * 8;
    const unsigned coff = c < g.Cd ? (unsigned)c * 2u : kOob;
#pragma unroll
    for (int i = 0; i < 4 * TM; ++i) {
      // sixteen waves: the accumulators are in the image; the fold's operands come now, two rows' worth at a time (their registers are the accumulators')
      if (LATE_FOLD && (i & 1) == 0) fold_loads(i, i + 2);
      const int row = r0 + 8 * i;
      const int m = m0 + wm0 + row;
      u32x4q v = *(const u32x4q*)(ep + row * kBfLdk + cc * 8);
      if (FOLD) {
        u32x4q o;
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) {
          float a0 = bf_lo(v[j2]) + bf_lo(f_d[i][j2]), a1 = bf_hi(v[j2]) + bf_hi(f_d[i][j2]);
          a0 = (f_m[i] >> (2 * j2)) & 1u ? a0 : 0.f; a1 = (f_m[i] >> (2 * j2 + 1)) & 1u ? a1 : 0.f;
          const unsigned w = pk_bf16(a0, a1);
          a0 = bf_lo(w); a1 = bf_hi(w);
          st_s[2 * j2] += a0; st_s[2 * j2 + 1] += a1;
          st_q[2 * j2] += a0 * ((bf_lo(f_x[i][j2]) - f_mu[2 * j2]) * f_is[2 * j2]);
          st_q[2 * j2 + 1] += a1 * ((bf_hi(f_x[i][j2]) - f_mu[2 * j2 + 1]) * f_is[2 * j2 + 1]);
          o[j2] = w;
        }
        v = o;
      } else if (STATS) {
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) {
          const float a0 = bf_lo(v[j2]), a1 = bf_hi(v[j2]);
          st_s[2 * j2] += a0; st_q[2 * j2] += a0 * a0; st_s[2 * j2 + 1] += a1; st_q[2 * j2 + 1] += a1 * a1;
        }
      }
      unsigned poff;
      if (dense_dst) {
        poff = m < g.Mg ? (unsigned)(m * g.Cd) * 2u : kOob;
      } else {
        const int mm = m < g.Mg ? m : 0;
        const int t2 = fdiv(mm, g.dWm); const int mw = mm - t2 * g.Wm; const int n = fdiv(t2, g.dHm); const int mh = t2 - n * g.Hm;
        const int pix = (n * g.Hd + mh * g.dst_st + g.dph) * g.Wd + mw * g.dst_st + g.dpw;
        poff = m < g.Mg ? (unsigned)(pix * g.Cd) * 2u : kOob;
      }
      const unsigned off = (poff + coff) | ((poff | coff) & kOob);
      bstore16(v, rs_dst, off);
      if (g.zfill) {
        const unsigned rowb = (unsigned)g.Cd * 2u, lineb = (unsigned)g.Wd * rowb;
        u32x4q z; z[0] = 0u; z[1] = 0u; z[2] = 0u; z[3] = 0u;
        bstore16(z, rs_dst, off + rowb); bstore16(z, rs_dst, off + lineb); bstore16(z, rs_dst, off + lineb + rowb);
      }
    }
    if (PER_TILE && (STATS || FOLD)) flush_stats(mt);
  }

  if ((STATS || FOLD) && !PER_TILE) flush_stats(bx);
}

template <int WM, int WN, int TM, bool STATS, bool FOLD, int NS>
__global__ __launch_bounds__(64 * WM * WN, WM * WN == 4 ? 2 : 1) void conv_bf16_act_dma_kernel(const unsigned short* __restrict__ src, const unsigned short* __restrict__ wgt,
                                                                          unsigned short* __restrict__ dst, ActGeo g, float* __restrict__ part, BfFuse fz) {
  bf16_act_dma_body<WM, WN, TM, STATS, FOLD, NS>(src, wgt, dst, g, part, fz, blockIdx.x, gridDim.x, blockIdx.y);
}
template <int WM, int WN, int TM, int NS>
__global__ __launch_bounds__(kBfThreads, 2) void conv_bf16_act_dma_classes_kernel(const unsigned short* __restrict__ src, const unsigned short* __restrict__ wgt,
                                                                                  unsigned short* __restrict__ dst, ActGeoSet gs) {
  const ActGeo g = gs.g[blockIdx.z];
  bf16_act_dma_body<WM, WN, TM, false, false, NS>(src, wgt, dst, g, nullptr, BfFuse{}, blockIdx.x, gridDim.x, blockIdx.y);
}

// ---------------------------------------------------------------------------------------------------------------
// weights [Cout][RS][Cin] -> [Cin][RS][Cout] (the data gradient's k-contiguous B operand); 64 x 64 tiles through LDS
__global__ __launch_bounds__(256) void wt_transpose_kernel(const unsigned short* __restrict__ w, unsigned short* __restrict__ wt, int Cout, int RS, int Cin) {
  __shared__ unsigned short t[64][64 + 2];
  const int tap = blockIdx.z, co0 = blockIdx.y * 64, ci0 = blockIdx.x * 64;
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    const int co = co0 + r, ci = ci0 + c;
    t[r][c] = (co < Cout && ci < Cin) ? w[((int64_t)co * RS + tap) * Cin + ci] : (unsigned short)0;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63;                           // r: ci, c: co
    const int ci = ci0 + r, co = co0 + c;
    if (ci < Cin && co < Cout) wt[((int64_t)ci * RS + tap) * Cout + co] = t[c][r];
  }
}

// the same for EVERY convolution weight of a flat bf16 arena in one launch: table[l] = {element offset of the layer inside both arenas, Cout, RS, Cin,
// first tile of the layer}; a block finds its layer by bisection over the tile starts
__global__ __launch_bounds__(256) void wt_transpose_flat_kernel(const unsigned short* __restrict__ base, unsigned short* __restrict__ base_t,
                                                                const int* __restrict__ table, int nlayers) {
  __shared__ unsigned short t[64][64 + 2];
  const int b = blockIdx.x;
  int lo = 0, hi = nlayers - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (table[5 * mid + 4] <= b) lo = mid; else hi = mid - 1; }
  const int off = table[5 * lo], Cout = table[5 * lo + 1], RS = table[5 * lo + 2], Cin = table[5 * lo + 3];
  int rel = b - table[5 * lo + 4];
  const int nci = (Cin + 63) / 64, nco = (Cout + 63) / 64;
  const int tci = rel % nci; rel /= nci; const int tco = rel % nco; const int tap = rel / nco;
  const unsigned short* w = base + off; unsigned short* wt = base_t + off;
  const int co0 = tco * 64, ci0 = tci * 64;
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    const int co = co0 + r, ci = ci0 + c;
    t[r][c] = (co < Cout && ci < Cin) ? w[((int64_t)co * RS + tap) * Cin + ci] : (unsigned short)0;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    const int ci = ci0 + r, co = co0 + c;
    if (ci < Cin && co < Cout) wt[((int64_t)ci * RS + tap) * Cout + co] = t[c][r];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradient.  Tile BM output channels x BN columns (tap, ci); K = pixels in chunks of 32.  LDS images as in memory: A [32 px][BM + 32],
// B [32 px][BN + 32] bf16; fragments by ds_read_b64_tr_b16: a 16-lane group reads a 4 (k) x 16 (column) block, lane 4 q + p supplying the
// address of row q, columns 4 p .. 4 p + 3, and lane i receiving column i with the 4 k values -- two of them = the 8 consecutive k of an MFMA
// operand.  With row strides of 2 B + 64 bytes the four rows of a half-wave's two blocks fall on all 64 banks once.
__device__ __forceinline__ bfrag tr_frag(const unsigned short* p, const int ld4) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + ld4));
  bfrag f;
  f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
  return f;
}

template <int WM, int WN, int TM, int TN, bool DENSE>
__global__ __launch_bounds__(kBfThreads, 2) void conv_bf16_wgrad_kernel(const unsigned short* __restrict__ dy, const unsigned short* __restrict__ x,
                                                                        float* __restrict__ dw, WgGeo g) {
  static_assert(WM * WN == 4, "four waves per workgroup");
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, WBK = kBfWBK;
  constexpr int PA = BM / 8, PB = BN / 8;                     // 16-byte pieces per k row
  constexpr int NA = WBK * PA / kBfThreads, NB = WBK * PB / kBfThreads;
  static_assert(NA >= 1 && NB >= 1, "tile too small for 256 threads");
  constexpr int LDA = BM + 32, LDB = BN + 32;                 // elements: row strides of 2 B + 64 bytes
  constexpr int SA = WBK * LDA, SB = WBK * LDB;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave / WN) * 32 * TM, wn0 = (wave % WN) * 32 * TN;
  const int ntn = (g.Ng + BN - 1) / BN;
  const int nchunks_all = (g.Mpix + WBK - 1) / WBK;
  const rsrc_t rs_dy = make_rsrc(dy, g.dy_bytes), rs_x = make_rsrc(x, g.x_bytes);
  // transposed-read base of this lane inside a k step: row 8 (group >> 1) + q, column 16 (group & 1) + 4 p
  const int grp = lane >> 4, li = lane & 15;
  const int trA = (8 * (grp >> 1) + (li >> 2)) * LDA + 16 * (grp & 1) + 4 * (li & 3);
  const int trB = (8 * (grp >> 1) + (li >> 2)) * LDB + 16 * (grp & 1) + 4 * (li & 3);
  const int l31 = lane & 31, h = lane >> 5;

  const int per = (g.tiles * g.split + 7) / 8;
  for (int slot = blockIdx.x; slot < 8 * per; slot += gridDim.x) {
    const int wi = (slot & 7) * per + (slot >> 3);
    if (wi >= g.tiles * g.split) continue;
    const int tile = wi % g.tiles, sp = wi / g.tiles;
    const int tm = tile / ntn, tn = tile - tm * ntn;
    const int co0 = tm * BM, j0 = tn * BN;
    const int ch_lo = sp * g.chunks_per_split;
    const int ch_hi = min(nchunks_all, ch_lo + g.chunks_per_split);

    // A pieces (dY rows as they lie): v = tid + 256 u -> k row v / PA, channel piece v % PA
    unsigned aoff[NA];
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int v = tid + kBfThreads * u; const int kr = v / PA, cq = v - kr * PA; const int co = co0 + 8 * cq;
      aoff[u] = co < g.Cout ? (unsigned)(kr * g.Cout + co) * 2u : kOob;
    }
    // B pieces: the column piece (tap, 8 channels) of a thread is the same for all its rows
    const int jq = tid % PB, krB = tid / PB;
    const int jj = j0 + 8 * jq;
    const int tapL = jj >> g.lgCin, ciL = jj & (g.Cin - 1);
    const int rL = fdiv(tapL, g.dS);
    const int drL = rL - g.pad, dsL = tapL - rL * g.S - g.pad;
    const bool okL = jj < g.Ng;
    const int laneoff = ((drL * g.W + dsL) * g.Cin + ciL) * 2;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int it = 0; it < TM; ++it)
#pragma unroll
      for (int jt = 0; jt < TN; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[it][jt][r] = 0.f;

    u32x4q ra[NA], rb[NB];
    auto load_chunk = [&](int ch) {
      const int mbase = ch * WBK;
      const unsigned abase = (unsigned)(mbase * g.Cout) * 2u;
#pragma unroll
      for (int u = 0; u < NA; ++u) ra[u] = bload16(rs_dy, aoff[u] + abase);
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int m = mbase + krB + (kBfThreads / PB) * u;
        if (DENSE) {
          rb[u] = bload16(rs_x, (okL && m < g.Mpix) ? (unsigned)(m * g.Cin + jj) * 2u : kOob);
        } else {
          const bool live = m < g.Mpix;
          const int mm = live ? m : 0;
          const int t2 = fdiv(mm, g.dWo); const int wo = mm - t2 * g.Wo; const int n = fdiv(t2, g.dHo); const int ho = t2 - n * g.Ho;
          const int h0 = ho * g.stride, w0 = wo * g.stride;
          const bool ok = live && okL && (unsigned)(h0 + drL) < (unsigned)g.H && (unsigned)(w0 + dsL) < (unsigned)g.W;
          rb[u] = bload16(rs_x, ok ? (unsigned)(((n * g.H + h0) * g.W + w0) * g.Cin * 2 + laneoff) : kOob);
        }
      }
    };
    auto store_chunk = [&](int buf) {
      unsigned short* base = smem + buf * (SA + SB);
#pragma unroll
      for (int u = 0; u < NA; ++u) { const int v = tid + kBfThreads * u; const int kr = v / PA, cq = v - kr * PA; *(u32x4q*)(base + kr * LDA + 8 * cq) = ra[u]; }
#pragma unroll
      for (int u = 0; u < NB; ++u) *(u32x4q*)(base + SA + (krB + (kBfThreads / PB) * u) * LDB + 8 * jq) = rb[u];
    };
    if (ch_lo < ch_hi) {
      load_chunk(ch_lo);
      __syncthreads();
      store_chunk(0);
      __syncthreads();
      for (int ch = ch_lo; ch < ch_hi; ++ch) {
        const int buf = (ch - ch_lo) & 1;
        const unsigned short* sA = smem + buf * (SA + SB);
        const unsigned short* sB = sA + SA;
        if (ch + 1 < ch_hi) load_chunk(ch + 1);
#pragma unroll
        for (int ks = 0; ks < WBK / 16; ++ks) {
          bfrag fa[TM], fb[TN];
#pragma unroll
          for (int it = 0; it < TM; ++it) fa[it] = tr_frag(sA + trA + 16 * ks * LDA + wm0 + it * 32, 4 * LDA);
#pragma unroll
          for (int jt = 0; jt < TN; ++jt) fb[jt] = tr_frag(sB + trB + 16 * ks * LDB + wn0 + jt * 32, 4 * LDB);
#pragma unroll
          for (int it = 0; it < TM; ++it)
#pragma unroll
            for (int jt = 0; jt < TN; ++jt) acc[it][jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[it], fb[jt], acc[it][jt], 0, 0, 0);
        }
        if (ch + 1 < ch_hi) store_chunk(buf ^ 1);
        __syncthreads();
      }
      // D[co][j]: column j on the lanes, rows co = (r & 3) + 8 (r >> 2) + 4 h on the registers
#pragma unroll
      for (int it = 0; it < TM; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + wm0 + it * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (co < g.Cout) {
#pragma unroll
            for (int jt = 0; jt < TN; ++jt) {
              const int jc = j0 + wn0 + jt * 32 + l31;
              if (g.dCin == g.Cin) {
                if (jc < g.Ng) atomicAdd(dw + (int64_t)co * g.Ng + jc, acc[it][jt][r]);
              } else {                                          // the stem: the padded input channels have no slot in dw
                const int tp = jc >> g.lgCin, ci = jc & (g.Cin - 1);
                if (jc < g.Ng && ci < g.dCin) atomicAdd(dw + ((int64_t)co * g.RS + tp) * g.dCin + ci, acc[it][jt][r]);
              }
            }
          }
        }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// The stem (7x7 / stride 2 / pad 3, 8 -> 64 channels, 3 of the 8 real) as its own forward kernel.  The generic kernel gathers every 16-byte piece (one
// tap of one pixel) on its own: each input pixel travels through the L1 49 / 4 times and the layer runs at a quarter of what its bytes need (844 us against
// an HBM floor of ~225 us at 512 rows).  Here a tile is ONE output row: the 7 input rows it needs are read once (8 bytes per pixel: channels 0..3, the
// fourth a zero -- channels 4..7 of the padded input are never fetched) into a compact LDS patch, and every MFMA operand is one 16-byte LDS read.
//   * patch [7 rows][W + 8 pixels] of 8 bytes, pixel column c at position c + 3: the 16 bytes at position 2 ow + 2 q are the pixels of taps (r, 2 q) and
//     (r, 2 q + 1) of output pixel ow -- a k group of v_mfma_f32_16x16x32_bf16; the four groups of a K step are the four tap pairs of filter row r
//     (tap (r, 7) does not exist: zero weights), so the 49 taps are SEVEN K steps instead of the thirteen the 8-channel pixels would take;
//   * the pixels of the NEXT tile are requested into registers before this tile's products and written to the patch after them; a tile waits for them with
//     a counted vmcnt that leaves its predecessor's output stores in flight;
//   * wave v owns output channels 16 v .. 16 v + 15; its 7 weight fragments stay in registers for the whole launch (no LDS for the weights);
//   * D'[channel][pixel] -> bf16 -> an LDS row image [Wo][64] -> 128-byte rows to memory; the BatchNorm statistics from the rounded values, one partial row
//     per workgroup (lec_bn_fwd_prestat's layout), as in the family's epilogue.
#ifndef LEC_STEM_DBG
#define LEC_STEM_DBG 0                    // 1 / 3: what-if builds (no products / no output stores); the product build is 0
#endif
struct StemGeo {
  int N, H, W, Ho, Wo, PC, tiles, npix;     // PC = W + 8 patch positions per row; npix = 7 * (W + 6) pixels a tile requests
  uint32_t x_bytes, w_bytes, y_bytes;
  FastDiv dRow, dHo;                        // / (W + 6), / Ho
};
typedef unsigned int u32x2r __attribute__((ext_vector_type(2)));

template <bool STATS, int NPB>              // NPB = Wo / 16: 16-pixel blocks of an output row (7 at 224 x 224 images)
__global__ __launch_bounds__(256, 3) void conv_bf16_stem_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ w,
                                                                unsigned short* __restrict__ y, StemGeo g, float* __restrict__ part) {
  constexpr int NLD = (7 * (32 * NPB + 6) + 255) / 256;        // pixel loads per thread and tile
  constexpr int PATCH = 7 * (32 * NPB + 8) * 8;                // bytes
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  char* const patch = (char*)smem;
  char* const outimg = patch + ((PATCH + 255) & ~255);         // [Wo][64] bf16
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kq = lane >> 4;
  const rsrc_t rs_x = make_rsrc(x, g.x_bytes), rs_w = make_rsrc(w, g.w_bytes), rs_y = make_rsrc(y, g.y_bytes);
  // weights: fragment r = taps (r, 2 kq), (r, 2 kq + 1) x channels 0..3 of output channel 16 wave + l15 (w [64][49][8])
  bfrag wf[7];
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    const unsigned o0 = (unsigned)(((16 * wave + l15) * 49 + r * 7 + 2 * kq) * 16);
    const u32x2r lo = __builtin_amdgcn_raw_buffer_load_b64(rs_w, (int)o0, 0, 0);
    const u32x2r hi = __builtin_amdgcn_raw_buffer_load_b64(rs_w, (int)(2 * kq + 1 < 7 ? o0 + 16u : kOob), 0, 0);
    u32x4q v; v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
    wf[r] = __builtin_bit_cast(bfrag, v);
  }
  // loader: pixel idx = tid + 256 u of the tile's 7 x (W + 6) window -> (row r, column c - 3)
  int rel[NLD]; int rr[NLD]; unsigned pofs[NLD];
#pragma unroll
  for (int u = 0; u < NLD; ++u) {
    const int idx = tid + 256 * u;
    const int r = fdiv(idx, g.dRow); const int cpos = idx - r * (g.W + 6);     // position = column + 3
    const int col = cpos - 3;
    const bool ok = idx < g.npix && (unsigned)col < (unsigned)g.W;
    rel[u] = ok ? (r * g.W + col) * 16 : -1;
    rr[u] = r;
    pofs[u] = idx < g.npix ? (unsigned)((r * g.PC + cpos) * 8) : 0xffffffffu;
  }
  u32x2r stg[NLD];
  auto request = [&](int t) {
    const int n = fdiv(t, g.dHo); const int oh = t - n * g.Ho;
    const int row0 = 2 * oh - 3;
    const int base = (n * g.H + row0) * g.W * 16;              // (may point before the image: the row test below covers it)
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const bool ok = rel[u] >= 0 && (unsigned)(row0 + rr[u]) < (unsigned)g.H;
      stg[u] = __builtin_amdgcn_raw_buffer_load_b64(rs_x, (int)(ok ? (unsigned)(base + rel[u]) : kOob), 0, 0);
    }
  };
  const int cc = tid & 7, r0 = tid >> 3;
  float st_s[8], st_q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { st_s[j] = 0.f; st_q[j] = 0.f; }
  // zero the patch once: its last two positions of every row are never written (and never multiplied by a non-zero weight, but must not hold a NaN)
  for (int i = tid; i < PATCH / 16; i += 256) { u32x4q z; z[0] = 0u; z[1] = 0u; z[2] = 0u; z[3] = 0u; *(u32x4q*)(patch + 16 * i) = z; }

  // Tile walk: workgroup ids go round-robin over the 8 XCDs (one L2 each); XCD x owns the CONTIGUOUS run of output rows [x per, (x + 1) per), so the rows in
  // flight on an XCD at any moment are neighbours and share their 7-row windows in that L2
  const int per = (g.tiles + 7) >> 3;
  auto tile_of = [&](int slot) { return (slot & 7) * per + (slot >> 3); };
  const int nslots = 8 * per;
  bool first = true;
  { const int t0 = tile_of(blockIdx.x); if ((int)blockIdx.x < nslots && t0 < g.tiles) request(t0); }
  for (int slot = blockIdx.x; slot < nslots; slot += gridDim.x) {
    const int t = tile_of(slot);
    const int tn = slot + (int)gridDim.x < nslots ? tile_of(slot + gridDim.x) : g.tiles;
    if (t >= g.tiles) continue;                                 // (only the last run can be short: its tail slots have no tile, and neither have their successors)
    // this tile's pixels were requested BEFORE the previous tile's output stores: wait for them and leave those stores (the wave's youngest 3 / 4, 2, 1
    // operations) in flight
    if (!first && LEC_STEM_DBG != 3) {
      if (NPB == 7) { if (wave < 2) wait_vmcnt<4>(); else wait_vmcnt<3>(); }
      else if (NPB == 4) wait_vmcnt<2>();
      else wait_vmcnt<1>();
    } else wait_vmcnt<0>();
    first = false;
    lds_barrier();                                              // the previous tile's patch and row image have been read by every wave
#pragma unroll
    for (int u = 0; u < NLD; ++u) if (pofs[u] != 0xffffffffu) *(u32x2r*)(patch + pofs[u]) = stg[u];
    lds_barrier();
    if (tn < g.tiles) request(tn);
    const char* pb = patch + l15 * 16 + kq * 16;
    f32x4v acc[NPB];
#pragma unroll
    for (int ib = 0; ib < NPB; ++ib)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[ib][r] = 0.f;
#pragma unroll
    for (int r = 0; r < (LEC_STEM_DBG == 1 ? 0 : 7); ++r) {
      bfrag xa[NPB];
#pragma unroll
      for (int ib = 0; ib < NPB; ++ib) xa[ib] = *(const bfrag*)(pb + r * (32 * NPB + 8) * 8 + ib * 256);
#pragma unroll
      for (int ib = 0; ib < NPB; ++ib) acc[ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[r], xa[ib], acc[ib], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);                        // (left alone the scheduler hoists every fragment read above the first MFMA and spills)
    }
    // D'[channel 4 kq + r][pixel l15] of block ib -> the row image
#pragma unroll
    for (int ib = 0; ib < NPB; ++ib) {
      u32x2q pk;
      pk[0] = pk_bf16(acc[ib][0], acc[ib][1]); pk[1] = pk_bf16(acc[ib][2], acc[ib][3]);
      *(u32x2q*)(outimg + (16 * ib + l15) * 128 + (16 * wave + 4 * kq) * 2) = pk;
    }
    lds_barrier();
    const int n = fdiv(t, g.dHo); const int oh = t - n * g.Ho;
    const unsigned rowbase = (unsigned)((n * g.Ho + oh) * g.Wo) * 128u;
    for (int px = r0; px < 16 * NPB; px += 32) {
      const u32x4q v = *(const u32x4q*)(outimg + px * 128 + cc * 16);
      if (LEC_STEM_DBG != 3) bstore16(v, rs_y, rowbase + (unsigned)px * 128u + (unsigned)cc * 16u);
      if (STATS) {
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) {
          const float a0 = bf_lo(v[j2]), a1 = bf_hi(v[j2]);
          st_s[2 * j2] += a0; st_q[2 * j2] += a0 * a0; st_s[2 * j2 + 1] += a1; st_q[2 * j2 + 1] += a1 * a1;
        }
      }
    }
  }
  if (STATS) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a = st_s[j], b = st_q[j];
      a += __shfl_xor(a, 8, kWave); b += __shfl_xor(b, 8, kWave);
      a += __shfl_xor(a, 16, kWave); b += __shfl_xor(b, 16, kWave);
      a += __shfl_xor(a, 32, kWave); b += __shfl_xor(b, 32, kWave);
      st_s[j] = a; st_q[j] = b;
    }
    lds_barrier();
    float* red = (float*)smem;                                  // [4 waves][2][64]
    if (lane < 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { red[(wave * 2 + 0) * 64 + lane * 8 + j] = st_s[j]; red[(wave * 2 + 1) * 64 + lane * 8 + j] = st_q[j]; }
    }
    lds_barrier();
    if (tid < 128) {
      const int sidx = tid >> 6, cidx = tid & 63;
      part[((int64_t)blockIdx.x * 2 + sidx) * 64 + cidx] = red[(0 * 2 + sidx) * 64 + cidx] + red[(1 * 2 + sidx) * 64 + cidx] + red[(2 * 2 + sidx) * 64 + cidx] + red[(3 * 2 + sidx) * 64 + cidx];
    }
  }
}


static int launch_bf16_stem(const unsigned short* x, const unsigned short* w, unsigned short* y, int N, int H, int W, float* part, int64_t part_bytes, int* nparts, hipStream_t st) {
  StemGeo g;
  g.N = N; g.H = H; g.W = W; g.Ho = H / 2; g.Wo = W / 2; g.PC = W + 8; g.tiles = N * g.Ho; g.npix = 7 * (W + 6);
  g.x_bytes = (uint32_t)((int64_t)N * H * W * 16); g.w_bytes = (uint32_t)(64 * 49 * 16); g.y_bytes = (uint32_t)((int64_t)N * g.Ho * g.Wo * 128);
  g.dRow = make_fastdiv(W + 6); g.dHo = make_fastdiv(g.Ho);
  // three workgroups per CU and more (27 KB of LDS at 224-pixel rows): 768 of them where the partial rows fit the caller's buffer (one row per workgroup)
  int gx = 768;
  if (part && part_bytes < (int64_t)gx * 2 * 64 * (int64_t)sizeof(float)) gx = kCfMaxPart;
  if (gx > g.tiles) gx = g.tiles;
  const size_t lds = (size_t)(((7 * (W + 8) * 8) + 255) & ~255) + (size_t)g.Wo * 128;
#define LEC_STEM_LAUNCH(NPB_) do { if (part) hipLaunchKernelGGL((conv_bf16_stem_kernel<true, NPB_>), dim3(gx), dim3(256), lds, st, x, w, y, g, part); \
                                   else hipLaunchKernelGGL((conv_bf16_stem_kernel<false, NPB_>), dim3(gx), dim3(256), lds, st, x, w, y, g, part); } while (0)
  if (W == 224) LEC_STEM_LAUNCH(7); else if (W == 128) LEC_STEM_LAUNCH(4); else LEC_STEM_LAUNCH(2);
#undef LEC_STEM_LAUNCH
  if (nparts) *nparts = gx;
  LEC_CHECK_LAUNCH("conv_bf16_stem_kernel");
  return LEC_OK;
}

// The stem's weight gradient on the same compact patch: dw[co][(r, s)][ci] = sum over output pixels of dy[p][co] * x[2 oh + r - 3][2 ow + s - 3][ci].  A tile is
// one output row; K = its pixels (zero rows up to a multiple of 32).  The "im2col" operand needs no copy: for filter row r, the 32 bf16 of pixel p -- 8 taps
// x 4 channels, tap 7 and channel 3 without a slot in dw -- are the 64 bytes at patch[r][2 p ...], so the K-major matrix is the patch itself with a row stride
// of 16 bytes, and ds_read_b64_tr_b16 transposes it on the way to the MFMA like any other K-major tile.  dY [pixel][64] is stored with its 32-byte granules
// XORed by ((p >> 1) & 1) | (((p >> 3) & 1) << 1) so that the transposed reads of rows {0..3, 8..11} meet all 64 banks once.  7 rows x 2 blocks of 16 columns
// = 14 column blocks over four waves (4, 4, 4, 2), 4 blocks of output channels each; sums stay in registers over the workgroup's tiles and leave as float atomics.
template <int NPB>
__global__ __launch_bounds__(256, 3) void conv_bf16_stem_wgrad_kernel(const unsigned short* __restrict__ dy, const unsigned short* __restrict__ x,
                                                                      float* __restrict__ dw, StemGeo g, int dcin) {
  constexpr int WO = 16 * NPB, KS = (WO + 31) / 32, KP = 32 * KS;
  constexpr int NLD = (7 * (32 * NPB + 6) + 255) / 256, NDY = (WO * 8 + 255) / 256;
  constexpr int ROWB = (32 * NPB + 8) * 8, PATCH = (7 * ROWB + 255) & ~255;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  char* const patch = (char*)smem;
  char* const dyt = patch + PATCH + 512;                       // [KP][64] bf16 (the 512 bytes between: the last K step's reads past the patch's end stay inside LDS)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = lane >> 4, li = lane & 15;
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
    pofs[u] = idx < g.npix ? (unsigned)(r * ROWB + cpos * 8) : 0xffffffffu;
  }
  unsigned dofs[NDY];                                           // LDS byte offset of this thread's dY pieces (pixel idx >> 3, channels 8 (idx & 7) ..)
#pragma unroll
  for (int u = 0; u < NDY; ++u) {
    const int idx = tid + 256 * u; const int p = idx >> 3, c8 = idx & 7;
    const int f = ((p >> 1) & 1) | (((p >> 3) & 1) << 1);
    dofs[u] = idx < WO * 8 ? (unsigned)(p * 128 + ((((c8 >> 1) ^ f) << 1) | (c8 & 1)) * 16) : 0xffffffffu;
  }
  u32x2r stx[NLD]; u32x4q sty[NDY];
  auto request = [&](int t) {
    const int n = fdiv(t, g.dHo); const int oh = t - n * g.Ho;
    const int row0 = 2 * oh - 3;
    const int base = (n * g.H + row0) * g.W * 16;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const bool ok = rel[u] >= 0 && (unsigned)(row0 + rr[u]) < (unsigned)g.H;
      stx[u] = __builtin_amdgcn_raw_buffer_load_b64(rs_x, (int)(ok ? (unsigned)(base + rel[u]) : kOob), 0, 0);
    }
    const unsigned dbase = (unsigned)(t * WO) * 128u;
#pragma unroll
    for (int u = 0; u < NDY; ++u) sty[u] = bload16(rs_dy, dofs[u] != 0xffffffffu ? dbase + (unsigned)(tid + 256 * u) * 16u : kOob);
  };
  for (int i = tid; i < (PATCH + 512 + KP * 128) / 16; i += 256) { u32x4q z; z[0] = 0u; z[1] = 0u; z[2] = 0u; z[3] = 0u; *(u32x4q*)(patch + 16 * i) = z; }

  // fragment addresses: dY^T block mb (rows = channels 16 mb ..), K step ks: k row 32 ks + 8 grp + (li >> 2) (+ 4), columns 4 (li & 3) ..
  const int fA = ((li >> 3) & 1) | ((grp & 1) << 1);
  const unsigned aoff = (unsigned)((8 * grp + (li >> 2)) * 128 + (li & 3) * 8);
  const unsigned boff = (unsigned)((8 * grp + (li >> 2)) * 16 + (li & 3) * 8);
  f32x4v acc[4][4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mb][nb][r] = 0.f;

  const int per = (g.tiles + 7) >> 3;
  auto tile_of = [&](int slot) { return (slot & 7) * per + (slot >> 3); };
  const int nslots = 8 * per;
  { const int t0 = tile_of(blockIdx.x); if ((int)blockIdx.x < nslots && t0 < g.tiles) request(t0); }
  for (int slot = blockIdx.x; slot < nslots; slot += gridDim.x) {
    const int t = tile_of(slot);
    const int tn = slot + (int)gridDim.x < nslots ? tile_of(slot + gridDim.x) : g.tiles;
    if (t >= g.tiles) continue;
    wait_vmcnt<0>();
    lds_barrier();                                              // the previous tile's operands have been read by every wave
#pragma unroll
    for (int u = 0; u < NLD; ++u) if (pofs[u] != 0xffffffffu) *(u32x2r*)(patch + pofs[u]) = stx[u];
#pragma unroll
    for (int u = 0; u < NDY; ++u) if (dofs[u] != 0xffffffffu) *(u32x4q*)(dyt + dofs[u]) = sty[u];
    lds_barrier();
    if (tn < g.tiles) request(tn);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bfrag fa[4], fb[4];
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) fa[mb] = tr_frag((const unsigned short*)(dyt + ks * 32 * 128 + aoff + ((mb ^ fA) * 32)), 4 * 64);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const int nbg = 4 * wave + nb;                          // column block: filter row nbg >> 1, columns 16 (nbg & 1) .. of its 32
        if (nbg < 14) fb[nb] = tr_frag((const unsigned short*)(patch + (nbg >> 1) * ROWB + ks * 32 * 16 + boff + (nbg & 1) * 32), 4 * 8);
      }
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
          if (4 * wave + nb < 14) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mb], fb[nb], acc[mb][nb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // D[channel 16 mb + 4 grp + r][column li of block nbg]: column j = 16 (nbg & 1) + li = tap s = j >> 2, input channel j & 3 of filter row nbg >> 1
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    const int nbg = 4 * wave + nb;
    if (nbg >= 14) continue;
    const int j = 16 * (nbg & 1) + li; const int s = j >> 2, ci = j & 3;
    if (s >= 7 || ci >= dcin) continue;
    const int tap = (nbg >> 1) * 7 + s;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(dw + ((int64_t)(16 * mb + 4 * grp + r) * 49 + tap) * dcin + ci, acc[mb][nb][r]);
  }
}

static int launch_bf16_stem_wgrad(const unsigned short* dy, const unsigned short* x, float* dw, int N, int H, int W, int dcin, hipStream_t st) {
  StemGeo g;
  g.N = N; g.H = H; g.W = W; g.Ho = H / 2; g.Wo = W / 2; g.PC = W + 8; g.tiles = N * g.Ho; g.npix = 7 * (W + 6);
  g.x_bytes = (uint32_t)((int64_t)N * H * W * 16); g.w_bytes = 0; g.y_bytes = (uint32_t)((int64_t)N * g.Ho * g.Wo * 128);
  g.dRow = make_fastdiv(W + 6); g.dHo = make_fastdiv(g.Ho);
  int gx = 512; if (gx > g.tiles) gx = g.tiles;
  const int KP = 32 * ((g.Wo + 31) / 32);
  const size_t lds = (size_t)(((7 * (W + 8) * 8) + 255) & ~255) + 512 + (size_t)KP * 128;
  if (W == 224) hipLaunchKernelGGL((conv_bf16_stem_wgrad_kernel<7>), dim3(gx), dim3(256), lds, st, dy, x, dw, g, dcin);
  else if (W == 128) hipLaunchKernelGGL((conv_bf16_stem_wgrad_kernel<4>), dim3(gx), dim3(256), lds, st, dy, x, dw, g, dcin);
  else hipLaunchKernelGGL((conv_bf16_stem_wgrad_kernel<2>), dim3(gx), dim3(256), lds, st, dy, x, dw, g, dcin);
  LEC_CHECK_LAUNCH("conv_bf16_stem_wgrad_kernel");
  return LEC_OK;
}

// ---------------------------------------------------------------------------------------------------------------
static inline int conv_bf16_check(const char* who, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad) {
  LEC_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && R > 0 && S > 0 && (stride == 1 || stride == 2) && pad >= 0 && pad < R && pad < S,
                "%s: bad geometry N=%d H=%d W=%d Cin=%d Cout=%d R=%d S=%d stride=%d pad=%d", who, N, H, W, Cin, Cout, R, S, stride, pad);
  LEC_CHECK_ARG(ilog2_exact(Cin) >= 3, "%s: Cin must be a power of two >= 8 (pad the stem's 3 channels to 8), got %d", who, Cin);
  LEC_CHECK_ARG(ilog2_exact(Cout) >= 3, "%s: Cout must be a power of two >= 8, got %d", who, Cout);
  const int64_t Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
  LEC_CHECK_ARG(Ho > 0 && Wo > 0, "%s: empty output", who);
  LEC_CHECK_ARG((int64_t)N * H * W * Cin * 2 < (1ll << 31) && (int64_t)N * Ho * Wo * Cout * 2 < (1ll << 31) && (int64_t)Cout * R * S * Cin * 2 < (1ll << 30),
                "%s: a tensor of this layer reaches 2 GiB (N=%d H=%d W=%d Cin=%d Cout=%d): split the batch", who, N, H, W, Cin, Cout);
  return LEC_OK;
}

template <bool STATS, bool FOLD>
static int launch_bf16_act(const unsigned short* src, const unsigned short* wgt, unsigned short* dst, const ActGeo& g, float* part, int* nparts,
                           hipStream_t st, const BfFuse& fz = BfFuse{}) {
  const bool narrow = g.Cd <= 64;
  const bool tapv = g.Cs % kBfBK != 0;
  const int BM = 128, BN = narrow ? 64 : 128;
  const int mtiles = (g.Mg + BM - 1) / BM, ntiles = (g.Cd + BN - 1) / BN;
  // grid: the column tiles of one m-tile are co-resident (and, gx a multiple of 8, on ONE XCD: the activation tile they share is fetched from
  // HBM once and re-read from that XCD's L2); workgroups walk their m-tiles
  int gx = (1024 / ntiles) & ~7;
  if (gx < 8) gx = 8;
  if (STATS || FOLD) { if (gx > kCfMaxPart) gx = kCfMaxPart; }
  if (gx > mtiles) gx = mtiles;
  if (gx < 1) gx = 1;
  LEC_CHECK_ARG(tapv || g.na * g.nb <= 32, "conv_bf16: more than 32 taps per launch need the per-piece tap path");
  LEC_CHECK_ARG(!(FOLD && tapv), "conv_bf16: the fold needs source channels that are a multiple of the K chunk (%d)", kBfBK);
  const size_t lds = (size_t)2 * (BM + BN) * kBfLdk * 2;
  const dim3 grid(gx, ntiles), blk(kBfThreads);
  ActGeo gg = g; gg.xcd_per = 0;
  if (!tapv && tuning().bf_dma) {
    // Tile policy (LEC_BF16_TILE: 1 = the two rules below (default), 0 = 128 x 128 always, 2 = 128 x 256 wherever >= 256 destination channels, 3 = 256 x 256
    // wherever it fits; same-box tables in profiles/EXPERIMENTS.md round 6 (3) items 3 and 8).
    const int bt = tuning().bf_tile;
    // (a) 256 x 256 tiles (sixteen waves, ONE workgroup per CU, two 64 KB stages): 128 FLOP per operand byte instead of 64 -- the CU's L1 stops being co-critical
    // with its matrix pipe.  It pays where the K loop is long (>= 1 024: the 3x3 layers, 1x1 from >= 1 024 channels) and the launch still has a tile for most
    // CUs (>= 160); the statistics / fold sums leave as one partial row per m-TILE, so there must be <= 512 of those.
    {
      const int mt2 = (g.Mg + 255) / 256, nt2 = (g.Cd + 255) / 256;
      const bool fits = !narrow && g.Cd >= 256 && mt2 <= kCfMaxPart;
      if (fits && (bt == 3 || (bt == 1 && g.Kg >= 1024 && mt2 * nt2 >= 160))) {
        int gx2 = (256 / nt2) & ~7; if (gx2 < 8) gx2 = 8; if (gx2 > mt2) gx2 = mt2;
        const size_t ring = (size_t)2 * 512 * 2 * kLnBK, epi = (size_t)16 * 64 * kBfLdk * 2;
        hipLaunchKernelGGL((conv_bf16_act_dma_kernel<4, 4, 2, STATS, FOLD, 2>), dim3(gx2, nt2), dim3(1024), ring > epi ? ring : epi, st, src, wgt, dst, gg, part, fz);
        if (nparts) *nparts = mt2;
        LEC_CHECK_LAUNCH("conv_bf16_act_dma_kernel");
        return LEC_OK;
      }
    }
    // (b) a 1x1 layer with exactly 256 destination channels takes ONE 128 x 256 column tile (eight waves, one workgroup per CU, three stages): its activations
    // are read once instead of twice and its K loop is too short to miss the second workgroup
    if (!narrow && ((bt == 1 && g.Cd == 256 && g.na * g.nb == 1) || (bt == 2 && g.Cd >= 256))) {
      const int BM2 = 128, BN2 = 256;
      const int mt2 = (g.Mg + BM2 - 1) / BM2, nt2 = (g.Cd + BN2 - 1) / BN2;
      int gx2 = (256 / nt2) & ~7; if (gx2 < 8) gx2 = 8; if (gx2 > mt2) gx2 = mt2;
      hipLaunchKernelGGL((conv_bf16_act_dma_kernel<2, 4, 2, STATS, FOLD, 3>), dim3(gx2, nt2), dim3(512), (size_t)3 * (BM2 + BN2) * 2 * kLnBK, st, src, wgt, dst, gg, part, fz);
      if (nparts) *nparts = gx2;
      LEC_CHECK_LAUNCH("conv_bf16_act_dma_kernel");
      return LEC_OK;
    }
    // two workgroups per CU: 2 stages x 32 KiB (128 x 128), 3 stages x 24 KiB (128 x 64); the epilogue image is smaller than either ring
    if (narrow) hipLaunchKernelGGL((conv_bf16_act_dma_kernel<4, 1, 1, STATS, FOLD, 3>), grid, blk, (size_t)3 * (BM + BN) * 2 * kLnBK, st, src, wgt, dst, gg, part, fz);
    else hipLaunchKernelGGL((conv_bf16_act_dma_kernel<2, 2, 2, STATS, FOLD, 2>), grid, blk, (size_t)2 * (BM + BN) * 2 * kLnBK, st, src, wgt, dst, gg, part, fz);
    if (nparts) *nparts = gx;
    LEC_CHECK_LAUNCH("conv_bf16_act_dma_kernel");
    return LEC_OK;
  }
  if constexpr (FOLD) {
    if (narrow) hipLaunchKernelGGL((conv_bf16_act_kernel<4, 1, 1, false, false, true>), grid, blk, lds, st, src, wgt, dst, gg, part, fz);
    else hipLaunchKernelGGL((conv_bf16_act_kernel<2, 2, 2, false, false, true>), grid, blk, lds, st, src, wgt, dst, gg, part, fz);
  } else {
    if (tapv) {
      if (narrow) hipLaunchKernelGGL((conv_bf16_act_kernel<4, 1, 1, STATS, true, false>), grid, blk, lds, st, src, wgt, dst, gg, part, fz);
      else hipLaunchKernelGGL((conv_bf16_act_kernel<2, 2, 2, STATS, true, false>), grid, blk, lds, st, src, wgt, dst, gg, part, fz);
    } else {
      if (narrow) hipLaunchKernelGGL((conv_bf16_act_kernel<4, 1, 1, STATS, false, false>), grid, blk, lds, st, src, wgt, dst, gg, part, fz);
      else hipLaunchKernelGGL((conv_bf16_act_kernel<2, 2, 2, STATS, false, false>), grid, blk, lds, st, src, wgt, dst, gg, part, fz);
    }
  }
  if (nparts) *nparts = gx;
  LEC_CHECK_LAUNCH("conv_bf16_act_kernel");
  return LEC_OK;
}

static inline void bf16_fwd_geo(ActGeo& g, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad) {
  const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
  g.zfill = 0; g.xcd_per = 0;
  g.Mg = N * Ho * Wo; g.Hm = Ho; g.Wm = Wo; g.Hs = H; g.Ws = W; g.Cs = Cin; g.lgCs = ilog2_exact(Cin); g.sst = stride;
  g.oh0 = -pad; g.ow0 = -pad; g.sg = 1; g.na = R; g.nb = S; g.r0 = 0; g.rstep = 1; g.s0 = 0; g.sstep = 1; g.S = S; g.RS = R * S;
  g.Cd = Cout; g.Cin = Cin; g.Hd = Ho; g.Wd = Wo; g.dst_st = 1; g.dph = 0; g.dpw = 0; g.Kg = R * S * Cin;
  g.src_bytes = (uint32_t)((int64_t)N * H * W * Cin * 2); g.wgt_bytes = (uint32_t)((int64_t)Cout * R * S * Cin * 2);
  g.dst_bytes = (uint32_t)((int64_t)g.Mg * Cout * 2);
  g.dWm = make_fastdiv(g.Wm); g.dHm = make_fastdiv(g.Hm); g.dnb = make_fastdiv(g.nb);
}

}  // namespace lec

extern "C" int lec_conv_bf16_supported(int Cin, int Cout, int R, int S, int stride, int pad) {
  using namespace lec;
  return ilog2_exact(Cin) >= 3 && ilog2_exact(Cout) >= 3 && R > 0 && S > 0 && R * S <= 64 && (stride == 1 || stride == 2) && pad >= 0 && pad < R && pad < S
         && (Cin % kBfBK == 0 || Cin < kBfBK);
}

extern "C" int lec_conv_bf16_wt_transpose(const void* w, void* wt, int Cout, int RS, int Cin, lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(w && wt && Cout > 0 && RS > 0 && Cin > 0, "conv_bf16_wt_transpose: bad arguments");
  hipLaunchKernelGGL(wt_transpose_kernel, dim3((Cin + 63) / 64, (Cout + 63) / 64, RS), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)w, (unsigned short*)wt, Cout, RS, Cin);
  LEC_CHECK_LAUNCH("wt_transpose_kernel");
  return LEC_OK;
}

// Every layer of a flat arena at once: base / base_t are the bf16 arena and its transposed twin (same offsets), table DEVICE int32 [n_layers][5] =
// {element offset, Cout, RS, Cin, first tile}, tiles of a layer = RS * ceil(Cout / 64) * ceil(Cin / 64), total_tiles their sum.
extern "C" int lec_conv_bf16_wt_transpose_flat(const void* base, void* base_t, const int32_t* table, int n_layers, int total_tiles, lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(base && base_t && table && n_layers > 0 && total_tiles > 0, "conv_bf16_wt_transpose_flat: bad arguments");
  hipLaunchKernelGGL(wt_transpose_flat_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)base, (unsigned short*)base_t,
                     (const int*)table, n_layers);
  LEC_CHECK_LAUNCH("wt_transpose_flat_kernel");
  return LEC_OK;
}

extern "C" int lec_conv_bf16_fwd(const void* x, const void* w, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                                 void* y, float* partials, int64_t partials_bytes, int* n_partials, lec_stream_t stream) {
  using namespace lec;
  if (int rc = conv_bf16_check("conv_bf16_fwd", N, H, W, Cin, Cout, R, S, stride, pad)) return rc;
  LEC_CHECK_ARG(x && w && y, "conv_bf16_fwd: null pointer");
  ActGeo g; bf16_fwd_geo(g, N, H, W, Cin, Cout, R, S, stride, pad);
  if (partials) {
    LEC_CHECK_ARG(n_partials && partials_bytes >= (int64_t)kCfMaxPart * 2 * Cout * (int64_t)sizeof(float), "conv_bf16_fwd: partials buffer too small");
    return launch_bf16_act<true, false>((const unsigned short*)x, (const unsigned short*)w, (unsigned short*)y, g, partials, n_partials, (hipStream_t)stream);
  }
  return launch_bf16_act<false, false>((const unsigned short*)x, (const unsigned short*)w, (unsigned short*)y, g, nullptr, nullptr, (hipStream_t)stream);
}

// The ResNet stem (7x7 / stride 2 / pad 3, 64 output channels) on a 3-channel image stored with 8 channels per pixel: channels 0..3 of x and w enter the
// product (the caller keeps channel 3 zero), channels 4..7 are never read.  Image widths 64, 128, 224 (conv_bf16_stem_kernel's instances); H even.
extern "C" int lec_conv_bf16_stem_supported(int H, int W) {
  return lec::tuning().bf_stem && H > 0 && H % 2 == 0 && (W == 224 || W == 128 || W == 64);
}

extern "C" int lec_conv_bf16_stem_fwd(const void* x, const void* w, int N, int H, int W, void* y, float* partials, int64_t partials_bytes, int* n_partials,
                                      lec_stream_t stream) {
  using namespace lec;
  if (int rc = conv_bf16_check("conv_bf16_stem_fwd", N, H, W, 8, 64, 7, 7, 2, 3)) return rc;
  LEC_CHECK_ARG(x && w && y, "conv_bf16_stem_fwd: null pointer");
  LEC_CHECK_ARG(lec_conv_bf16_stem_supported(H, W), "conv_bf16_stem_fwd: image %d x %d is not one of the stem kernel's sizes (even height; width 64, 128 or 224)", H, W);
  LEC_CHECK_ARG(!partials || (n_partials && partials_bytes >= (int64_t)kCfMaxPart * 2 * 64 * (int64_t)sizeof(float)), "conv_bf16_stem_fwd: partials buffer too small");
  return launch_bf16_stem((const unsigned short*)x, (const unsigned short*)w, (unsigned short*)y, N, H, W, partials, partials_bytes, n_partials, (hipStream_t)stream);
}

// Data gradient from TRANSPOSED weights wt [Cin][R*S][Cout] (lec_conv_bf16_wt_transpose).  The fold arguments (all or none; stride 1 only):
// see BfFuse -- the result is then g = mask * (dx + dres) and `partials` receives n_partials rows of [2][Cin].
extern "C" int lec_conv_bf16_dgrad(const void* dy, const void* wt, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                                   void* dx, const void* dres, const void* xbn, const uint8_t* mask, const float* mean, const float* invstd,
                                   float* partials, int64_t partials_bytes, int* n_partials, lec_stream_t stream) {
  using namespace lec;
  if (int rc = conv_bf16_check("conv_bf16_dgrad", N, H, W, Cin, Cout, R, S, stride, pad)) return rc;
  LEC_CHECK_ARG(dy && wt && dx, "conv_bf16_dgrad: null pointer");
  LEC_CHECK_ARG(Cout % kBfBK == 0, "conv_bf16_dgrad: Cout must be a multiple of %d", kBfBK);
  const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
  const bool fold = xbn || mean || invstd || partials;
  LEC_CHECK_ARG(!fold || (stride == 1 && xbn && mean && invstd && partials && n_partials), "conv_bf16_dgrad: the fold needs a stride-1 layer and xbn, mean, invstd, partials, n_partials");
  LEC_CHECK_ARG(!fold || partials_bytes >= (int64_t)kCfMaxPart * 2 * Cin * (int64_t)sizeof(float), "conv_bf16_dgrad: partials buffer too small");
  const unsigned short* src = (const unsigned short*)dy; const unsigned short* wg = (const unsigned short*)wt; unsigned short* dst = (unsigned short*)dx;
  hipStream_t st = (hipStream_t)stream;
  const bool one_launch = stride == 2 && R == 1 && S == 1 && pad == 0 && H % 2 == 0 && W % 2 == 0;
  const bool merged = stride == 2 && !one_launch;
  ActGeoSet gs; int ncls = 0;
  for (int ph = 0; ph < (one_launch ? 1 : stride); ++ph) {
    for (int pw = 0; pw < (one_launch ? 1 : stride); ++pw) {
      ActGeo g;
      g.zfill = one_launch ? 1 : 0; g.xcd_per = 0;
      g.Hm = (H - ph + stride - 1) / stride; g.Wm = (W - pw + stride - 1) / stride;
      if (g.Hm <= 0 || g.Wm <= 0) continue;
      g.Mg = N * g.Hm * g.Wm; g.Hs = Ho; g.Ws = Wo; g.Cs = Cout; g.lgCs = ilog2_exact(Cout); g.sst = 1;
      g.r0 = (ph + pad) % stride; g.s0 = (pw + pad) % stride; g.rstep = stride; g.sstep = stride;
      g.na = g.r0 < R ? (R - g.r0 + stride - 1) / stride : 0; g.nb = g.s0 < S ? (S - g.s0 + stride - 1) / stride : 0;
      g.oh0 = (ph + pad - g.r0) / stride; g.ow0 = (pw + pad - g.s0) / stride; g.sg = -1;
      g.S = S; g.RS = R * S; g.Cd = Cin; g.Cin = Cin; g.Hd = H; g.Wd = W; g.dst_st = stride; g.dph = ph; g.dpw = pw;
      g.Kg = g.na * g.nb * Cout;
      g.src_bytes = (uint32_t)((int64_t)N * Ho * Wo * Cout * 2); g.wgt_bytes = (uint32_t)((int64_t)Cout * R * S * Cin * 2);
      g.dst_bytes = (uint32_t)((int64_t)N * H * W * Cin * 2);
      g.dWm = make_fastdiv(g.Wm); g.dHm = make_fastdiv(g.Hm); g.dnb = make_fastdiv(g.nb);
      if (merged) { gs.g[ncls++] = g; continue; }
      if (fold) {
        BfFuse fz{}; fz.dres = (const unsigned short*)dres; fz.xbn = (const unsigned short*)xbn; fz.mask = mask; fz.mean = mean; fz.invstd = invstd;
        fz.mask_bytes = (uint32_t)((int64_t)N * H * W * (Cin / 8));
        return launch_bf16_act<false, true>(src, wg, dst, g, partials, n_partials, st, fz);
      }
      if (int rc = launch_bf16_act<false, false>(src, wg, dst, g, nullptr, nullptr, st)) return rc;
    }
  }
  if (merged && ncls > 0) {
    for (int a = 1; a < ncls; ++a)
      for (int b = a; b > 0 && gs.g[b].Kg > gs.g[b - 1].Kg; --b) { const ActGeo t = gs.g[b]; gs.g[b] = gs.g[b - 1]; gs.g[b - 1] = t; }
    for (int a = ncls; a < 4; ++a) { gs.g[a] = gs.g[0]; gs.g[a].Mg = 0; }
    const bool narrow = Cin <= 64;
    const int BM = 128, BN = narrow ? 64 : 128;
    const int ntiles = (Cin + BN - 1) / BN;
    int gx = 1;
    for (int a = 0; a < ncls; ++a) { LEC_CHECK_ARG(gs.g[a].na * gs.g[a].nb <= 32, "conv_bf16_dgrad: more than 32 taps per class"); const int mt = (gs.g[a].Mg + BM - 1) / BM; if (mt > gx) gx = mt; }
    int cap = (512 / ntiles) & ~7; if (cap < 8) cap = 8;
    if (gx > cap) gx = cap;
    const size_t lds = (size_t)2 * (BM + BN) * kBfLdk * 2;
    if (tuning().bf_dma) {
      if (narrow) hipLaunchKernelGGL((conv_bf16_act_dma_classes_kernel<4, 1, 1, 3>), dim3(gx, ntiles, ncls), dim3(kBfThreads), (size_t)3 * (BM + BN) * 2 * kLnBK, st, src, wg, dst, gs);
      else hipLaunchKernelGGL((conv_bf16_act_dma_classes_kernel<2, 2, 2, 2>), dim3(gx, ntiles, ncls), dim3(kBfThreads), (size_t)2 * (BM + BN) * 2 * kLnBK, st, src, wg, dst, gs);
    }
    else if (narrow) hipLaunchKernelGGL((conv_bf16_act_classes_kernel<4, 1, 1>), dim3(gx, ntiles, ncls), dim3(kBfThreads), lds, st, src, wg, dst, gs);
    else hipLaunchKernelGGL((conv_bf16_act_classes_kernel<2, 2, 2>), dim3(gx, ntiles, ncls), dim3(kBfThreads), lds, st, src, wg, dst, gs);
    LEC_CHECK_LAUNCH("conv_bf16_act_classes_kernel");
  }
  return LEC_OK;
}

// Weight gradient: dw [Cout][R*S][dw_cin] fp32 += (float atomics over the K split).  dw_cin = Cin, or fewer for a stem whose input carries zero
// padding channels (x [N, H, W, 8] with 3 real channels: dw_cin = 3).
extern "C" int lec_conv_bf16_wgrad(const void* dy, const void* x, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                                   float* dw, int dw_cin, lec_stream_t stream) {
  using namespace lec;
  if (int rc = conv_bf16_check("conv_bf16_wgrad", N, H, W, Cin, Cout, R, S, stride, pad)) return rc;
  LEC_CHECK_ARG(dy && x && dw && dw_cin > 0 && dw_cin <= Cin, "conv_bf16_wgrad: bad arguments");
  if (Cin == 8 && dw_cin <= 4 && Cout == 64 && R == 7 && S == 7 && stride == 2 && pad == 3 && tuning().bf_stem && H % 2 == 0 && (W == 224 || W == 128 || W == 64))
    return launch_bf16_stem_wgrad((const unsigned short*)dy, (const unsigned short*)x, dw, N, H, W, dw_cin, (hipStream_t)stream);   // the stem: its own kernel (same sums)
  WgGeo g;
  g.Ho = (H + 2 * pad - R) / stride + 1; g.Wo = (W + 2 * pad - S) / stride + 1; g.Mpix = N * g.Ho * g.Wo;
  g.H = H; g.W = W; g.Cin = Cin; g.lgCin = ilog2_exact(Cin); g.Cout = Cout; g.S = S; g.RS = R * S; g.stride = stride; g.pad = pad;
  g.Ng = R * S * Cin; g.dCin = dw_cin;
  g.dy_bytes = (uint32_t)((int64_t)g.Mpix * Cout * 2); g.x_bytes = (uint32_t)((int64_t)N * H * W * Cin * 2);
  g.dWo = make_fastdiv(g.Wo); g.dHo = make_fastdiv(g.Ho); g.dS = make_fastdiv(S);
  g.HoWo = g.Ho * g.Wo; g.dHW = make_fastdiv(g.HoWo);
  const bool dense = R == 1 && S == 1 && stride == 1 && pad == 0;
  const int BM = Cout % 128 == 0 ? 128 : 64, BN = 128;
  const int tiles = ((Cout + BM - 1) / BM) * ((g.Ng + BN - 1) / BN);
  const int nchunks = (g.Mpix + kBfWBK - 1) / kBfWBK;
  int split = 1024 / tiles;
  if (split > nchunks) split = nchunks;
  if (split < 1) split = 1;
  g.chunks_per_split = (nchunks + split - 1) / split;
  split = (nchunks + g.chunks_per_split - 1) / g.chunks_per_split;
  g.tiles = tiles; g.split = split;
  const size_t lds = (size_t)2 * kBfWBK * ((BM + 32) + (BN + 32)) * 2;
  const int total = 8 * ((tiles * split + 7) / 8);
  const dim3 grid(total < 2048 ? total : 2048), blk(kBfThreads);
  hipStream_t st = (hipStream_t)stream;
  const unsigned short* a = (const unsigned short*)dy; const unsigned short* b = (const unsigned short*)x;
  if (BM == 128) {
    if (dense) hipLaunchKernelGGL((conv_bf16_wgrad_kernel<2, 2, 2, 2, true>), grid, blk, lds, st, a, b, dw, g);
    else hipLaunchKernelGGL((conv_bf16_wgrad_kernel<2, 2, 2, 2, false>), grid, blk, lds, st, a, b, dw, g);
  } else {
    if (dense) hipLaunchKernelGGL((conv_bf16_wgrad_kernel<2, 2, 1, 2, true>), grid, blk, lds, st, a, b, dw, g);
    else hipLaunchKernelGGL((conv_bf16_wgrad_kernel<2, 2, 1, 2, false>), grid, blk, lds, st, a, b, dw, g);
  }
  LEC_CHECK_LAUNCH("conv_bf16_wgrad_kernel");
  return LEC_OK;
}
