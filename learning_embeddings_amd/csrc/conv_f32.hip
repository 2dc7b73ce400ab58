// Convolutions of the ResNet backbone at the REFERENCE's precision: fp32 activations, fp32 weights, fp32 arithmetic.
//
// The reference runs torchvision's ResNet in plain fp32 (oe_h.py:281-328 FeatCNN18 / :331-378 FeatCNN, no AMP anywhere); on the
// MI355X that is the f32-input matrix instruction v_mfma_f32_32x32x2_f32: exact fp32 (bit for bit a k-ordered fmaf chain), 64
// FLOP/clk/SIMD = 157 TFLOP/s chip-wide, 1/16 of the bf16 rate.  At that rate EVERY convolution of ResNet-50 above layer1's 1x1
// layers is bound by the matrix pipe, not by HBM, and the memory system has slack to spare (a 128 x 128 tile consumes 32 KB of
// operands per 4096 matrix-pipe cycles): one implicit-GEMM kernel family covers every layer, direction and stride:
//
//     forward        Y[m, co]  = sum_{tap, ci} X[pixel(m) + tap, ci] * W[co, tap, ci]          M = N*Ho*Wo, N = Cout, K = R*S*Cin
//     data gradient  dX[m, ci] = sum_{tap, co} dY[src(m, tap), co]  * W[co, tap, ci]           M = N*H*W,   N = Cin,  K = taps*Cout
//     weight grad.   dW[co, (tap, ci)] += sum_m dY[m, co] * X[pixel(m) + tap, ci]               M = Cout, N = R*S*Cin, K = N*Ho*Wo
//
// A workgroup is 4 waves, each owning a 64 x 64 block (2 x 2 MFMA tiles, 64 accumulator registers) of a 128 x 128, 256 x 64 or
// 64 x 256 output tile; K advances in chunks of 32: the chunk's two operand tiles travel global -> registers (16-byte loads, a
// chunk ahead) -> LDS (double-buffered, one barrier per chunk) in the orientation they have in memory -- no transposes on the
// staging path: an operand whose k index is contiguous in memory is read back with ds_read_b128 (4 consecutive k per lane), one
// whose k index is the slow one with 4 x ds_read_b32, and the k ORDER inside an 8-group (k = 8q + 4h + t for lane half h, MFMA
// step t) is the same for both operands, which is all a sum over k needs.  Two workgroups per CU (72 KB of LDS each): one's staging
// runs under the other's MFMAs.
// A strided layer's data gradient runs as stride^2 launches, one per parity class of the input pixels: a class has a fixed subset
// of the taps (3x3 / stride 2: 1, 2, 2 and 4 of the 9), so no MFMA is spent on structural zeros.
// Forward optionally leaves the BatchNorm statistics partials of its output (lec_bn_fwd_prestat_f32's layout): the workgroups
// loop over their m-tiles and keep per-channel sum / sum of squares in registers, so the statistics pass over Y disappears.
//
// Roofline: MFMA (f32).  Algorithmic flops 2*M*N*K per launch; bytes are noise except for layer1's 1x1 layers.
#include "lec_common.h"

namespace lec {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int kCfBK = 32;                 // K chunk (floats)
constexpr int kCfLdk = kCfBK + 4;         // row stride of a k-contiguous LDS tile: 144 B, conflict-free ds_read_b128
constexpr int kCfThreads = 256;
constexpr int kCfMaxPart = 512;           // = kBnMaxBlocks: statistics partial rows

// exact unsigned division by a launch-invariant divisor (Granlund-Montgomery round-up form: exact for every 32-bit dividend)
struct FastDiv { uint32_t mul, sh1, sh2, d; };
static inline FastDiv make_fastdiv(int d_) {
  FastDiv f; const uint32_t d = (uint32_t)(d_ < 1 ? 1 : d_); f.d = d;
  uint32_t l = 0; while ((1ull << l) < d) ++l;
  f.mul = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << l) - d)) / d + 1);
  f.sh1 = l < 1 ? l : 1; f.sh2 = l > 0 ? l - 1 : 0;
  return f;
}
#if defined(__HIPCC__)
__device__ __forceinline__ int fdiv(int n, const FastDiv& f) {
  const uint32_t t = __umulhi(f.mul, (uint32_t)n);
  return (int)((t + (((uint32_t)n - t) >> f.sh1)) >> f.sh2);
}
#endif

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
  FastDiv dWm, dHm, dnb;                   // divisions by Wm, Hm, nb
};

// A predicated 16-byte load WITHOUT a branch: the address is clamped to the tensor's base when the predicate is false and the
// result replaced by zeros afterwards.  (`if (ok) v = *p;` makes hipcc branch around every load and wait for it before the next
// one: eight serialized memory round trips per chunk instead of eight loads in flight.)
// The zeroing happens when the piece is written to LDS (a chunk later), so that nothing uses a loaded register -- and the wave
// waits for none -- until the current chunk's MFMAs have issued.
__device__ __forceinline__ f32x4v ldg4(const float* __restrict__ base, int64_t off, bool ok, unsigned& mask, int bit) {
  mask |= (ok ? 1u : 0u) << bit;
  return *(const f32x4v*)(base + (ok ? off : 0));
}
__device__ __forceinline__ f32x4v keep4(f32x4v v, unsigned mask, int bit) {
  const f32x4v z = {0.f, 0.f, 0.f, 0.f};
  return (mask >> bit) & 1u ? v : z;
}

// MFMAs of one K chunk on the workgroup's LDS tiles.  sA: [BM][kCfLdk] (A_KC) or [32][LDA] (k slow); sB likewise.
template <bool A_KC, bool B_KC, int LDA, int LDB, int TM, int TN>
__device__ __forceinline__ void mma_chunk(const float* __restrict__ sA, const float* __restrict__ sB, int wm0, int wn0, int lane,
                                          f32x16 (&acc)[TM][TN]) {
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int q = 0; q < kCfBK / 8; ++q) {
    float a[TM][4], b[TN][4];
#pragma unroll
    for (int it = 0; it < TM; ++it) {
      if (A_KC) {
        const f32x4v v = *(const f32x4v*)(sA + (wm0 + it * 32 + l31) * kCfLdk + 8 * q + 4 * h);
        a[it][0] = v[0]; a[it][1] = v[1]; a[it][2] = v[2]; a[it][3] = v[3];
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) a[it][t] = sA[(8 * q + 4 * h + t) * LDA + wm0 + it * 32 + l31];
      }
    }
#pragma unroll
    for (int it = 0; it < TN; ++it) {
      if (B_KC) {
        const f32x4v v = *(const f32x4v*)(sB + (wn0 + it * 32 + l31) * kCfLdk + 8 * q + 4 * h);
        b[it][0] = v[0]; b[it][1] = v[1]; b[it][2] = v[2]; b[it][3] = v[3];
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) b[it][t] = sB[(8 * q + 4 * h + t) * LDB + wn0 + it * 32 + l31];
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int it = 0; it < TM; ++it)
#pragma unroll
        for (int jt = 0; jt < TN; ++jt)
          acc[it][jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[it][t], b[jt][t], acc[it][jt], 0, 0, 0);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// forward / data gradient: A = gathered activations (k contiguous), B = weights (k contiguous: forward; k slow: data gradient)
template <bool B_KC, int WM, int WN, int TM, int TN, bool STATS>
__global__ __launch_bounds__(kCfThreads, 2) void conv_f32_act_kernel(const float* __restrict__ src, const float* __restrict__ wgt,
                                                                     float* __restrict__ dst, ActGeo g, float* __restrict__ part) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  static_assert(WM * WN == 4, "four waves per workgroup");
  constexpr int NA = BM * 8 / kCfThreads;                     // 16-byte pieces of the A tile per thread
  constexpr int NB = BN * 8 / kCfThreads;                     // ... of the B tile (same count in either orientation)
  constexpr int SA = BM * kCfLdk;
  constexpr int SB = B_KC ? BN * kCfLdk : kCfBK * BN;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  auto sAb = [&](int b) -> float* { return smem + b * (SA + SB); };
  auto sBb = [&](int b) -> float* { return smem + b * (SA + SB) + SA; };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * 32 * TM, wn0 = (wave % WN) * 32 * TN;
  const int n0 = blockIdx.y * BN;
  const int nchunks = (g.Kg + kCfBK - 1) / kCfBK;
  const int mtiles = (g.Mg + BM - 1) / BM;
  const int kqA = tid & 7;                                    // this thread's 16-byte column of the k-contiguous A tile
  float st_s[TN], st_q[TN];
#pragma unroll
  for (int jt = 0; jt < TN; ++jt) { st_s[jt] = 0.f; st_q[jt] = 0.f; }
  const bool dense_dst = g.dst_st == 1;                        // destination pixel index == m (forward, stride-1 data gradient)

  for (int mt = blockIdx.x; mt < mtiles; mt += gridDim.x) {
    const int m0 = mt * BM;
    // rows of the A tile this thread stages: row = tid / 8 + 32 u
    int pixn[NA], hb[NA], wb[NA];
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int m = m0 + (tid >> 3) + 32 * u;
      if (m < g.Mg) {
        const int t2 = fdiv(m, g.dWm); const int mw = m - t2 * g.Wm; const int n = fdiv(t2, g.dHm); const int mh = t2 - n * g.Hm;
        pixn[u] = n * g.Hs * g.Ws; hb[u] = mh * g.sst + g.oh0; wb[u] = mw * g.sst + g.ow0;
      } else { pixn[u] = -1; hb[u] = 0; wb[u] = 0; }
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int it = 0; it < TM; ++it)
#pragma unroll
      for (int jt = 0; jt < TN; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[it][jt][r] = 0.f;

    f32x4v ra[NA], rb[NB];
    unsigned okm = 0;                                           // predicates of the pieces in flight: bit u (A), bit 16 + u (B)
    auto load_chunk = [&](int ch) {
      okm = 0;
      const int k0 = ch * kCfBK;
      // A: one tap / channel position per 16-byte piece (a chunk lies inside one tap whenever Cs >= 32; the stem has Cs = 4)
      const int kA = k0 + 4 * kqA;
      const int tapA = kA >> g.lgCs, cA = kA & (g.Cs - 1);
      const int ta = fdiv(tapA, g.dnb), tb = tapA - ta * g.nb;
      const int dh = g.sg * ta, dw = g.sg * tb;
      const bool tap_ok = tapA < g.na * g.nb;
#pragma unroll
      for (int u = 0; u < NA; ++u) {
        const int hs = hb[u] + dh, ws = wb[u] + dw;
        const bool ok = tap_ok && pixn[u] >= 0 && (unsigned)hs < (unsigned)g.Hs && (unsigned)ws < (unsigned)g.Ws;
        ra[u] = ldg4(src, (int64_t)(pixn[u] + hs * g.Ws + ws) * g.Cs + cA, ok, okm, u);
      }
      if (B_KC) {
        // B tile [BN rows = output channel co][32 k]: W[co][tap][ci], k contiguous
        const int tw = (g.r0 + g.rstep * ta) * g.S + g.s0 + g.sstep * tb;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const int co = n0 + (tid >> 3) + 32 * u;
          rb[u] = ldg4(wgt, ((int64_t)co * g.RS + tw) * g.Cin + cA, tap_ok && co < g.Cd, okm, 16 + u);
        }
      } else {
        // B tile [32 k rows = channel co of the gradient][BN columns = ci]: W[co][tap][ci], ci contiguous
        constexpr int PR = BN / 4;                              // 16-byte pieces per k row
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const int v_ = tid + kCfThreads * u;
          const int kr = v_ / PR, jq = v_ - kr * PR;
          const int kB = k0 + kr;
          const int tapB = kB >> g.lgCs, cB = kB & (g.Cs - 1);
          const int ta2 = fdiv(tapB, g.dnb), tb2 = tapB - ta2 * g.nb;
          const int tw2 = (g.r0 + g.rstep * ta2) * g.S + g.s0 + g.sstep * tb2;
          const int ci = n0 + 4 * jq;
          rb[u] = ldg4(wgt, ((int64_t)cB * g.RS + tw2) * g.Cin + ci, tapB < g.na * g.nb && ci < g.Cd, okm, 16 + u);
        }
      }
    };
    auto store_chunk = [&](int buf) {
      float* sA = sAb(buf); float* sB = sBb(buf);
#pragma unroll
      for (int u = 0; u < NA; ++u) *(f32x4v*)(sA + ((tid >> 3) + 32 * u) * kCfLdk + 4 * kqA) = keep4(ra[u], okm, u);
      if (B_KC) {
#pragma unroll
        for (int u = 0; u < NB; ++u) *(f32x4v*)(sB + ((tid >> 3) + 32 * u) * kCfLdk + 4 * kqA) = keep4(rb[u], okm, 16 + u);
      } else {
#pragma unroll
        for (int u = 0; u < NB; ++u) *(f32x4v*)(sB + (tid + kCfThreads * u) * 4) = keep4(rb[u], okm, 16 + u);      // [kr][jq] is linear in v
      }
    };

    if (nchunks > 0) {
      load_chunk(0);
      __syncthreads();                                          // the previous m-tile's reads of buffer 0 are done
      store_chunk(0);
      __syncthreads();
      for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunks) load_chunk(ch + 1);               // global -> registers, under this chunk's MFMAs
        mma_chunk<true, B_KC, 0, BN, TM, TN>(sAb(buf), sBb(buf), wm0, wn0, lane, acc);
        if (ch + 1 < nchunks) store_chunk(buf ^ 1);             // (its last readers passed the barrier one chunk ago)
        __syncthreads();
      }
    }

    // epilogue: rows on the registers, 32 consecutive channels on the lanes
    const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int it = 0; it < TM; ++it) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + it * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < g.Mg) {
          int64_t pix = m;
          if (!dense_dst) {
            const int t2 = fdiv(m, g.dWm); const int mw = m - t2 * g.Wm; const int n = fdiv(t2, g.dHm); const int mh = t2 - n * g.Hm;
            pix = ((int64_t)n * g.Hd + mh * g.dst_st + g.dph) * g.Wd + mw * g.dst_st + g.dpw;
          }
#pragma unroll
          for (int jt = 0; jt < TN; ++jt) {
            const int c = n0 + wn0 + jt * 32 + l31;
            if (c < g.Cd) dst[pix * g.Cd + c] = acc[it][jt][r];
          }
        }
      }
    }
    if (STATS) {
#pragma unroll
      for (int jt = 0; jt < TN; ++jt)
#pragma unroll
        for (int it = 0; it < TM; ++it)
#pragma unroll
          for (int r = 0; r < 16; ++r) { const float v = acc[it][jt][r]; st_s[jt] += v; st_q[jt] += v * v; }   // rows past Mg are 0
    }
  }

  if (STATS) {
    // lane halves -> waves of the same column block -> one partial row per workgroup: part[blockIdx.x][2][Cd]
    __syncthreads();
    float* red = smem;                                          // [WM][2 stats][BN]
    const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int jt = 0; jt < TN; ++jt) {
      st_s[jt] += __shfl_xor(st_s[jt], 32, kWave); st_q[jt] += __shfl_xor(st_q[jt], 32, kWave);
      if (h == 0) {
        red[((wave / WN) * 2 + 0) * BN + wn0 + jt * 32 + l31] = st_s[jt];
        red[((wave / WN) * 2 + 1) * BN + wn0 + jt * 32 + l31] = st_q[jt];
      }
    }
    __syncthreads();
    for (int i = tid; i < 2 * BN; i += kCfThreads) {
      const int s = i / BN, c = i - s * BN;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) v += red[(w * 2 + s) * BN + c];
      if (n0 + c < g.Cd) part[((int64_t)blockIdx.x * 2 + s) * g.Cd + n0 + c] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradient: A[co][k = pixel] = dY (k slow), B[k = pixel][(tap, ci)] = X gathered (k slow); split over K, float atomics
struct WgGeo {
  int Mpix;                                // N * Ho * Wo
  int Ho, Wo, H, W, Cin, lgCin, Cout;
  int S, RS, stride, pad;
  int Ng;                                  // RS * Cin
  int chunks_per_split;
  FastDiv dWo, dHo, dS;
};

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(kCfThreads, 2) void conv_f32_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                       float* __restrict__ dw, WgGeo g) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  static_assert(WM * WN == 4, "four waves per workgroup");
  constexpr int NA = BM / 32, NB = BN / 32;                   // 16-byte pieces per thread (32 k rows x BM/4 or BN/4 pieces)
  constexpr int SA = kCfBK * BM, SB = kCfBK * BN;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  auto sAb = [&](int b) -> float* { return smem + b * (SA + SB); };
  auto sBb = [&](int b) -> float* { return smem + b * (SA + SB) + SA; };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * 32 * TM, wn0 = (wave % WN) * 32 * TN;
  const int ntn = (g.Ng + BN - 1) / BN;
  const int tm = blockIdx.x / ntn, tn = blockIdx.x - tm * ntn;
  const int co0 = tm * BM, j0 = tn * BN;
  const int nchunks_all = (g.Mpix + kCfBK - 1) / kCfBK;
  const int ch_lo = blockIdx.y * g.chunks_per_split;
  const int ch_hi = min(nchunks_all, ch_lo + g.chunks_per_split);

  // this thread's pieces: A: v = tid + 256 u -> (k row, co piece); B: (k row, j piece)
  constexpr int PA = BM / 4, PB = BN / 4;
  int krA[NA], cqA[NA], krB[NB], ciB[NB], drB[NB], dsB[NB]; bool okB[NB];
#pragma unroll
  for (int u = 0; u < NA; ++u) { const int v = tid + kCfThreads * u; krA[u] = v / PA; cqA[u] = v - krA[u] * PA; }
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    const int v = tid + kCfThreads * u; krB[u] = v / PB; const int jq = v - krB[u] * PB;
    const int j = j0 + 4 * jq;
    const int tap = j >> g.lgCin; ciB[u] = j & (g.Cin - 1);
    const int r = fdiv(tap, g.dS); drB[u] = r - g.pad; dsB[u] = tap - r * g.S - g.pad; okB[u] = j < g.Ng;
  }
  f32x16 acc[TM][TN];
#pragma unroll
  for (int it = 0; it < TM; ++it)
#pragma unroll
    for (int jt = 0; jt < TN; ++jt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[it][jt][r] = 0.f;
  const bool dense = g.RS == 1 && g.stride == 1 && g.pad == 0;   // 1x1 stride 1: the source pixel of m is m

  f32x4v ra[NA], rb[NB];
  unsigned okm = 0;
  auto load_chunk = [&](int ch) {
    okm = 0;
    const int mbase = ch * kCfBK;
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int m = mbase + krA[u]; const int co = co0 + 4 * cqA[u];
      ra[u] = ldg4(dy, (int64_t)m * g.Cout + co, m < g.Mpix && co < g.Cout, okm, u);
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int m = mbase + krB[u];
      bool ok = m < g.Mpix && okB[u];
      int64_t pix = m;
      if (!dense) {
        const int mm = ok ? m : 0;
        const int t2 = fdiv(mm, g.dWo); const int wo = mm - t2 * g.Wo; const int n = fdiv(t2, g.dHo); const int ho = t2 - n * g.Ho;
        const int hs = ho * g.stride + drB[u], ws = wo * g.stride + dsB[u];
        ok = ok && (unsigned)hs < (unsigned)g.H && (unsigned)ws < (unsigned)g.W;
        pix = (int64_t)(n * g.H + hs) * g.W + ws;
      }
      rb[u] = ldg4(x, pix * g.Cin + ciB[u], ok, okm, 16 + u);
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int u = 0; u < NA; ++u) *(f32x4v*)(sAb(buf) + (tid + kCfThreads * u) * 4) = keep4(ra[u], okm, u);
#pragma unroll
    for (int u = 0; u < NB; ++u) *(f32x4v*)(sBb(buf) + (tid + kCfThreads * u) * 4) = keep4(rb[u], okm, 16 + u);
  };
  if (ch_lo < ch_hi) {
    load_chunk(ch_lo);
    store_chunk(0);
    __syncthreads();
    for (int ch = ch_lo; ch < ch_hi; ++ch) {
      const int buf = (ch - ch_lo) & 1;
      if (ch + 1 < ch_hi) load_chunk(ch + 1);
      mma_chunk<false, false, BM, BN, TM, TN>(sAb(buf), sBb(buf), wm0, wn0, lane, acc);
      if (ch + 1 < ch_hi) store_chunk(buf ^ 1);
      __syncthreads();
    }
    const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int it = 0; it < TM; ++it)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm0 + it * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (co < g.Cout) {
#pragma unroll
          for (int jt = 0; jt < TN; ++jt) {
            const int j = j0 + wn0 + jt * 32 + l31;
            if (j < g.Ng) atomicAdd(dw + (int64_t)co * g.Ng + j, acc[it][jt][r]);
          }
        }
      }
  }
}

static int ilog2_exact(int v) {
  int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1;
}

template <bool B_KC, bool STATS>
static int launch_act(const float* src, const float* wgt, float* dst, const ActGeo& g, float* part, int* nparts, hipStream_t st) {
  // column tile: 128 wide unless the layer has 64 output channels
  const bool narrow = g.Cd <= 64;
  const int BM = 128, BN = narrow ? 64 : 128;
  const int mtiles = (g.Mg + BM - 1) / BM, ntiles = (g.Cd + BN - 1) / BN;
  int gx = mtiles;
  if (STATS) { const int cap = kCfMaxPart; if (gx > cap) gx = cap; }
  else { const int cap = 2048 / (ntiles > 8 ? 8 : ntiles); if (gx > cap) gx = cap; }
  if (gx < 1) gx = 1;
  const size_t lds = (size_t)2 * (BM * kCfLdk + (B_KC ? BN * kCfLdk : kCfBK * BN)) * 4;      // 74 / 55 KB: two workgroups per CU
  if (narrow) hipLaunchKernelGGL((conv_f32_act_kernel<B_KC, 2, 2, 2, 1, STATS>), dim3(gx, ntiles), dim3(kCfThreads), lds, st, src, wgt, dst, g, part);
  else hipLaunchKernelGGL((conv_f32_act_kernel<B_KC, 2, 2, 2, 2, STATS>), dim3(gx, ntiles), dim3(kCfThreads), lds, st, src, wgt, dst, g, part);
  if (nparts) *nparts = gx;
  LEC_CHECK_LAUNCH("conv_f32_act_kernel");
  return LEC_OK;
}

static int conv_check(const char* who, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad) {
  LEC_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && R > 0 && S > 0 && (stride == 1 || stride == 2) && pad >= 0 && pad < R && pad < S,
                "%s: bad geometry N=%d H=%d W=%d Cin=%d Cout=%d R=%d S=%d stride=%d pad=%d", who, N, H, W, Cin, Cout, R, S, stride, pad);
  LEC_CHECK_ARG(Cin % 4 == 0 && ilog2_exact(Cin) >= 2, "%s: Cin must be a power of two >= 4 (pad the stem's 3 channels to 4), got %d", who, Cin);
  LEC_CHECK_ARG(Cout % 4 == 0 && ilog2_exact(Cout) >= 2, "%s: Cout must be a power of two >= 4, got %d", who, Cout);
  LEC_CHECK_ARG((H + 2 * pad - R) / stride + 1 > 0 && (W + 2 * pad - S) / stride + 1 > 0, "%s: empty output", who);
  LEC_CHECK_ARG((int64_t)N * H * W < (1ll << 31) / 4, "%s: too many pixels for 32-bit pixel indices", who);
  return LEC_OK;
}

}  // namespace lec

extern "C" int lec_conv_f32_fwd(const float* x, const float* w, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                                float* y, float* partials, int64_t partials_bytes, int* n_partials, lec_stream_t stream) {
  using namespace lec;
  if (int rc = conv_check("conv_f32_fwd", N, H, W, Cin, Cout, R, S, stride, pad)) return rc;
  LEC_CHECK_ARG(x && w && y, "conv_f32_fwd: null pointer");
  const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
  ActGeo g;
  g.Mg = N * Ho * Wo; g.Hm = Ho; g.Wm = Wo; g.Hs = H; g.Ws = W; g.Cs = Cin; g.lgCs = ilog2_exact(Cin); g.sst = stride;
  g.oh0 = -pad; g.ow0 = -pad; g.sg = 1; g.na = R; g.nb = S; g.r0 = 0; g.rstep = 1; g.s0 = 0; g.sstep = 1; g.S = S; g.RS = R * S;
  g.Cd = Cout; g.Cin = Cin; g.Hd = Ho; g.Wd = Wo; g.dst_st = 1; g.dph = 0; g.dpw = 0; g.Kg = R * S * Cin;
  g.dWm = make_fastdiv(g.Wm); g.dHm = make_fastdiv(g.Hm); g.dnb = make_fastdiv(g.nb);
  if (partials) {
    LEC_CHECK_ARG(n_partials && partials_bytes >= (int64_t)kCfMaxPart * 2 * Cout * (int64_t)sizeof(float), "conv_f32_fwd: partials buffer too small");
    return launch_act<true, true>(x, w, y, g, partials, n_partials, (hipStream_t)stream);
  }
  return launch_act<true, false>(x, w, y, g, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int lec_conv_f32_dgrad(const float* dy, const float* w, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                                  float* dx, lec_stream_t stream) {
  using namespace lec;
  if (int rc = conv_check("conv_f32_dgrad", N, H, W, Cin, Cout, R, S, stride, pad)) return rc;
  LEC_CHECK_ARG(dy && w && dx, "conv_f32_dgrad: null pointer");
  const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
  for (int ph = 0; ph < stride; ++ph) {
    for (int pw = 0; pw < stride; ++pw) {
      ActGeo g;
      g.Hm = (H - ph + stride - 1) / stride; g.Wm = (W - pw + stride - 1) / stride;
      if (g.Hm <= 0 || g.Wm <= 0) continue;
      g.Mg = N * g.Hm * g.Wm; g.Hs = Ho; g.Ws = Wo; g.Cs = Cout; g.lgCs = ilog2_exact(Cout); g.sst = 1;
      g.r0 = (ph + pad) % stride; g.s0 = (pw + pad) % stride; g.rstep = stride; g.sstep = stride;
      g.na = g.r0 < R ? (R - g.r0 + stride - 1) / stride : 0; g.nb = g.s0 < S ? (S - g.s0 + stride - 1) / stride : 0;
      g.oh0 = (ph + pad - g.r0) / stride; g.ow0 = (pw + pad - g.s0) / stride; g.sg = -1;
      g.S = S; g.RS = R * S; g.Cd = Cin; g.Cin = Cin; g.Hd = H; g.Wd = W; g.dst_st = stride; g.dph = ph; g.dpw = pw;
      g.Kg = g.na * g.nb * Cout;
      g.dWm = make_fastdiv(g.Wm); g.dHm = make_fastdiv(g.Hm); g.dnb = make_fastdiv(g.nb);   // (Kg = 0: the loop is empty, zeros are stored)
      if (int rc = launch_act<false, false>(dy, w, dx, g, nullptr, nullptr, (hipStream_t)stream)) return rc;
    }
  }
  return LEC_OK;
}

extern "C" int lec_conv_f32_wgrad(const float* dy, const float* x, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                                  float* dw, lec_stream_t stream) {
  using namespace lec;
  if (int rc = conv_check("conv_f32_wgrad", N, H, W, Cin, Cout, R, S, stride, pad)) return rc;
  LEC_CHECK_ARG(dy && x && dw, "conv_f32_wgrad: null pointer");
  WgGeo g;
  g.Ho = (H + 2 * pad - R) / stride + 1; g.Wo = (W + 2 * pad - S) / stride + 1; g.Mpix = N * g.Ho * g.Wo;
  g.H = H; g.W = W; g.Cin = Cin; g.lgCin = ilog2_exact(Cin); g.Cout = Cout; g.S = S; g.RS = R * S; g.stride = stride; g.pad = pad;
  g.Ng = R * S * Cin;
  g.dWo = make_fastdiv(g.Wo); g.dHo = make_fastdiv(g.Ho); g.dS = make_fastdiv(S);
  const bool narrow = Cout <= 64;
  const int BM = narrow ? 64 : 128, BN = 128;
  const int tiles = ((Cout + BM - 1) / BM) * ((g.Ng + BN - 1) / BN);
  const int nchunks = (g.Mpix + kCfBK - 1) / kCfBK;
  int split = (1024 + tiles - 1) / tiles;                      // ~4 workgroups per CU in total
  if (split > nchunks) split = nchunks;
  if (split < 1) split = 1;
  g.chunks_per_split = (nchunks + split - 1) / split;
  split = (nchunks + g.chunks_per_split - 1) / g.chunks_per_split;
  const size_t lds = (size_t)2 * kCfBK * (BM + BN) * 4;
  if (narrow) hipLaunchKernelGGL((conv_f32_wgrad_kernel<2, 2, 1, 2>), dim3(tiles, split), dim3(kCfThreads), lds, (hipStream_t)stream, dy, x, dw, g);
  else hipLaunchKernelGGL((conv_f32_wgrad_kernel<2, 2, 2, 2>), dim3(tiles, split), dim3(kCfThreads), lds, (hipStream_t)stream, dy, x, dw, g);
  LEC_CHECK_LAUNCH("conv_f32_wgrad_kernel");
  return LEC_OK;
}
