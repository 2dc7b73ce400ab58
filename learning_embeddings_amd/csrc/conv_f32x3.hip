// fp32 convolutions on the bf16 matrix pipe: every fp32 operand is cut into three bf16 pieces and every fp32 product into six
// bf16 products that the matrix cores compute EXACTLY, accumulated in fp32.
//
// Why.  The reference computes its ResNet in plain fp32 (oe_h.py:281-328 / :331-378).  On gfx950 the f32-input MFMA
// (conv_f32.hip) runs on the vector ALUs at the fp32 vector rate, 157 TFLOP/s; the matrix cores proper take 16-bit inputs at
// 2.5 PFLOP/s.  An fp32 number has 24 significand bits = three bf16 significands (8 bits each, same exponent range):
//     x = h + m + l,   h = trunc_bf16(x),  m = trunc_bf16(x - h),  l = x - h - m      (exact: every step is exact in fp32)
// and  a * b = ah*bh + (ah*bm + am*bh) + (ah*bl + am*bm + al*bh) + O(2^-24 |a b|).
// A bf16 x bf16 product has 16 significant bits, so each of the six products is exact in the MFMA's fp32 accumulation; the three
// dropped products (am*bl, al*bm, al*bl) are below 2^-23 of the product, the size of ONE fp32 rounding of it.  The result carries
// the error of an fp32 dot product (tests/test_fp32_gpu.py measures it against fp64 next to the native-fp32 kernel's), at 6 bf16
// MFMAs per 16 k instead of 8 f32 MFMAs of twice the duration: 2.67x the matrix throughput at equal clocks.
//
// Layout.  The same implicit GEMMs, geometry structs and launch decomposition as conv_f32.hip.  Differences:
//   * a workgroup is 4 CONSUMER waves (accumulators; per 16-k step 12 ds_read_b128 of the next chunk's fragments and 24 MFMAs, nothing
//     else) and 4 PRODUCER waves (fetch four chunks ahead, split, store), one of each per SIMD: the bf16 MFMA, unlike the f32 one, executes
//     beside the producers' vector instructions.  One workgroup per CU walks its tiles; chunks number through them without a break;
//   * weights arrive PRE-SPLIT and TILE-MAJOR (lec_conv_f32x3_split_weights, once per optimizer step and layer):
//     [column tile][k chunk][h | m | l][rows][16 k] bf16 in the forward and in the data-gradient order, so the B operand of a step is one
//     contiguous 6 / 12 KB block read as full cache lines and copied to LDS untouched;
//   * activations / output gradients are loaded as fp32 (16-byte buffer loads of whole 128-byte lines, hardware range check = padding
//     zeros), split in registers -- 11 vector instructions per pair of elements (4 v_and, 4 v_sub, 3 v_perm) -- and stored as three planes;
//   * LDS: planes of [rows][16 k] bf16, 32-byte rows without padding, the two 16-byte halves of a row swapped on rows with bit 3 set
//     (conflict-free for the consumers' reads and the producers' writes: x3_half); a stage = two chunks, two stages (98 KB for
//     128 x 128), ONE workgroup barrier per stage, placed before the last four MFMAs of the consumers' step; the barrier waits for LDS
//     traffic only, so the producers' global loads stay in flight across it;
//   * no branch around a load: the producers' stream runs past the last chunk (dead loads fall out of range), which keeps the compiler's
//     vmcnt bookkeeping exact -- with a guard it waits for vmcnt(0) at every step;
//   * the weight gradient transposes while it splits: a producer wave owns one octet of a chunk's 16 pixels and one operand, lane L the
//     tile's channels L and L + 64; pixel pairs are packed per channel, so the LDS image is [channel][16 pixels] for both operands;
//     work items (tile, K split) are dealt to the 8 XCDs in contiguous runs so that tiles sharing a K range share an L2.
// Roofline: MFMA (bf16 dense, 2.5 PFLOP/s) on 6x the algorithmic flops.
#include <type_traits>
#include "conv_geo.h"
#include "tuning.h"

namespace lec {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

constexpr int kX3BK = 16;                 // K chunk = one v_mfma_f32_32x32x16_bf16 step
constexpr int kX3Row = 32;                // bytes of an LDS row of one plane (16 bf16, no padding: see x3_half)
constexpr int kX3Threads = 512;           // 4 consumer + 4 producer waves
constexpr int kX3ActProd = 256;            // producer threads of the activation-gather kernel (two producer waves per SIMD were tried: no gain)
constexpr int kX3ActThreads = 256 + kX3ActProd;
constexpr int kX3Prio = 1;                // s_setprio of the consumer waves (the second-dispatched half of a workgroup loses the VALU arbitration otherwise)
constexpr int kX3KQ = kX3BK / 4;          // 16-byte fp32 pieces per row of an activation tile

// Pre-split weights are stored TILE-MAJOR: [n tile][k chunk][plane h|m|l][BN rows][16 k] bf16, so that the B operand of one step is one
// contiguous block of 3 * BN * 32 bytes (full cache lines, read once) in exactly the order of its LDS image.  BN = x3_bn(columns).
struct X3Wgt { int KC; };                 // k chunks per n tile
static inline int x3_bn(int cols) { return cols <= 64 ? 64 : 128; }

// (x0, x1) -> the packed h | m | l pieces of both: x = h + m + l exactly (every subtraction is exact in fp32).  11 vector instructions
// per pair.  TRUNCATED pieces (v_and_b32 to cut, v_perm_b32 to pack): the products the kernels drop (m*l, l*m, l*l) are below
// 2^-23 of the product.  (Pieces rounded to nearest -- v_cvt_pk_bf16_f32 + shifts: dropped products below 2^-25 and of either sign -- made the
// kernels 12-20 % slower for no measurable change of the error against fp64; removed.)
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  const unsigned a0 = __float_as_uint(x0), a1 = __float_as_uint(x1);
  const float q0 = x0 - __uint_as_float(a0 & 0xffff0000u), q1 = x1 - __uint_as_float(a1 & 0xffff0000u);
  const unsigned b0 = __float_as_uint(q0), b1 = __float_as_uint(q1);
  const float s0 = q0 - __uint_as_float(b0 & 0xffff0000u), s1 = q1 - __uint_as_float(b1 & 0xffff0000u);
  h = __builtin_amdgcn_perm(a1, a0, 0x07060302u); m = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
  l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}

// x[0..3] -> three planes of 4 bf16 (2 dwords each)
__device__ __forceinline__ void split4(const f32x4v x, u32x2v& h, u32x2v& m, u32x2v& l) {
  const float x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];     // (scalar copies: a bit cast applied to a vector ELEMENT reads element 0)
  unsigned a, b, c;
  split_pair(x0, x1, a, b, c); h[0] = a; m[0] = b; l[0] = c;
  split_pair(x2, x3, a, b, c); h[1] = a; m[1] = b; l[1] = c;
}

// Six products: m l, l m (<= 2^-24 of the product each, truncated pieces) and l l (2^-32) are dropped.  (Eight -- every dropped term below 2^-32,
// far under one fp32 rounding -- gives the same error against fp64 in every digit: tests/test_fp32_gpu.py's bound is the fp32 accumulation's.)
constexpr int kX3NP = 6;
static_assert(kX3NP == 6 || kX3NP == 8, "six or eight products");
// the products of one 32 x 32 x 16 step, small terms first
__device__ __forceinline__ f32x16 mma6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 c) {
  if (kX3NP == 8) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[1], c, 0, 0, 0);
  }
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
  return c;
}

// LDS image of one operand plane: [rows][16 k] bf16 = 32-byte rows, NO padding; the two 16-byte halves of a row are swapped on
// rows with bit 3 set.  Reads (ds_read_b128, 16 consecutive rows per quarter-wave at the same half) then touch every bank once,
// and so do the writes (8 consecutive rows per 256 bytes).
__device__ __forceinline__ int x3_half(int row, int half) { return (half ^ ((row >> 3) & 1)) << 4; }

// Workgroup barrier that orders LDS traffic only: s_waitcnt lgkmcnt(0) (vmcnt / expcnt left alone: the producers' global loads stay in
// flight), through the builtins so that the compiler's own wait-count bookkeeping sees the counter drained.
__device__ __forceinline__ void x3_barrier() {
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_s_barrier();
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
}

template <int BM, int BN, int TM, int TN>
__device__ __forceinline__ void x3_read_frags(const char* __restrict__ sA, const char* __restrict__ sB, int wm0, int wn0, int lane,
                                              bf16x8 (&a)[TM][3], bf16x8 (&b)[TN][3]) {
  const int l31 = lane & 31, hs = x3_half(l31, lane >> 5);
#pragma unroll
  for (int it = 0; it < TM; ++it)
#pragma unroll
    for (int p = 0; p < 3; ++p) a[it][p] = *(const bf16x8*)(sA + (p * BM + wm0 + it * 32 + l31) * kX3Row + hs);
#pragma unroll
  for (int jt = 0; jt < TN; ++jt)
#pragma unroll
    for (int p = 0; p < 3; ++p) b[jt][p] = *(const bf16x8*)(sB + (p * BN + wn0 + jt * 32 + l31) * kX3Row + hs);
}

// The consumers' loop (shared by the activation-gather kernel and the weight gradient).  An LDS stage holds TWO chunks and the workgroup
// synchronises once per stage (every second step): chunks 2k and 2k + 1 are published by the barrier that ends step 2k + 1.
//   step t:   producers: split chunk t -> stage (t >> 1) & 1, half t & 1; barrier after odd t
//             consumers: fragments of chunk t - 2 <- LDS (published one barrier ago), MFMAs of chunk t - 3 (fragments read one step ago)
// After the last chunk of a tile (every `nchunks` chunks) tile_done() stores / resets `acc`.
template <int BM, int BN, int TM, int TN, class TileDone>
__device__ __forceinline__ void x3_consumer_loop(const char* __restrict__ smem, int Q, int T, int nchunks, int wm0, int wn0, int lane,
                                                 f32x16 (&acc)[TM][TN], TileDone&& tile_done) {
  constexpr int SA = 3 * BM * kX3Row, SBUF = 3 * (BM + BN) * kX3Row;
  bf16x8 fa[2][TM][3], fb[2][TN][3];
  int mm_ch = 0;                                                // chunks of the current tile already accumulated
  auto tile_end = [&]() __attribute__((always_inline)) {
    if (++mm_ch == nchunks) { tile_done(); mm_ch = 0; }
  };
  // LDS image of the chunk read at a step with t & 3 == J: chunk t - 2 -> stage ((t - 2) >> 1) & 1, half (t - 2) & 1
  auto chunk_base = [&](auto j_c) __attribute__((always_inline)) {
    constexpr int J = decltype(j_c)::value;
    return smem + ((((J + 2) >> 1) & 1) * 2 + (J & 1)) * SBUF;
  };
  // prologue / tail steps: every part conditional, barrier at the end of odd steps
  auto slow_step = [&](int t, auto j_c) __attribute__((always_inline)) {
    constexpr int J = decltype(j_c)::value;                     // = t & 3; fragments of chunk t - 2 go to set J & 1, chunk t - 3's are in set (J & 1) ^ 1
    if (t >= 2 && t <= Q + 1) {
      const char* sA = chunk_base(j_c);
      x3_read_frags<BM, BN, TM, TN>(sA, sA + SA, wm0, wn0, lane, fa[J & 1], fb[J & 1]);
    }
    if (t >= 3 && t <= Q + 2) {
#pragma unroll
      for (int it = 0; it < TM; ++it)
#pragma unroll
        for (int jt = 0; jt < TN; ++jt)
          acc[it][jt] = mma6(fa[(J & 1) ^ 1][it], fb[(J & 1) ^ 1][jt], acc[it][jt]);
      tile_end();
    }
    if (J & 1) x3_barrier();
  };
  // steady-state step (3 <= t <= Q + 1): the fragment reads of the next chunk are spread between the first MFMAs; on odd steps the
  // barrier sits before the last TM * TN MFMAs, which keep the matrix pipe busy while the workgroup synchronises
  auto fast_step = [&](auto j_c) __attribute__((always_inline)) {
    constexpr int J = decltype(j_c)::value;
    constexpr int RS = J & 1, MS = RS ^ 1;                      // register sets: read into RS, multiply from MS
    const char* sA = chunk_base(j_c);
    x3_read_frags<BM, BN, TM, TN>(sA, sA + SA, wm0, wn0, lane, fa[RS], fb[RS]);
    constexpr int PA[8] = {1, 2, 2, 0, 1, 1, 0, 0}, PB[8] = {2, 1, 0, 2, 1, 0, 1, 0};   // small terms first (as mma6); six products: from index 2
    constexpr int P0 = 8 - kX3NP;
#pragma unroll
    for (int p = P0; p < 7; ++p)
#pragma unroll
      for (int it = 0; it < TM; ++it)
#pragma unroll
        for (int jt = 0; jt < TN; ++jt)
          acc[it][jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[MS][it][PA[p]], fb[MS][jt][PB[p]], acc[it][jt], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 3 * (TM + TN); ++i) {                    // one LDS read, one MFMA, ...
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, (kX3NP - 1) * TM * TN - 3 * (TM + TN), 0);
    if (J & 1) {
      __builtin_amdgcn_sched_barrier(0);
      x3_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int it = 0; it < TM; ++it)
#pragma unroll
      for (int jt = 0; jt < TN; ++jt)
        acc[it][jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[MS][it][PA[7]], fb[MS][jt][PB[7]], acc[it][jt], 0, 0, 0);
    tile_end();
  };
  using J0 = std::integral_constant<int, 0>; using J1 = std::integral_constant<int, 1>;
  using J2 = std::integral_constant<int, 2>; using J3 = std::integral_constant<int, 3>;
  slow_step(0, J0{}); slow_step(1, J1{}); slow_step(2, J2{}); slow_step(3, J3{});
  int t = 4;
  for (; t + 3 <= Q + 1; t += 4) { fast_step(J0{}); fast_step(J1{}); fast_step(J2{}); fast_step(J3{}); }
  for (; t < T; t += 4) { slow_step(t, J0{}); slow_step(t + 1, J1{}); slow_step(t + 2, J2{}); slow_step(t + 3, J3{}); }
}

// ---------------------------------------------------------------------------------------------------------------
// forward / data gradient.  A = gathered fp32 activations (split on the way to LDS), B = pre-split bf16 weights, k contiguous.
// A workgroup is 8 waves, one per role and SIMD: waves 0-3 (consumers) own the accumulators -- per step 12 ds_read_b128 of the NEXT
// chunk's fragments and the 24 MFMAs of the current one, nothing else; waves 4-7 (producers) fetch (two chunks ahead), split and
// store.  The producers' vector instructions execute beside the consumers' MFMAs on the same SIMD, which is what the bf16 matrix
// pipe (unlike the f32 one) allows; one barrier per step couples the two.
//   step t:   producers: split chunk t -> LDS stage t & 1, issue the loads of chunk t + 2
//             consumers: fragments of chunk t - 1 <- LDS stage (t - 1) & 1;  MFMAs of chunk t - 2 (fragments read one step ago)
// Chunks number through the workgroup's m-tiles without a break, so the producers run ahead into the next tile while the
// consumers store the finished one.
template <int WM, int WN, int TM, int TN, bool STATS, bool TAPV>
__global__ __launch_bounds__(kX3ActThreads) void conv_f32x3_act_kernel(const float* __restrict__ src, const uint16_t* __restrict__ wpl,
                                                                    float* __restrict__ dst, ActGeo g, X3Wgt wg, float* __restrict__ part) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  static_assert(WM * WN == 4, "four consumer waves per workgroup");
  constexpr int RPP = kX3ActProd / kX3KQ;                      // rows staged per pass of the producers
  constexpr int NA = BM / RPP;                               // fp32 pieces of the A tile per producer thread
  constexpr int NBP = 6 * BN;                                  // 16-byte bf16 pieces of the B tile: 3 planes x BN rows x 2 halves
  constexpr int NB = (NBP + kX3ActProd - 1) / kX3ActProd;
  constexpr int SA = 3 * BM * kX3Row, SBUF = 3 * (BM + BN) * kX3Row;   // bytes
  extern __shared__ __attribute__((aligned(16))) char smem_x3[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= 4;
  const int n0 = blockIdx.y * BN;
  const int nchunks = (g.Kg + kX3BK - 1) / kX3BK;
  const int mtiles = (g.Mg + BM - 1) / BM;
  const int ntl = (int)blockIdx.x < mtiles ? (mtiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;   // m-tiles of this workgroup
  const int Q = ntl * nchunks;                                 // chunks of this workgroup
  const int T = (Q + 3 + 3) & ~3;                              // steps (the consumers trail the producers by three), a multiple of the unroll factor 4

  if (producer) {
    if (Q == 0) return;
    __builtin_amdgcn_s_setprio(kX3Prio);  // the second-dispatched half of a workgroup loses the VALU arbitration otherwise
    const int ptid = tid - kCfThreads;
    const int kqA = ptid & (kX3KQ - 1), rowA = ptid / kX3KQ;
    const rsrc_t rs_src = make_rsrc(src, g.src_bytes), rs_wgt = make_rsrc(wpl, g.wgt_bytes);
    // B pieces of this thread: 16-byte piece v of a chunk's contiguous block [plane][row][half] -> the same place of the LDS image
    unsigned vB[NB], ldsB[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int v = ptid + kX3ActProd * u;
      const int pl = v / (2 * BN), rem = v - pl * 2 * BN, row = rem >> 1, half = rem & 1;
      vB[u] = v < NBP ? (unsigned)v * 16u : kOob;
      ldsB[u] = (unsigned)(SA + (pl * BN + row) * kX3Row + x3_half(row, half));
    }
    const unsigned wtile = (unsigned)blockIdx.y * (unsigned)wg.KC;        // first chunk block of this column tile
    const unsigned ldsA = (unsigned)(rowA * kX3Row + x3_half(rowA, kqA >> 1) + 8 * (kqA & 1));   // (rows rowA + 64 u share bit 3)
    const int ntaps = g.na * g.nb;
    // A stream: the chunk pair to fetch next and its tile's row geometry.  Activations are fetched 32 channels (one 128-byte line per
    // pixel) at a time -- the two 16-channel chunks of a pair -- so that no line is requested from L2 twice.
    int a_ch = 0, a_mt = blockIdx.x, cur_tap = -1;
    int b_ch = 0;                                               // B stream (does not depend on the m-tile)
    int rowoff[NA]; unsigned tapmask[NA]; int hb[NA], wb[NA], pixn[NA];
    unsigned cur[NA];
    f32x4v ra[2][NA][2]; u32x4v rb[4][NB];
    auto tile_setup = [&](int mt) __attribute__((always_inline)) {
      const int m0 = mt * BM;
#pragma unroll
      for (int u = 0; u < NA; ++u) {
        const int m = m0 + rowA + RPP * u;
        const bool live = m < g.Mg;
        const int mm = live ? m : 0;
        const int t2 = fdiv(mm, g.dWm); const int mw = mm - t2 * g.Wm; const int n = fdiv(t2, g.dHm); const int mh = t2 - n * g.Hm;
        hb[u] = mh * g.sst + g.oh0; wb[u] = mw * g.sst + g.ow0; pixn[u] = live ? n * g.Hs * g.Ws : -1;
        rowoff[u] = (((n * g.Hs + hb[u]) * g.Ws + wb[u]) << g.lgCs) * 4 + 16 * kqA;
        unsigned msk = 0;
        if (!TAPV) {
          for (int t = 0; t < ntaps; ++t) {
            const int ta = fdiv(t, g.dnb), tb = t - ta * g.nb;
            const int hs = hb[u] + g.sg * ta, ws = wb[u] + g.sg * tb;
            msk |= ((unsigned)hs < (unsigned)g.Hs && (unsigned)ws < (unsigned)g.Ws ? 1u : 0u) << t;
          }
        }
        tapmask[u] = live ? msk : 0u;
      }
      cur_tap = -1;
    };
    // chunks a_c, a_c + 1 (the same tap: source channels are a multiple of 32) -> ra[DSET][u][0 | 1]
    // The stream's position is kept INCREMENTALLY (tap, byte offset of the channel block inside the pixel): a producer wave's
    // instruction stream is the critical path of the kernel (one wave per SIMD issues ~1 instruction per 9 cycles next to the
    // MFMAs), and the from-scratch decode (shifts, masks, an exact division per chunk) was a third of it.
    int a_tap = 0; unsigned a_c0b = 0;
    const unsigned a_row_bytes = (unsigned)g.Cs * 4u;
    auto issueA = [&](auto dset_c) __attribute__((always_inline)) {
      constexpr int DSET = decltype(dset_c)::value;
      if (a_ch == 0) { tile_setup(a_mt); a_tap = 0; a_c0b = 0; } // (past the last tile every row is dead: the loads fall out of range)
      if (a_tap != cur_tap) {
        cur_tap = a_tap;
        const int ta = fdiv(a_tap, g.dnb), tb = a_tap - ta * g.nb;
        const int toff = (((g.sg * ta) * g.Ws + g.sg * tb) << g.lgCs) * 4;
        const unsigned tapbit = 1u << a_tap;
#pragma unroll
        for (int u = 0; u < NA; ++u) cur[u] = (tapmask[u] & tapbit) ? (unsigned)(rowoff[u] + toff) : kOob;
      }
#pragma unroll
      for (int u = 0; u < NA; ++u) {
        ra[DSET][u][0] = bload4(rs_src, cur[u] + a_c0b);        // (a poisoned offset stays out of range: a_c0b < 2^14)
        ra[DSET][u][1] = bload4(rs_src, cur[u] + a_c0b + 64u);
      }
      a_c0b += 128u;
      if (a_c0b == a_row_bytes) { a_c0b = 0; ++a_tap; }
      a_ch += 2;
      if (a_ch >= nchunks) { a_ch = 0; a_mt += gridDim.x; }
    };
    // the stem (4 source channels, 4 taps per chunk): one chunk, one tap per 16-byte piece -> ra[DSET][u][HALF]
    auto issueA_tapv = [&](auto dset_c, auto half_c) __attribute__((always_inline)) {
      constexpr int DSET = decltype(dset_c)::value, HALF = decltype(half_c)::value;
      if (a_ch == 0) tile_setup(a_mt);
      const int kA = a_ch * kX3BK + 4 * kqA;
      const int tapA = kA >> g.lgCs, cA = kA & (g.Cs - 1);
      const int ta2 = fdiv(tapA, g.dnb), tb2 = tapA - ta2 * g.nb;
      const int dh = g.sg * ta2, dw = g.sg * tb2;
      const bool tap_ok = tapA < ntaps;
#pragma unroll
      for (int u = 0; u < NA; ++u) {
        const int hs = hb[u] + dh, ws = wb[u] + dw;
        const bool ok = tap_ok && pixn[u] >= 0 && (unsigned)hs < (unsigned)g.Hs && (unsigned)ws < (unsigned)g.Ws;
        ra[DSET][u][HALF] = bload4(rs_src, ok ? (unsigned)(((pixn[u] + hs * g.Ws + ws) << g.lgCs) + cA) * 4u : kOob);
      }
      if (++a_ch == nchunks) { a_ch = 0; a_mt += gridDim.x; }
    };
    constexpr unsigned kBlk = (unsigned)(3 * BN * kX3BK * 2);    // bytes of one chunk's B block
    const bool b_linear = g.rstep == 1 && g.sstep == 1;          // the launch walks every tap in order (forward, stride-1 data gradient): block index = chunk index
    unsigned b_wsc = wtile * kBlk;
    auto issueB = [&](auto set_c) __attribute__((always_inline)) {
      constexpr int SET = decltype(set_c)::value;
      unsigned wsc = b_wsc;
      if (!b_linear) {                                          // a parity class of a strided data gradient: its taps are a subset, decode
        const int k0 = b_ch * kX3BK;
        const int tap = k0 >> g.lgCs, c0 = k0 & (g.Cs - 1);
        const int ta = fdiv(tap, g.dnb), tb = tap - ta * g.nb;
        const int tw = (g.r0 + g.rstep * ta) * g.S + g.s0 + g.sstep * tb;
        wsc = (wtile + ((unsigned)(tw * g.Cs + c0) >> 4)) * kBlk;
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) rb[SET][u] = __builtin_bit_cast(u32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs_wgt, (int)(vB[u] + wsc), 0, 0));
      b_wsc += kBlk;
      if (++b_ch == nchunks) { b_ch = 0; b_wsc = wtile * kBlk; }  // (past the last chunk: harmless re-reads)
    };
    // chunk t (= i mod 4): registers -> LDS stage t & 1, then refill.  No branch around a load or a store: the steps past the last
    // chunk run too (dead loads, a stage nobody reads), so that the compiler's vmcnt bookkeeping stays exact and the loads really
    // stay four chunks ahead -- with a guard it falls back to vmcnt(0) at every step.
    auto step = [&](auto i_c) __attribute__((always_inline)) {
      constexpr int I = decltype(i_c)::value;
      {
        char* base = smem_x3 + I * SBUF;                         // stage I >> 1, half I & 1 (a stage = two chunks)
#pragma unroll
        for (int u = 0; u < NA; ++u) {
          u32x2v h, m, l;
          split4(ra[I >> 1][u][I & 1], h, m, l);
          char* p = base + ldsA + u * RPP * kX3Row;
          *(u32x2v*)(p) = h; *(u32x2v*)(p + BM * kX3Row) = m; *(u32x2v*)(p + 2 * BM * kX3Row) = l;
        }
#pragma unroll
        for (int u = 0; u < NB; ++u)
          if (NBP % kX3ActProd == 0 || ptid + kX3ActProd * u < NBP) *(u32x4v*)(base + ldsB[u]) = rb[I][u];
        issueB(i_c);
        if (TAPV) issueA_tapv(std::integral_constant<int, (I >> 1)>{}, std::integral_constant<int, (I & 1)>{});
        else if (I & 1) issueA(std::integral_constant<int, (I >> 1)>{});
      }
      if (I & 1) x3_barrier();                                  // a stage is complete
    };
    {                                    // the first four chunks, in the order the loop issues them
      using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
            using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
      issueB(I0{}); if (TAPV) issueA_tapv(I0{}, I0{});
      issueB(I1{}); if (TAPV) issueA_tapv(I0{}, I1{}); else issueA(I0{});
      issueB(I2{}); if (TAPV) issueA_tapv(I1{}, I0{});
      issueB(I3{}); if (TAPV) issueA_tapv(I1{}, I1{}); else issueA(I1{});
    }
    for (int t = 0; t < T; t += 4) {
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{});
    }
  } else {
    // ---- consumers
    const int wm0 = (wave / WN) * 32 * TM, wn0 = (wave % WN) * 32 * TN;
    const rsrc_t rs_dst = make_rsrc(dst, g.dst_bytes);
    const bool dense_dst = g.dst_st == 1;
    const int l31 = lane & 31, h = lane >> 5;
    float st_s[TN], st_q[TN];
#pragma unroll
    for (int jt = 0; jt < TN; ++jt) { st_s[jt] = 0.f; st_q[jt] = 0.f; }
    unsigned coff[TN];
#pragma unroll
    for (int jt = 0; jt < TN; ++jt) { const int c = n0 + wn0 + jt * 32 + l31; coff[jt] = c < g.Cd ? (unsigned)c * 4u : kOob; }
    f32x16 acc[TM][TN];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int it = 0; it < TM; ++it)
#pragma unroll
        for (int jt = 0; jt < TN; ++jt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[it][jt][r] = 0.f;
    };
    auto epilogue = [&](int mt) __attribute__((always_inline)) {
      const int m0 = mt * BM;
      // Full tiles of a dense destination (the common case): one vector add per accumulator row, the column block as the store's immediate
      // offset, no per-row branch.  (The general path below spent 64 uniform branches and ~6 vector instructions per row on every tile.)
      if (dense_dst && m0 + BM <= g.Mg && n0 + BN <= g.Cd) {
        const unsigned rowbytes = (unsigned)g.Cd * 4u;
        const unsigned base = (unsigned)(m0 + wm0 + 4 * h) * rowbytes + (unsigned)(n0 + wn0 + l31) * 4u;
#pragma unroll
        for (int it = 0; it < TM; ++it)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const unsigned off = base + (unsigned)(it * 32 + (r & 3) + 8 * (r >> 2)) * rowbytes;
#pragma unroll
            for (int jt = 0; jt < TN; ++jt) bstore1(acc[it][jt][r], rs_dst, off + (unsigned)jt * 128u);
          }
      } else {
#pragma unroll
        for (int it = 0; it < TM; ++it) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm0 + it * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            unsigned poff;
            if (dense_dst) {
              poff = m < g.Mg ? (unsigned)(m * g.Cd) * 4u : kOob;
            } else {
              const int mm = m < g.Mg ? m : 0;
              const int t2 = fdiv(mm, g.dWm); const int mw = mm - t2 * g.Wm; const int n = fdiv(t2, g.dHm); const int mh = t2 - n * g.Hm;
              const int pix = (n * g.Hd + mh * g.dst_st + g.dph) * g.Wd + mw * g.dst_st + g.dpw;
              poff = m < g.Mg ? (unsigned)(pix * g.Cd) * 4u : kOob;
            }
            // valid offsets are < 2^31 and coff < 2^14: the sum of two valid parts cannot reach the kOob bit, and a poisoned part keeps it
#pragma unroll
            for (int jt = 0; jt < TN; ++jt) bstore1(acc[it][jt][r], rs_dst, (poff + coff[jt]) | ((poff | coff[jt]) & kOob));
          }
        }
      }
      if (STATS) {
#pragma unroll
        for (int jt = 0; jt < TN; ++jt)
#pragma unroll
          for (int it = 0; it < TM; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float v = acc[it][jt][r]; st_s[jt] += v; st_q[jt] += v * v; }
      }
    };
    zero_acc();
    if (nchunks == 0) {                                         // a parity class without taps: its pixels of dx are zeros
      for (int mt = blockIdx.x; mt < mtiles; mt += gridDim.x) epilogue(mt);
      return;
    }
    if (Q > 0) {
      int mm_mt = blockIdx.x;                                   // the m-tile the accumulators belong to
      auto done = [&]() __attribute__((always_inline)) { epilogue(mm_mt); zero_acc(); mm_mt += gridDim.x; };
      x3_consumer_loop<BM, BN, TM, TN>(smem_x3, Q, T, nchunks, wm0, wn0, lane, acc, done);
    }
    if (STATS) {
      // lane halves -> waves of the same column block -> one partial row per workgroup: part[blockIdx.x][2][Cd]
      float* red = (float*)smem_x3;                             // [WM][2 stats][BN]; the ring is drained (last barrier passed)
#pragma unroll
      for (int jt = 0; jt < TN; ++jt) {
        st_s[jt] += __shfl_xor(st_s[jt], 32, kWave); st_q[jt] += __shfl_xor(st_q[jt], 32, kWave);
        if (h == 0) {
          red[((wave / WN) * 2 + 0) * BN + wn0 + jt * 32 + l31] = st_s[jt];
          red[((wave / WN) * 2 + 1) * BN + wn0 + jt * 32 + l31] = st_q[jt];
        }
      }
    }
  }
  if (STATS) {
    __syncthreads();
    const float* red = (const float*)smem_x3;
    for (int i = tid; i < 2 * BN; i += kX3ActThreads) {
      const int s = i / BN, c = i - s * BN;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) v += red[(w * 2 + s) * BN + c];
      if (n0 + c < g.Cd) part[((int64_t)blockIdx.x * 2 + s) * g.Cd + n0 + c] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------------
// weight gradient: dW[co][(tap, ci)] += sum over pixels m of dY[m][co] * X[pixel(m) + tap][ci].   M = Cout, N = R*S*Cin, K = pixels.
// Both operands are fp32 activations whose k index (the pixel) is the SLOW one in memory, and the bf16 MFMA wants 8 consecutive k per
// lane: the producers transpose while they split.  A chunk is 16 pixels; a producer wave owns ONE octet of them and one operand
// (waves 4, 5: dY octets 0, 1; waves 6, 7: X octets 0, 1); its lane L owns the tile's channels L and L + 64 and loads them for
// the octet's 8 pixels (16 dword loads, each a contiguous 256 bytes across the wave), splits the 16 values and packs PIXEL pairs:
// 8 k of one channel = 16 bytes per plane = the LDS image [channel][16 k] the consumers read with ds_read_b128.
// The source pixel of an X row is wave-uniform: lane l of the wave decodes pixel (group base + l) once per four chunks (64 pixels,
// vector ALU, ~20 instructions) and the row offsets are read back with v_readlane; the tile's tap is uniform because a 128-column
// tile lies inside one tap (Cin >= 128).  Work item = (128 x 128 tile of dW, K split); float atomics into dW, as in conv_f32.hip.
struct WgX3Geo {
  int Mpix, Ho, Wo, H, W, Cin, lgCin, Cout, S, stride, pad, Ng;
  int cps;                                 // chunks per work item (a multiple of 4; rows past Mpix are zeros)
  int tiles_n, tiles, items;
  int per;                                 // ceil(items / 8): work items per XCD
  uint32_t dy_bytes, x_bytes;
  FastDiv dWo, dHo, dS;
};

__device__ __forceinline__ float bload1(rsrc_t rsrc, unsigned voff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, 0, 0));
}

// Work item of slot i (i = workgroup id + k * grid).  Workgroup ids go round-robin over the 8 XCDs, each with its own L2, and the
// items are numbered K-split major (split * tiles + tile): XCD x takes the CONTIGUOUS items [x * per, (x + 1) * per), so the tiles
// that share a K range -- the same dY rows for every tap / column tile, the same X rows for every Cout tile -- run side by side
// on one XCD and the second to ninth reader of a chunk hits that XCD's L2 instead of HBM.  -1: no such item.
__device__ __forceinline__ int x3_wg_item(int i, int items, int per) {
  if (i >= 8 * per) return -1;
  const int it = (i & 7) * per + (i >> 3);
  return ((i >> 3) < per && it < items) ? it : -1;
}

template <bool DENSE, bool NARROW>
__global__ __launch_bounds__(kX3Threads) void conv_f32x3_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                      float* __restrict__ dw, WgX3Geo g) {
  constexpr int BM = 128, BN = 128, TM = 2, TN = 2, WN = 2;
  constexpr int SA = 3 * BM * kX3Row, SBUF = 3 * (BM + BN) * kX3Row;
  extern __shared__ __attribute__((aligned(16))) char smem_x3[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int nit = 0;                                                  // items of this workgroup (slots blockIdx.x, + grid, ...)
  for (int i = blockIdx.x; i < 8 * g.per; i += gridDim.x) nit += x3_wg_item(i, g.items, g.per) >= 0 ? 1 : 0;
  const int Q = nit * g.cps;
  const int T = (Q + 3 + 3) & ~3;
  if (Q == 0) return;

  if (wave >= 4) {
    __builtin_amdgcn_s_setprio(kX3Prio);
    const int pw = wave - 4;
    const bool isB = pw >= 2;                                   // (wave-uniform)
    const int oct = pw & 1;
    const rsrc_t rs = isB ? make_rsrc(x, g.x_bytes) : make_rsrc(dy, g.dy_bytes);
    const int C = isB ? g.Cin : g.Cout;                         // channels per pixel of this wave's source tensor
    const unsigned ldsW = (unsigned)((isB ? SA : 0) + lane * kX3Row + x3_half(lane, oct));   // channel row `lane`; row lane + 64: + 64 rows (same bit 3)
    // load stream: item, chunk inside the item
    int ld_slot = blockIdx.x, ld_ch = 0;
    unsigned vcol = 0, vcol1 = 0;                               // byte offsets of this lane's two channels inside a pixel (NARROW: the second may not exist: kOob)
    int dr = 0, ds = 0, dr1 = 0, ds1 = 0;                       // the taps of the lane's two columns, relative to the output pixel (B; NARROW: they may differ)
    bool tap_ok = true, tap1_ok = true;
    int ch0 = 0;                                                // first chunk of the item
    unsigned pixo = 0, pixo1 = 0;                               // B: byte offset of the source pixel of (group base + lane) per tap, kOob if there is none
    constexpr int D = 4;                                        // chunks in flight (register sets)
    float rr[D][16];
    auto item_setup = [&]() __attribute__((always_inline)) {
      int wi = x3_wg_item(ld_slot, g.items, g.per);
      while (wi < 0 && ld_slot < 8 * g.per) { ld_slot += gridDim.x; wi = x3_wg_item(ld_slot, g.items, g.per); }
      if (wi < 0) wi = 0;                                         // (past the last item: dead loads of the steps that drain the ring)
      const int tile = wi % g.tiles, sp = wi / g.tiles;
      const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
      ch0 = sp * g.cps;
      if (isB) {
        const int j0 = tn * BN;
        const int tap = j0 >> g.lgCin, ci0 = j0 & (g.Cin - 1);
        const int r_ = fdiv(tap, g.dS);
        dr = r_ - g.pad; ds = tap - r_ * g.S - g.pad;
        vcol = (unsigned)(ci0 + lane) * 4u; vcol1 = vcol + 256u;
        if (NARROW) {                                           // layers of 64 channels / ragged column tiles: the second column has its own tap and may not exist
          const int j1 = j0 + 64;
          const int tap1 = j1 >> g.lgCin, ci1 = j1 & (g.Cin - 1);
          const int r1 = fdiv(tap1, g.dS);
          dr1 = r1 - g.pad; ds1 = tap1 - r1 * g.S - g.pad;
          tap_ok = j0 < g.Ng; tap1_ok = j1 < g.Ng;              // (Ng = taps * Cin and 64 | Cin: a 64-column half lies inside one tap or past the last)
          vcol1 = tap1_ok ? (unsigned)(ci1 + lane) * 4u : kOob;
        }
      } else {
        vcol = (unsigned)(tm * BM + lane) * 4u; vcol1 = vcol + 256u;
        if (NARROW && tm * BM + 64 >= g.Cout) vcol1 = kOob;      // Cout = 64: rows 64 .. 127 of the tile do not exist
      }
    };
    auto issue = [&](auto i_c) __attribute__((always_inline)) {   // chunk ld_ch of the current item -> rr[I % D]; I == chunk & 3
      constexpr int I = decltype(i_c)::value, SET = I % D;
      if (ld_ch == 0) item_setup();
      const int mb = (ch0 + ld_ch) * kX3BK;                      // first pixel of the chunk
      if (isB && !DENSE && I == 0) {                              // decode the 64 pixels of chunks ld_ch .. ld_ch + 3
        const int m = mb + lane;
        const bool live = m < g.Mpix;
        const int mm = live ? m : 0;
        const int t2 = fdiv(mm, g.dWo); const int wo = mm - t2 * g.Wo; const int n = fdiv(t2, g.dHo); const int ho = t2 - n * g.Ho;
        const int hs = ho * g.stride + dr, ws = wo * g.stride + ds;
        const bool ok = live && tap_ok && (unsigned)hs < (unsigned)g.H && (unsigned)ws < (unsigned)g.W;
        pixo = ok ? (unsigned)(((n * g.H + hs) * g.W + ws) << g.lgCin) * 4u : kOob;
        if (NARROW) {
          const int hs1 = ho * g.stride + dr1, ws1 = wo * g.stride + ds1;
          const bool ok1 = live && tap1_ok && (unsigned)hs1 < (unsigned)g.H && (unsigned)ws1 < (unsigned)g.W;
          pixo1 = ok1 ? (unsigned)(((n * g.H + hs1) * g.W + ws1) << g.lgCin) * 4u : kOob;
        }
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        unsigned srow;
        if (isB && !DENSE) srow = (unsigned)__builtin_amdgcn_readlane((int)pixo, 16 * I + 8 * oct + r);
        else srow = (unsigned)((mb + 8 * oct + r) * C) * 4u;      // (past the tensor: out of range, zeros)
        const unsigned o = vcol + srow;
        rr[SET][2 * r] = bload1(rs, o);
        if (!NARROW) rr[SET][2 * r + 1] = bload1(rs, o + 256u);
        else {
          // (a missing column AND a missing pixel wrap to a small valid offset: finite garbage in a tile row / column the epilogue skips)
          const unsigned srow1 = (isB && !DENSE) ? (unsigned)__builtin_amdgcn_readlane((int)pixo1, 16 * I + 8 * oct + r) : srow;
          rr[SET][2 * r + 1] = bload1(rs, vcol1 + srow1);
        }
      }
      if (++ld_ch == g.cps) { ld_ch = 0; ld_slot += gridDim.x; }
    };
    auto step = [&](auto i_c) __attribute__((always_inline)) {
      constexpr int I = decltype(i_c)::value;
      char* base = smem_x3 + I * SBUF + ldsW;                    // stage I >> 1, half I & 1
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        u32x4v ph, pm, pl;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          unsigned a_, b_, c_;
          split_pair(rr[I % D][2 * (2 * q) + j], rr[I % D][2 * (2 * q + 1) + j], a_, b_, c_);
          ph[q] = a_; pm[q] = b_; pl[q] = c_;
        }
        char* p = base + j * 64 * kX3Row;
        *(u32x4v*)(p) = ph; *(u32x4v*)(p + BM * kX3Row) = pm; *(u32x4v*)(p + 2 * BM * kX3Row) = pl;     // (BM == BN)
      }
      issue(std::integral_constant<int, (I + D) & 3>{});           // refill the set with the chunk D steps ahead
      if (I & 1) x3_barrier();
    };
    issue(std::integral_constant<int, 0>{}); issue(std::integral_constant<int, 1>{});
    if (D == 4) { issue(std::integral_constant<int, 2>{}); issue(std::integral_constant<int, 3>{}); }
    for (int t = 0; t < T; t += 4) {
      step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
    }
  } else {
    const int wm0 = (wave / WN) * 32 * TM, wn0 = (wave % WN) * 32 * TN;
    const int l31 = lane & 31, h = lane >> 5;
    f32x16 acc[TM][TN];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int it = 0; it < TM; ++it)
#pragma unroll
        for (int jt = 0; jt < TN; ++jt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[it][jt][r] = 0.f;
    };
    zero_acc();
    int cur_slot = blockIdx.x;
    x3_consumer_loop<BM, BN, TM, TN>(smem_x3, Q, T, g.cps, wm0, wn0, lane, acc, [&]() __attribute__((always_inline)) {
      int wi = x3_wg_item(cur_slot, g.items, g.per);
      while (wi < 0) { cur_slot += gridDim.x; wi = x3_wg_item(cur_slot, g.items, g.per); }   // (terminates: this is one of the nit items)
      const int tile = wi % g.tiles;
      const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
      float* out = dw + (int64_t)(tm * BM + wm0) * g.Ng + tn * BN + wn0 + l31;
      // (NARROW: Cout and Ng are multiples of 64, so a wave's 64 x 64 block is inside dW or outside it as a whole)
      if (!NARROW || (tm * BM + wm0 < g.Cout && tn * BN + wn0 < g.Ng)) {
#pragma unroll
        for (int it = 0; it < TM; ++it)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = it * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
            for (int jt = 0; jt < TN; ++jt) atomicAdd(out + (int64_t)row * g.Ng + jt * 32, acc[it][jt][r]);
          }
      }
      zero_acc();
      cur_slot += gridDim.x;
    });
  }
}

// weight split: w[Cout][RS][Cin] fp32 -> tile-major bf16 planes (X3Wgt).  One thread per (n, k) of the padded GEMM operand B[n][k]:
//   forward (T = false):       n = co, k = tap * Cin + ci
//   data gradient (T = true):  n = ci, k = tap * Cout + co
template <bool T>
__global__ __launch_bounds__(256) void x3_split_weights_kernel(const float* __restrict__ w, int Cout, int RS, int Cin, int BN, int NT, int KC,
                                                               uint16_t* __restrict__ out) {
  const int64_t total = (int64_t)NT * KC * BN * kX3BK;
  const int ncols = T ? Cin : Cout, kdim = T ? RS * Cout : RS * Cin;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int kk = (int)(e % kX3BK); int64_t q = e / kX3BK;
    const int r = (int)(q % BN); q /= BN;
    const int kc = (int)(q % KC); const int nt = (int)(q / KC);
    const int n = nt * BN + r, k = kc * kX3BK + kk;
    float x = 0.f;
    if (n < ncols && k < kdim) {
      if (T) { const int tap = k / Cout, co = k - tap * Cout; x = w[((int64_t)co * RS + tap) * Cin + n]; }
      else x = w[(int64_t)n * kdim + k];
    }
    unsigned hb, mb, lb;
    split_pair(x, 0.f, hb, mb, lb);
    const int64_t o = ((int64_t)(nt * KC + kc) * 3 * BN + r) * kX3BK + kk;      // plane 0; planes are BN * 16 elements apart
    out[o] = (uint16_t)hb; out[o + (int64_t)BN * kX3BK] = (uint16_t)mb; out[o + 2 * (int64_t)BN * kX3BK] = (uint16_t)lb;
  }
}

static inline void x3_plane_geo(int Cout, int RS, int Cin, bool transposed, int* BN, int* NT, int* KC) {
  const int ncols = transposed ? Cin : Cout, kdim = transposed ? RS * Cout : RS * Cin;
  *BN = x3_bn(ncols); *NT = (ncols + *BN - 1) / *BN; *KC = (kdim + kX3BK - 1) / kX3BK;
}

template <bool STATS>
static int launch_act_x3(const float* src, const uint16_t* wpl, float* dst, const ActGeo& g, const X3Wgt& wg, float* part, int* nparts, hipStream_t st) {
  // tile: 128 x 128, or 256 x 64 for layers of 64 output channels; one workgroup (8 waves) per CU, which walks its m-tiles
  const bool narrow = g.Cd <= 64;
  const int BM = narrow ? 256 : 128, BN = narrow ? 64 : 128;
  const int mtiles = (g.Mg + BM - 1) / BM, ntiles = (g.Cd + BN - 1) / BN;
  const int wgs = tuning().x3_wgs;
  int gx = (wgs + ntiles - 1) / ntiles;
  if (gx > mtiles) gx = mtiles;
  if (STATS && gx > kCfMaxPart) gx = kCfMaxPart;
  if (gx < 1) gx = 1;
  const size_t lds = (size_t)4 * 3 * (BM + BN) * kX3Row;        // 2 stages x 2 chunks: 96 / 120 KB
  const bool tapv = g.Cs < 2 * kX3BK;                           // source channels narrower than a chunk pair (the stem)
  LEC_CHECK_ARG(tapv || g.na * g.nb <= 32, "conv_f32x3: more than 32 taps per launch need the per-piece tap path");
  LEC_CHECK_ARG(!tapv || (g.rstep == 1 && g.sstep == 1 && g.r0 == 0 && g.s0 == 0),
                "conv_f32x3: layers with fewer than 32 source channels are supported in the forward direction only");
  LEC_CHECK_ARG(BN == x3_bn(g.Cd), "conv_f32x3: tile width and weight layout disagree");
  const dim3 grid(gx, ntiles), blk(kX3ActThreads);
  if (tapv) {
    if (narrow) hipLaunchKernelGGL((conv_f32x3_act_kernel<4, 1, 2, 2, STATS, true>), grid, blk, lds, st, src, wpl, dst, g, wg, part);
    else hipLaunchKernelGGL((conv_f32x3_act_kernel<2, 2, 2, 2, STATS, true>), grid, blk, lds, st, src, wpl, dst, g, wg, part);
  } else {
    if (narrow) hipLaunchKernelGGL((conv_f32x3_act_kernel<4, 1, 2, 2, STATS, false>), grid, blk, lds, st, src, wpl, dst, g, wg, part);
    else hipLaunchKernelGGL((conv_f32x3_act_kernel<2, 2, 2, 2, STATS, false>), grid, blk, lds, st, src, wpl, dst, g, wg, part);
  }
  if (nparts) *nparts = gx;
  LEC_CHECK_LAUNCH("conv_f32x3_act_kernel");
  return LEC_OK;
}

}  // namespace lec

extern "C" int64_t lec_conv_f32x3_planes_elems(int Cout, int RS, int Cin, int transposed) {
  int BN, NT, KC; lec::x3_plane_geo(Cout, RS, Cin, transposed != 0, &BN, &NT, &KC);
  return (int64_t)NT * KC * 3 * BN * lec::kX3BK;
}

extern "C" int lec_conv_f32x3_split_weights(const float* w, int Cout, int RS, int Cin, uint16_t* planes_fwd, uint16_t* planes_t, lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(w && (planes_fwd || planes_t) && Cout > 0 && RS > 0 && Cin > 0, "conv_f32x3_split_weights: bad arguments");
  for (int t = 0; t < 2; ++t) {
    uint16_t* out = t ? planes_t : planes_fwd;
    if (!out) continue;
    int BN, NT, KC; x3_plane_geo(Cout, RS, Cin, t != 0, &BN, &NT, &KC);
    const int64_t total = (int64_t)NT * KC * BN * kX3BK;
    int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
    if (t) hipLaunchKernelGGL(x3_split_weights_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, Cout, RS, Cin, BN, NT, KC, out);
    else hipLaunchKernelGGL(x3_split_weights_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, Cout, RS, Cin, BN, NT, KC, out);
  }
  LEC_CHECK_LAUNCH("x3_split_weights_kernel");
  return LEC_OK;
}

extern "C" int lec_conv_f32x3_fwd(const float* x, const uint16_t* w_planes, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                                  float* y, float* partials, int64_t partials_bytes, int* n_partials, lec_stream_t stream) {
  using namespace lec;
  if (int rc = conv_check("conv_f32x3_fwd", N, H, W, Cin, Cout, R, S, stride, pad)) return rc;
  LEC_CHECK_ARG(x && w_planes && y, "conv_f32x3_fwd: null pointer");
  const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
  ActGeo g;
  g.zfill = 0;
  g.Mg = N * Ho * Wo; g.Hm = Ho; g.Wm = Wo; g.Hs = H; g.Ws = W; g.Cs = Cin; g.lgCs = ilog2_exact(Cin); g.sst = stride;
  g.oh0 = -pad; g.ow0 = -pad; g.sg = 1; g.na = R; g.nb = S; g.r0 = 0; g.rstep = 1; g.s0 = 0; g.sstep = 1; g.S = S; g.RS = R * S;
  g.Cd = Cout; g.Cin = Cin; g.Hd = Ho; g.Wd = Wo; g.dst_st = 1; g.dph = 0; g.dpw = 0; g.Kg = R * S * Cin;
  g.src_bytes = (uint32_t)((int64_t)N * H * W * Cin * 4); g.dst_bytes = (uint32_t)((int64_t)g.Mg * Cout * 4);
  X3Wgt wg; int bn_, nt_; x3_plane_geo(Cout, R * S, Cin, false, &bn_, &nt_, &wg.KC);
  g.wgt_bytes = (uint32_t)(lec_conv_f32x3_planes_elems(Cout, R * S, Cin, 0) * 2);
  g.dWm = make_fastdiv(g.Wm); g.dHm = make_fastdiv(g.Hm); g.dnb = make_fastdiv(g.nb);
  if (partials) {
    LEC_CHECK_ARG(n_partials && partials_bytes >= (int64_t)kCfMaxPart * 2 * Cout * (int64_t)sizeof(float), "conv_f32x3_fwd: partials buffer too small");
    return launch_act_x3<true>(x, w_planes, y, g, wg, partials, n_partials, (hipStream_t)stream);
  }
  return launch_act_x3<false>(x, w_planes, y, g, wg, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int lec_conv_f32x3_dgrad(const float* dy, const uint16_t* w_planes_t, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                                    float* dx, lec_stream_t stream) {
  using namespace lec;
  if (int rc = conv_check("conv_f32x3_dgrad", N, H, W, Cin, Cout, R, S, stride, pad)) return rc;
  LEC_CHECK_ARG(dy && w_planes_t && dx, "conv_f32x3_dgrad: null pointer");
  LEC_CHECK_ARG(Cout % (2 * kX3BK) == 0, "conv_f32x3_dgrad: Cout must be a multiple of 32, got %d", Cout);
  const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
  X3Wgt wg; int bn_, nt_; x3_plane_geo(Cout, R * S, Cin, true, &bn_, &nt_, &wg.KC);
  const uint32_t wbytes = (uint32_t)(lec_conv_f32x3_planes_elems(Cout, R * S, Cin, 1) * 2);
  for (int ph = 0; ph < stride; ++ph) {
    for (int pw = 0; pw < stride; ++pw) {
      ActGeo g;
      g.zfill = 0;
      g.Hm = (H - ph + stride - 1) / stride; g.Wm = (W - pw + stride - 1) / stride;
      if (g.Hm <= 0 || g.Wm <= 0) continue;
      g.Mg = N * g.Hm * g.Wm; g.Hs = Ho; g.Ws = Wo; g.Cs = Cout; g.lgCs = ilog2_exact(Cout); g.sst = 1;
      g.r0 = (ph + pad) % stride; g.s0 = (pw + pad) % stride; g.rstep = stride; g.sstep = stride;
      g.na = g.r0 < R ? (R - g.r0 + stride - 1) / stride : 0; g.nb = g.s0 < S ? (S - g.s0 + stride - 1) / stride : 0;
      g.oh0 = (ph + pad - g.r0) / stride; g.ow0 = (pw + pad - g.s0) / stride; g.sg = -1;
      g.S = S; g.RS = R * S; g.Cd = Cin; g.Cin = Cin; g.Hd = H; g.Wd = W; g.dst_st = stride; g.dph = ph; g.dpw = pw;
      g.Kg = g.na * g.nb * Cout;
      g.src_bytes = (uint32_t)((int64_t)N * Ho * Wo * Cout * 4); g.wgt_bytes = wbytes;
      g.dst_bytes = (uint32_t)((int64_t)N * H * W * Cin * 4);
      g.dWm = make_fastdiv(g.Wm); g.dHm = make_fastdiv(g.Hm); g.dnb = make_fastdiv(g.nb);
      if (int rc = launch_act_x3<false>(dy, w_planes_t, dx, g, wg, nullptr, nullptr, (hipStream_t)stream)) return rc;
    }
  }
  return LEC_OK;
}

extern "C" int lec_conv_f32x3_wgrad_supported(int Cin, int Cout, int R, int S) {
  return Cin >= 64 && Cout >= 64 && (Cin & (Cin - 1)) == 0 && (Cout & (Cout - 1)) == 0 && R > 0 && S > 0 && R * S <= 32;
}

extern "C" int lec_conv_f32x3_wgrad(const float* dy, const float* x, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                                    float* dw, lec_stream_t stream) {
  using namespace lec;
  if (int rc = conv_check("conv_f32x3_wgrad", N, H, W, Cin, Cout, R, S, stride, pad)) return rc;
  LEC_CHECK_ARG(dy && x && dw, "conv_f32x3_wgrad: null pointer");
  LEC_CHECK_ARG(lec_conv_f32x3_wgrad_supported(Cin, Cout, R, S), "conv_f32x3_wgrad: needs Cin and Cout to be powers of two >= 64 and at most 32 taps (got %d, %d): use lec_conv_f32_wgrad", Cin, Cout);
  WgX3Geo g;
  g.Ho = (H + 2 * pad - R) / stride + 1; g.Wo = (W + 2 * pad - S) / stride + 1; g.Mpix = N * g.Ho * g.Wo;
  g.H = H; g.W = W; g.Cin = Cin; g.lgCin = ilog2_exact(Cin); g.Cout = Cout; g.S = S; g.stride = stride; g.pad = pad; g.Ng = R * S * Cin;
  g.dy_bytes = (uint32_t)((int64_t)g.Mpix * Cout * 4); g.x_bytes = (uint32_t)((int64_t)N * H * W * Cin * 4);
  g.dWo = make_fastdiv(g.Wo); g.dHo = make_fastdiv(g.Ho); g.dS = make_fastdiv(S);
  const bool dense = R == 1 && S == 1 && stride == 1 && pad == 0;
  g.tiles_n = (g.Ng + 127) / 128; g.tiles = ((Cout + 127) / 128) * g.tiles_n;
  const bool force_narrow = tuning().x3_force_narrow != 0;   // experiments
  const bool narrow = Cin < 128 || Cout < 128 || force_narrow;   // 64-channel layers: half-empty tiles, two taps per column tile
  const int nchunks = (g.Mpix + kX3BK - 1) / kX3BK;
  const int wgs = tuning().x3_wgs;
  // K split: accumulation chains of at most 512 chunks (8192 pixels: the rounding error of a longer fp32 chain shows against fp64),
  // and among the splits up to ~2048 work items the one that fills the last round of workgroups best (fewest atomics on a tie)
  int best = 1; double best_eff = -1.0;
  const int chain = tuning().x3_chain;
  const int smin = (nchunks + chain - 1) / chain;
  for (int sp = smin < 1 ? 1 : smin; sp <= nchunks && (sp == smin || (int64_t)g.tiles * sp <= 2048); ++sp) {
    const int items = g.tiles * sp;
    const double eff = (double)items / (double)(((items + wgs - 1) / wgs) * wgs);
    if (eff > best_eff + 0.02) { best_eff = eff; best = sp; }
  }
  int cps = (nchunks + best - 1) / best; cps = (cps + 3) & ~3; if (cps < 4) cps = 4;
  const int split = (nchunks + cps - 1) / cps;
  g.cps = cps; g.items = g.tiles * split; g.per = (g.items + 7) / 8;
  const int grid = 8 * g.per < wgs ? 8 * g.per : wgs;
  const size_t lds = (size_t)4 * 3 * (128 + 128) * kX3Row;
  hipStream_t st = (hipStream_t)stream;
  if (dense && narrow) hipLaunchKernelGGL((conv_f32x3_wgrad_kernel<true, true>), dim3(grid), dim3(kX3Threads), lds, st, dy, x, dw, g);
  else if (dense) hipLaunchKernelGGL((conv_f32x3_wgrad_kernel<true, false>), dim3(grid), dim3(kX3Threads), lds, st, dy, x, dw, g);
  else if (narrow) hipLaunchKernelGGL((conv_f32x3_wgrad_kernel<false, true>), dim3(grid), dim3(kX3Threads), lds, st, dy, x, dw, g);
  else hipLaunchKernelGGL((conv_f32x3_wgrad_kernel<false, false>), dim3(grid), dim3(kX3Threads), lds, st, dy, x, dw, g);
  LEC_CHECK_LAUNCH("conv_f32x3_wgrad_kernel");
  return LEC_OK;
}
