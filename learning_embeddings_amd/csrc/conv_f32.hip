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
// A workgroup is 4 waves, each owning a 64 x 64 block (2 x 2 MFMA tiles, 64 accumulator registers); K advances in chunks of 32: the
// chunk's two operand tiles travel global -> registers (16-byte buffer loads, a chunk ahead) -> LDS (double-buffered, one barrier
// per chunk) in the orientation they have in memory -- no transposes on the staging path: an operand whose k index is contiguous in
// memory is read back with ds_read_b128 (4 consecutive k per lane), one whose k index is the slow one with 4 x ds_read_b32, and the
// k ORDER inside an 8-group (k = 8q + 4h + t for lane half h, MFMA step t) is the same for both operands, which is all a sum over k
// needs.  Two workgroups per CU.
//
// THE rule this file is written by (measured, profiles/r02_conv_f32_pmc.md): the f32 MFMA runs ON the vector ALUs -- rocprofv3 reads
// SQ_VALU_MFMA_COEXEC_CYCLES = 0 for these kernels, and a second wave per SIMD hides nothing -- so every VALU instruction of any
// resident wave is matrix time lost (the first version spent 2.3 VALU instructions per MFMA on 64-bit address arithmetic, predicate
// selects and branches and reached 62 % of the matrix peak where the same loop with trivial addresses reaches 92 %).  Hence:
//   * operands are fetched with raw buffer loads: a 32-bit byte offset per piece, and the hardware's range check returns the zeros
//     of padding, row tails and K tails (offset 0x80000000 = out of range), so no data is ever masked;
//   * a chunk lies inside one filter tap: tap decode, channel offset and weight offset are wave-uniform (scalar ALU); a piece's
//     offset is a per-row base computed once per m-tile plus that scalar; its validity is one bit of a per-row tap mask;
//   * the weight gradient stages 256-wide rows so that one wave-instruction covers ONE pixel: the pixel decode is scalar too;
//   * no 64-bit vector arithmetic, no branches inside the K loop.
// A strided layer's data gradient runs as stride^2 launches, one per parity class of the input pixels: a class has a fixed subset
// of the taps (3x3 / stride 2: 1, 2, 2 and 4 of the 9), so no MFMA is spent on structural zeros.
// Forward optionally leaves the BatchNorm statistics partials of its output (lec_bn_fwd_prestat_f32's layout): the workgroups
// loop over their m-tiles and keep per-channel sum / sum of squares in registers, so the statistics pass over Y disappears.
//
// Roofline: MFMA (f32).  Algorithmic flops 2*M*N*K per launch; bytes are noise except for layer1's 1x1 layers.
#include "conv_geo.h"
#include "tuning.h"
#include <mutex>
#include <atomic>
#include <map>
#include <utility>

constexpr int kWgXfBlocks = 2;         // workgroups per CU the on-load weight gradient is compiled for (4: 128 registers, a few spills -- measured slower)

namespace lec {


// One dword of a k-slow LDS tile: per-lane base + a compile-time row offset.  Volatile on purpose: it keeps the access a single
// ds_read_b32 with a 16-bit immediate offset; left to itself the compiler pairs two rows into ds_read2_b32, whose 8-bit offsets cannot
// span a row, and spends a VALU add per pair on a new base -- vector instructions the f32 MFMA loop cannot hide.
typedef __attribute__((address_space(3))) const volatile float lds_cvfloat;
__device__ __forceinline__ float lds_ld1(const float* base, int elem_off) {
  return *((lds_cvfloat*)base + elem_off);                    // explicit LDS address space: a volatile GENERIC access would become flat_load
}

// MFMAs of one K chunk on the workgroup's LDS tiles.  sA: [BM][kCfLdk] (A_KC) or [32][LDA] (k slow); sB likewise.
template <bool A_KC, bool B_KC, int LDA, int LDB, int TM, int TN, int BKT = kCfBK>
__device__ __forceinline__ void mma_chunk(const float* __restrict__ sA, const float* __restrict__ sB, int wm0, int wn0, int lane,
                                          f32x16 (&acc)[TM][TN]) {
  const int l31 = lane & 31, h = lane >> 5;
  // (Reading the fragments of group q + 1 under the MFMAs of group q -- two register sets, pinned with a scheduling barrier -- was tried:
  // the convolutions of a step take 123.8 instead of 118.7 ms; the second workgroup per CU already hides the LDS latency.)
#pragma unroll
  for (int q = 0; q < BKT / 8; ++q) {
    float a[TM][4], b[TN][4];
#pragma unroll
    for (int it = 0; it < TM; ++it) {
      if (A_KC) {
        const f32x4v v = *(const f32x4v*)(sA + (wm0 + it * 32 + l31) * kCfLdk + 8 * q + 4 * h);
        a[it][0] = v[0]; a[it][1] = v[1]; a[it][2] = v[2]; a[it][3] = v[3];
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) a[it][t] = lds_ld1(sA + 4 * h * LDA + wm0 + it * 32 + l31, (8 * q + t) * LDA);
      }
    }
#pragma unroll
    for (int it = 0; it < TN; ++it) {
      if (B_KC) {
        const f32x4v v = *(const f32x4v*)(sB + (wn0 + it * 32 + l31) * kCfLdk + 8 * q + 4 * h);
        b[it][0] = v[0]; b[it][1] = v[1]; b[it][2] = v[2]; b[it][3] = v[3];
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) b[it][t] = lds_ld1(sB + 4 * h * LDB + wn0 + it * 32 + l31, (8 * q + t) * LDB);
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
// TAPV: the tap of a 16-byte piece varies inside a chunk (source channels < chunk width: the stem's 4 channels, 8 taps per
// chunk); otherwise a chunk lies inside ONE tap and everything but the per-row validity bit and one add per piece is scalar.
// Fused BatchNorm pieces (round 3).  At fp32 the convolutions are bound by the matrix pipe and leave HBM idle, the BatchNorm passes are
// bound by HBM and leave the matrix pipe idle: bytes moved from a BatchNorm pass into a convolution's loader or epilogue are hidden.
//   XF (A operand on load, 1x1 layers): the kernel is handed g and the BatchNorm input x instead of dy and forms
//       dy = cA[c] * g + cB[c] * x + cD[c]   (pass 2 of the BatchNorm backward BEHIND this layer; c = channel of dy = the k index)
//     per 16-byte piece on its way from registers to LDS (coefficients from lec_bn_bwd_coeffs_f32: three 16-byte loads per chunk).
//   FOLD (epilogue, dense destination): instead of dx the kernel writes g = mask * (dx + dres) -- pass 1 of the backward of the
//     BatchNorm IN FRONT of this layer (whose output this layer consumed; dres = the gradient of that output's other consumer) -- and
//     leaves the per-channel partials sum g, sum g * xhat (xhat from that BatchNorm's input) in lec_bn_bwd_f32's workspace layout.
//   AFF (epilogue, forward of an eval-mode network): the BatchNorm behind the layer is a per-channel affine map of running statistics; the kernel
//     applies it (+ residual, + ReLU) to the accumulators -- the arithmetic of bn_apply_kernel, bit for bit -- and the raw convolution output is
//     never written: inference runs without a single BatchNorm pass.
struct ActFuse {
  const float* xsrc; const float* coef;                       // XF: second source tensor (same geometry as src), coefficients [3][Cs]
  const float* dres; const float* xbn; const unsigned char* mask; const float* mean; const float* invstd;   // FOLD
  uint32_t mask_bytes;
  const float* scale; const float* shift; const float* res; int relu;   // AFF: y = [relu](acc * scale[c] + shift[c] [+ res]) -- an eval-mode BatchNorm in the epilogue
};

template <bool B_KC, int WM, int WN, int TM, int TN, bool STATS, bool TAPV, int FUSE = 0>
__global__ __launch_bounds__(kCfThreads, 2) void conv_f32_act_kernel(const float* __restrict__ src, const float* __restrict__ wgt,
                                                                     float* __restrict__ dst, ActGeo g, float* __restrict__ part,
                                                                     ActFuse fz) {
#include "conv_f32_act_body.inc"
}

// The parity classes of a STRIDED data gradient as ONE launch (round 4): blockIdx.z = class, the classes ordered by their tap count, longest first.
// As four launches of one stream every class ended in a round of workgroup slots of its own (3x3 / stride 2 at 512 images: 784 tiles on 512 slots, four
// times over); as one grid the hardware deals 1-, 2- and 4-tap tiles to whichever slot frees up, longest first, and the launch has one tail.
template <int TN>
__global__ __launch_bounds__(kCfThreads, 2) void conv_f32_act_classes_kernel(const float* __restrict__ src, const float* __restrict__ wgt,
                                                                             float* __restrict__ dst, ActGeoSet gs) {
  constexpr bool B_KC = false, STATS = false, TAPV = false; constexpr int WM = 2, WN = 2, TM = 2, FUSE = 0;
  const ActGeo g = gs.g[blockIdx.z];
  float* const part = nullptr; const ActFuse fz{};
#include "conv_f32_act_body.inc"
}

// ---------------------------------------------------------------------------------------------------------------
// Balanced ("stream-K") form of the forward / stride-1 data gradient, for launches whose tiles do not fill whole rounds of workgroup slots.
//
// Why (tools/exp_tile_quantization.py, round 3): every ResNet-50 layer from layer2 on has 392 x 2^k equal 128 x 128 tiles at the bench batch --
// 1 568 / 784 / 392, i.e. 3.06 / 1.53 / 0.77 rounds of the 512 resident workgroups -- and the kernel above runs them at 100 - 105 TFLOP/s; the SAME
// kernel on a pixel count that fills whole rounds (1 024, 1 536 tiles) runs at 131 - 137.  The K loop was never the problem (94 % busy in steady state,
// profiles/r03_conv_f32_pmc.md): a quarter of the time went to a last round that is nearly empty.
//
// How: the launch's work is the sequence of all (tile, k chunk) iterations, tile-major; workgroup w of G takes the contiguous share
// [I w / G, I (w + 1) / G) -- equal work for everyone, whole tiles where the share covers them, at most one unfinished tile at each end.  A tile
// whose chunks are spread over several workgroups is put together by the LAST of them to arrive (ticket counter per tile; nobody waits, so nothing
// can deadlock): every contributor writes its partial accumulators to its own slot of a scratch buffer, and the finalizer sums the slots of ALL
// contributors in workgroup order (its own included, read back from memory) -- a fixed order, so the output bits do not depend on who arrives
// last -- and runs the epilogue (store / fold, statistics).  Statistics partials are written per TILE (row = m-tile index), again independent of
// who computed them.  The slabs travel write-through: sc1 stores, each wave's vmcnt drain, barrier, one lane's agent-scope ticket; the finalizer reads
// them with sc1 loads only (CDNA4 guide 6 G16, form R1 -- an agent-scope release fence per slab writes back the XCD's whole L2 and cost 100 - 160 us
// per launch on the 1x1 layers).  Scratch: 2 slots of BM x BN floats per workgroup + one counter per tile, registered per stream
// (lec_conv_f32_scratch); counters are zero between launches (the finalizer re-arms its tile's counter).
// (sc0 sc1 on the slab accesses and the acquire below were first built while
// chasing a 5.5e-3 gradient discrepancy that turned out to be ONE ReLU decision flipped by the different summation order; all three gave the same bits as the defaults.)
constexpr int kSkLoadAux = 16, kSkStoreAux = 16;      // sc1 on the slab accesses
// The finalizer takes an agent-scope ACQUIRE in front of its loads (round 4): it only invalidates this CU's L1 and runs in at most one workgroup per
// tile; with it the hand-off no longer leans on the sc1 loads alone (the relaxed ticket orders nothing by itself).
struct SkArgs { float* slots; unsigned int* counters; int ntn; int tiles; };

template <bool B_KC, int TM, int TN, bool STATS, int FUSE>
__global__ __launch_bounds__(kCfThreads, 2) void conv_f32_act_sk_kernel(const float* __restrict__ src, const float* __restrict__ wgt,
                                                                        float* __restrict__ dst, ActGeo g, float* __restrict__ part,
                                                                        ActFuse fz, SkArgs sk) {
  constexpr int WM = 2, WN = 2;
  constexpr bool FOLD = (FUSE & 2) != 0, AFF = (FUSE & 4) != 0;
  static_assert((FUSE & 1) == 0 && !(FOLD && STATS) && (!AFF || FUSE == 4), "stream-K serves the plain, the fold and the affine epilogues");
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  constexpr int NA = BM * kCfKQ / kCfThreads, NB = BN * kCfKQ / kCfThreads;
  constexpr int SA = BM * kCfLdk;
  constexpr int SB = B_KC ? BN * kCfLdk : kCfBK * BN;
  constexpr int PR = BN / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * 32 * TM, wn0 = (wave % WN) * 32 * TN;
  const int nchunks = (g.Kg + kCfBK - 1) / kCfBK;
  const int kqA = tid & (kCfKQ - 1), rowA = tid / kCfKQ;
  const rsrc_t rs_src = make_rsrc(src, g.src_bytes), rs_wgt = make_rsrc(wgt, g.wgt_bytes), rs_dst = make_rsrc(dst, g.dst_bytes);
  const int rsc = g.RS * g.Cin;
  const unsigned ldsA = (unsigned)((rowA * kCfLdk + 4 * kqA) * 4);
  const unsigned ldsB = (unsigned)((SA + (B_KC ? rowA * kCfLdk + 4 * kqA : tid * 4)) * 4);
  const int ntaps = g.na * g.nb;
  const int l31 = lane & 31, h = lane >> 5;
  const long long I = (long long)sk.tiles * nchunks, G = gridDim.x;
  const rsrc_t rs_slots = make_rsrc(sk.slots, (uint32_t)(gridDim.x * 2u * (unsigned)(BM * BN * 4)));
  auto wg_start = [&](long long w) { return I * w / G; };
  long long it0 = wg_start(blockIdx.x);
  const long long it1 = wg_start(blockIdx.x + 1);

  while (it0 < it1) {
    const int tile = (int)(it0 / nchunks);
    const int c0 = (int)(it0 - (long long)tile * nchunks);
    const int c1 = (int)((it1 - it0) < (long long)(nchunks - c0) ? c0 + (it1 - it0) : nchunks);
    it0 += c1 - c0;
    const int mt = tile / sk.ntn, nt = tile - mt * sk.ntn;      // column tiles of the same rows are neighbours in the sequence
    const int m0 = mt * BM, n0 = nt * BN;
    unsigned wB[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      if (B_KC) { const int co = n0 + rowA + kCfRP * u; wB[u] = co < g.Cd ? (unsigned)(co * rsc + 4 * kqA) * 4u : kOob; }
      else { const int v_ = tid + kCfThreads * u; const int kr = v_ / PR, jq = v_ - kr * PR; const int ci = n0 + 4 * jq;
             wB[u] = ci < g.Cd ? (unsigned)(kr * rsc + ci) * 4u : kOob; }
    }
    int rowoff[NA]; unsigned tapmask[NA];
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int m = m0 + rowA + kCfRP * u;
      const bool live = m < g.Mg;
      const int mm = live ? m : 0;
      const int t2 = fdiv(mm, g.dWm); const int mw = mm - t2 * g.Wm; const int n = fdiv(t2, g.dHm); const int mh = t2 - n * g.Hm;
      const int hb = mh * g.sst + g.oh0, wb = mw * g.sst + g.ow0;
      rowoff[u] = (((n * g.Hs + hb) * g.Ws + wb) << g.lgCs) * 4 + 16 * kqA;
      unsigned msk = 0;
      for (int t = 0; t < ntaps; ++t) {
        const int ta = fdiv(t, g.dnb), tb = t - ta * g.nb;
        const int hs = hb + g.sg * ta, ws = wb + g.sg * tb;
        msk |= ((unsigned)hs < (unsigned)g.Hs && (unsigned)ws < (unsigned)g.Ws ? 1u : 0u) << t;
      }
      tapmask[u] = live ? msk : 0u;
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int it = 0; it < TM; ++it)
#pragma unroll
      for (int jt = 0; jt < TN; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[it][jt][r] = 0.f;
    f32x4v ra[NA], rb[NB];
    unsigned cur[NA];
    int cur_tap = -1;
    auto load_chunk = [&](int ch) {
      const int k0 = ch * kCfBK;
      const int tap = k0 >> g.lgCs, c0k = k0 & (g.Cs - 1);
      const int ta = fdiv(tap, g.dnb), tb = tap - ta * g.nb;
      if (tap != cur_tap) {
        cur_tap = tap;
        const int toff = (((g.sg * ta) * g.Ws + g.sg * tb) << g.lgCs) * 4;
        const unsigned tapbit = tap < 32 ? 1u << tap : 0u;
#pragma unroll
        for (int u = 0; u < NA; ++u) cur[u] = (tapmask[u] & tapbit) ? (unsigned)(rowoff[u] + toff) : kOob;
      }
      const int tw = (g.r0 + g.rstep * ta) * g.S + g.s0 + g.sstep * tb;
      const unsigned wsc = (unsigned)(B_KC ? tw * g.Cin + c0k : c0k * rsc + tw * g.Cin) * 4u;
      const unsigned c0b = (unsigned)c0k * 4u;
#pragma unroll
      for (int u = 0; u < NA; ++u) ra[u] = bload4(rs_src, cur[u] + c0b);
#pragma unroll
      for (int u = 0; u < NB; ++u) rb[u] = bload4(rs_wgt, wB[u] + wsc);
    };
    auto store_chunk = [&](int buf) {
      char* base = (char*)smem + buf * (SA + SB) * 4;
#pragma unroll
      for (int u = 0; u < NA; ++u) *(f32x4v*)(base + ldsA + u * kCfRP * kCfLdk * 4) = ra[u];
#pragma unroll
      for (int u = 0; u < NB; ++u) *(f32x4v*)(base + ldsB + u * (B_KC ? kCfRP * kCfLdk * 4 : kCfThreads * 16)) = rb[u];
    };
    load_chunk(c0);
    __syncthreads();                                            // the previous segment's reads of buffer 0 (and of the reduction scratch) are done
    store_chunk(0);
    __syncthreads();
    for (int ch = c0; ch < c1; ++ch) {
      const int buf = (ch - c0) & 1;
      const float* sA = smem + buf * (SA + SB);
      if (ch + 1 < c1) load_chunk(ch + 1);
      mma_chunk<true, B_KC, 0, BN, TM, TN>(sA, sA + SA, wm0, wn0, lane, acc);
      if (ch + 1 < c1) store_chunk(buf ^ 1);
      __syncthreads();
    }

    bool finalize = c0 == 0 && c1 == nchunks;
    if (!finalize) {
      // contributors of this tile: the workgroups whose shares meet [T0, T1), in order
      const long long T0 = (long long)tile * nchunks, T1 = T0 + nchunks;
      long long wa = T0 * G / I; while (wg_start(wa + 1) <= T0) ++wa; while (wg_start(wa) > T0) --wa;
      long long wb_ = (T1 - 1) * G / I; while (wg_start(wb_ + 1) <= T1 - 1) ++wb_; while (wg_start(wb_) > T1 - 1) --wb_;
      const int nc = (int)(wb_ - wa + 1);
      // the slab leaves this workgroup WRITE-THROUGH (sc1: no release fence, which would write back the whole XCD L2); every wave drains its
      // stores, the barrier collects the waves, one lane draws the tile's ticket
      const unsigned mine = (unsigned)((blockIdx.x * 2 + (c0 > 0 ? 0 : 1)) * (BM * BN * 4)) + (unsigned)tid * 16u;
#pragma unroll
      for (int it = 0; it < TM; ++it)
#pragma unroll
        for (int jt = 0; jt < TN; ++jt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4v v; v[0] = acc[it][jt][4 * q]; v[1] = acc[it][jt][4 * q + 1]; v[2] = acc[it][jt][4 * q + 2]; v[3] = acc[it][jt][4 * q + 3];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4r, v), rs_slots, (int)(mine + (unsigned)(((it * TN + jt) * 4 + q) * kCfThreads * 16)), 0, kSkStoreAux);
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        const unsigned prev = __hip_atomic_fetch_add(sk.counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = prev == (unsigned)(nc - 1);
        if (last) __hip_atomic_store(sk.counters + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-arm for the next (stream-ordered) launch
        s_last = last;
      }
      __syncthreads();
      finalize = s_last != 0;
      if (finalize) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // the last arriver: every slab of the tile is in memory; EVERY load of them is an sc1 load (past this CU's L1, which no other CU's store
        // refreshes).  Fixed order: the sum does not depend on who finalizes.
#pragma unroll
        for (int it = 0; it < TM; ++it)
#pragma unroll
          for (int jt = 0; jt < TN; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[it][jt][r] = 0.f;
        // The slabs are added in workgroup order, one 32 x 32 sub-tile of the wave at a time, two slabs' loads of that sub-tile in flight (8 x 16 bytes per lane).
        // (Rounds 3 - 4 kept two WHOLE slabs in flight: 128 registers on top of the accumulators made every instance a 251 - 256-register kernel -- two waves per
        // SIMD then fill the register file and no wave of another stream's kernel fits beside them (profiles/EXPERIMENTS.md, round 5).  Element by element the
        // order of the additions is the same: same bits.)
        auto slab = [&](long long w) { return (unsigned)(((int)w * 2 + (wg_start(w) > T0 ? 0 : 1)) * (BM * BN * 4)) + (unsigned)tid * 16u; };
#pragma unroll
        for (int it = 0; it < TM; ++it)
#pragma unroll
          for (int jt = 0; jt < TN; ++jt) {
            const unsigned sub = (unsigned)((it * TN + jt) * 4 * kCfThreads * 16);
            auto add4 = [&](const f32x4v (&v)[4]) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                acc[it][jt][4 * q] += v[q][0]; acc[it][jt][4 * q + 1] += v[q][1]; acc[it][jt][4 * q + 2] += v[q][2]; acc[it][jt][4 * q + 3] += v[q][3];
              }
            };
            long long w = wa;
            for (; w + 1 <= wb_; w += 2) {
              const unsigned s0 = slab(w) + sub, s1 = slab(w + 1) + sub;
              f32x4v v0[4], v1[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) v0[q] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs_slots, (int)(s0 + (unsigned)(q * kCfThreads * 16)), 0, kSkLoadAux));
#pragma unroll
              for (int q = 0; q < 4; ++q) v1[q] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs_slots, (int)(s1 + (unsigned)(q * kCfThreads * 16)), 0, kSkLoadAux));
              add4(v0); add4(v1);
            }
            if (w <= wb_) {
              const unsigned s0 = slab(w) + sub;
              f32x4v v0[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) v0[q] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs_slots, (int)(s0 + (unsigned)(q * kCfThreads * 16)), 0, kSkLoadAux));
              add4(v0);
            }
          }
      }
    }
    if (!finalize) continue;

    // ---- epilogue of a complete tile (dense destination): plain store, or the fold; statistics partials per tile
    float st_s[TN], st_q[TN];
#pragma unroll
    for (int jt = 0; jt < TN; ++jt) { st_s[jt] = 0.f; st_q[jt] = 0.f; }
    unsigned coff[TN];
#pragma unroll
    for (int jt = 0; jt < TN; ++jt) { const int c = n0 + wn0 + jt * 32 + l31; coff[jt] = c < g.Cd ? (unsigned)c * 4u : kOob; }
    const unsigned rowbytes = (unsigned)g.Cd * 4u;
    if (AFF) {
      // y = [relu](acc * scale + shift [+ res]): bn_apply_kernel's expression (separate multiply and add); dense destination (forward)
      const rsrc_t rs_res = make_rsrc(fz.res ? fz.res : dst, fz.res ? g.dst_bytes : 0u);
      const bool has_res = fz.res != nullptr, relu = fz.relu != 0;
      float sc[TN], sh[TN]; unsigned coffa[TN];
#pragma unroll
      for (int jt = 0; jt < TN; ++jt) {
        const int c = n0 + wn0 + jt * 32 + l31; const bool okc = c < g.Cd;
        sc[jt] = okc ? fz.scale[c] : 0.f; sh[jt] = okc ? fz.shift[c] : 0.f; coffa[jt] = okc ? (unsigned)c * 4u : kOob;
      }
#pragma unroll
      for (int it = 0; it < TM; ++it) {
#pragma unroll
        for (int r8 = 0; r8 < 16; r8 += 8) {
          float rv[8][TN];
#pragma unroll
          for (int rr = 0; rr < 8; ++rr) {
            const int r = r8 + rr;
            const int m = m0 + wm0 + it * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const unsigned poff = m < g.Mg ? (unsigned)m * rowbytes : kOob;
#pragma unroll
            for (int jt = 0; jt < TN; ++jt)
              rv[rr][jt] = has_res ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_res, (int)((poff + coffa[jt]) | ((poff | coffa[jt]) & kOob)), 0, 0)) : 0.f;
          }
#pragma unroll
          for (int rr = 0; rr < 8; ++rr) {
            const int r = r8 + rr;
            const int m = m0 + wm0 + it * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const unsigned poff = m < g.Mg ? (unsigned)m * rowbytes : kOob;
#pragma unroll
            for (int jt = 0; jt < TN; ++jt) {
              float v = acc[it][jt][r] * sc[jt] + sh[jt];
              if (has_res) v += rv[rr][jt];
              if (relu) v = v > 0.0f ? v : 0.0f;
              bstore1(v, rs_dst, (poff + coffa[jt]) | ((poff | coffa[jt]) & kOob));
            }
          }
        }
      }
    } else
    if (FOLD) {
      const rsrc_t rs_dres = make_rsrc(fz.dres ? fz.dres : dst, fz.dres ? g.dst_bytes : 0u), rs_xbn = make_rsrc(fz.xbn, g.dst_bytes);
      const rsrc_t rs_mask = make_rsrc(fz.mask ? (const void*)fz.mask : (const void*)dst, fz.mask ? fz.mask_bytes : 0u);
      const bool has_mask = fz.mask != nullptr;
      const unsigned cvrow = (unsigned)g.Cd >> 3;
      float mu[TN], is[TN]; unsigned mcv[TN], mbit[TN];
#pragma unroll
      for (int jt = 0; jt < TN; ++jt) {
        const int c = n0 + wn0 + jt * 32 + l31; const bool okc = c < g.Cd;
        mu[jt] = okc ? fz.mean[c] : 0.f; is[jt] = okc ? fz.invstd[c] : 0.f;
        const int hc = g.Cd >> 1; const int cc = c < hc ? c : c - hc;
        mcv[jt] = (unsigned)(cc >> 2); mbit[jt] = (unsigned)((cc & 3) + (c < hc ? 0 : 4));
      }
#pragma unroll
      for (int it = 0; it < TM; ++it) {
#pragma unroll
        for (int r8 = 0; r8 < 16; r8 += 2) {
          __builtin_amdgcn_sched_barrier(0);     // one small batch of loads live at a time: hoisted, they made this a 254-register instance
          float dv[2][TN], xv[2][TN]; unsigned mb[2][TN];
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            const int r = r8 + rr;
            const int m = m0 + wm0 + it * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const unsigned poff = m < g.Mg ? (unsigned)m * rowbytes : kOob;
#pragma unroll
            for (int jt = 0; jt < TN; ++jt) {
              const unsigned off = (poff + coff[jt]) | ((poff | coff[jt]) & kOob);
              dv[rr][jt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_dres, (int)off, 0, 0));
              xv[rr][jt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_xbn, (int)off, 0, 0));
              mb[rr][jt] = has_mask ? (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs_mask, (int)(m < g.Mg ? (unsigned)m * cvrow + mcv[jt] : kOob), 0, 0) : 0xffu;
            }
          }
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            const int r = r8 + rr;
            const int m = m0 + wm0 + it * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const unsigned poff = m < g.Mg ? (unsigned)m * rowbytes : kOob;
#pragma unroll
            for (int jt = 0; jt < TN; ++jt) {
              float a = acc[it][jt][r] + dv[rr][jt];
              a = ((mb[rr][jt] >> mbit[jt]) & 1u) && m < g.Mg ? a : 0.f;
              st_s[jt] += a; st_q[jt] += a * ((xv[rr][jt] - mu[jt]) * is[jt]);
              bstore1(a, rs_dst, (poff + coff[jt]) | ((poff | coff[jt]) & kOob));
            }
          }
        }
      }
    } else {
#pragma unroll
      for (int it = 0; it < TM; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm0 + it * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const unsigned poff = m < g.Mg ? (unsigned)m * rowbytes : kOob;
#pragma unroll
          for (int jt = 0; jt < TN; ++jt) {
            bstore1(acc[it][jt][r], rs_dst, (poff + coff[jt]) | ((poff | coff[jt]) & kOob));
            if (STATS) { const float v = acc[it][jt][r]; st_s[jt] += v; st_q[jt] += v * v; }    // rows past Mg are 0
          }
        }
    }
    if (STATS || FOLD) {
      // lane halves -> the two waves of a column block -> the tile's partial row: part[mt][2][Cd]
      float* red = smem;                                        // [WM][2 stats][BN]  (the K loop's last barrier has passed)
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
        const int sx = i / BN, c = i - sx * BN;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) v += red[(w * 2 + sx) * BN + c];
        if (n0 + c < g.Cd) part[((int64_t)mt * 2 + sx) * g.Cd + n0 + c] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradient: A[co][k = pixel] = dY (k slow), B[k = pixel][(tap, ci)] = X gathered (k slow); split over K, float atomics.
// The tile is 64 (co) x 256 (tap, ci): a 256-wide B row is ONE wave-instruction (64 lanes x 16 bytes), so the pixel a piece
// belongs to is wave-uniform and its decode runs on the scalar ALU; per lane only the (static) tap of its four columns matters.

// SM (gathered layers with Cin >= 64: a 256-column tile spans at most four taps): the bounds test of a gathered piece runs on the SCALAR
// unit.  The pixel of a piece is wave-uniform and so is each tap of the tile; which lanes belong to which tap is a constant 64-bit lane
// mask per tap "slot" (ballot, once per work item): valid lanes of a piece = OR over the slots of (the slot's tap lands inside the image at
// this pixel ? its lane mask : 0) -- scalar compares and selects, issued in the shadow of the wave's own MFMAs -- and reaches the
// vector unit as the predicate of ONE v_cndmask per piece (round 2: two adds, two compares and a select per piece on the vector ALU,
// 1.69 vector instructions per MFMA, which the f32 MFMA cannot overlap).
// MEASURED (round 3, same box, 512 images, us per launch, vector test / scalar masks): 3x3 64 -> 64 @56 1689 / 1777, 128 -> 128 @28 1078 / 1114,
// stride-2 128 -> 128 1080 / 1115, 256 -> 256 @14 1222 / 1297, 512 -> 512 @7 1131 / 1237, strided 1x1 512 -> 1024 871 / 909: 3 - 9 % SLOWER.  The
// loader drops from 28 to 8 vector instructions per chunk but grows from ~140 to ~290 scalar ones, and a wave issues in order: the scalar
// run sits between two MFMA blocks (the prefetch is a basic block of its own) and the matrix pipe waits for it.  Off by default
// (LEC_WGRAD_SMASK=1 enables it); it pays only once the prefetch is interleaved with the MFMAs instruction by instruction.
// SH ("shifted dense", round 4): a stride-1 layer whose output grid equals its input grid (3x3 / pad 1) and whose column tile lies inside ONE tap.
// Then dW[co][tap][ci] = sum_m dY[m][co] * X[m + shift(tap)][ci] over the pixels m whose tap lands inside the image: the B operand is the DENSE
// loader on x at a constant byte offset (no per-piece pixel decode, no bounds test, no scalar work at all; rows that fall off the tensor are the
// buffer's range check), and the pixels whose tap leaves the image are dropped on the A side -- the dY row -- by ONE table lookup per piece: a
// 16-bit mask of the taps that leave the image per pixel position, built once per workgroup in LDS; the position walk is scalar, the entry is read
// one chunk ahead and lands in bit 31 of the load's OFFSET (out of range: the load returns zeros) -- three vector instructions per piece where
// selects on the loaded data and a per-lane walk took eleven (1.11 -> see profiles/r04_conv_f32_pmc.md).  With it the tile can follow the layer (128 x 128 for Cin = 128: no half-empty fifth 256-column tile) and the kernel is the dense
// split-K weight gradient, which holds the matrix pipe 83 % busy where the gathered form holds it 76 %.
// SH = 2: the same for STRIDE-2 layers whose input grid is exactly twice the output grid (3x3 / pad 1 and the 1x1 downsample layers of ResNet):
// the source pixel of output pixel m = (n, ho, wo) under tap (r, s) is 4 m - 2 wo + const -- affine in m but for the column wo, which every B piece
// reads from a second wrapped table ((i mod Wo) * bytes) at "scalar column of the chunk's first row + own row", one chunk ahead like the A side.
template <int WM, int WN, int TM, int TN, int WBK, bool DENSE, bool XF = false, bool SM = false, int SH = 0>
__global__ __launch_bounds__(kCfThreads, (XF && WM <= 2 && TM * TN == 4) ? kWgXfBlocks : 2) void conv_f32_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                       float* __restrict__ dw, WgGeo g, const float* __restrict__ xsrc = nullptr,
                                                                       const float* __restrict__ coef = nullptr) {
  static_assert(!XF || DENSE, "the on-load BatchNorm form serves the dense (1x1 / stride 1) layers");
  static_assert(!SH || (DENSE && !XF && !SM), "the shifted-dense form is the dense loader");
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  static_assert(WM * WN == 4, "four waves per workgroup");
  static_assert(DENSE || BN == 256, "a gathered B row must be one wave-instruction (256 columns) for the scalar pixel decode");
  constexpr int PA = BM / 4, PB = BN / 4;                     // 16-byte pieces per k row
  constexpr int NA = WBK * PA / kCfThreads, NB = WBK * PB / kCfThreads;
  static_assert(NA >= 1 && NB >= 1, "tile too small for 256 threads");
  constexpr int SA = WBK * BM, SB = WBK * BN;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave / WN) * 32 * TM, wn0 = (wave % WN) * 32 * TN;
  const int ntn = (g.Ng + BN - 1) / BN;
  const int nchunks_all = (g.Mpix + WBK - 1) / WBK;
  const rsrc_t rs_dy = make_rsrc(dy, g.dy_bytes), rs_x = make_rsrc(x, g.x_bytes);
  const rsrc_t rs_x2 = make_rsrc(XF ? xsrc : dy, g.dy_bytes);
  // SH tables in LDS, built once per workgroup, both with WBK wrapped entries at the end so that "position of the chunk's first row (scalar) + row of
  // the piece (per-thread constant)" indexes them without a modulo: vtab[i] = taps that LEAVE the image at pixel position i mod HoWo (bit t = tap t);
  // SH = 2: wtab[i] = (i mod Wo) * 2 Cin * 4, the bytes "- 2 wo" takes off the source offset
  unsigned short* vtab = (unsigned short*)(smem + 2 * (SA + SB));
  int* wtab = (int*)(vtab + ((g.HoWo + WBK + 1) & ~1));
  if (SH) {
    for (int i = tid; i < g.HoWo + WBK; i += kCfThreads) {
      const int p = i < g.HoWo ? i : i - g.HoWo;
      const int ho = fdiv(p, g.dWo), wo = p - ho * g.Wo;
      unsigned m = 0;
      for (int tp = 0; tp < g.RS; ++tp) {
        const int r = fdiv(tp, g.dS), s_ = tp - r * g.S;
        if (!((unsigned)(ho * g.stride + r - g.pad) < (unsigned)g.H && (unsigned)(wo * g.stride + s_ - g.pad) < (unsigned)g.W)) m |= 1u << tp;
      }
      vtab[i] = (unsigned short)m;
    }
    if (SH == 2)
      for (int i = tid; i < g.Wo + WBK; i += kCfThreads) wtab[i] = (i - fdiv(i, g.dWo) * g.Wo) * 2 * g.Cin * 4;
    __syncthreads();
  }
  // work items = (output tile, K split); a workgroup walks its share when LEC_WGRAD_WGS caps the grid.  (Measured on the fp32 step:
  // one workgroup per CU leaves the main stream's HBM-bound BatchNorm kernels room -- their time drops 63.8 -> 50.9 ms -- but the
  // convolutions beside it stretch more than that: 164.7 ms per step against 157.0 uncapped, so the default is no cap.)
  // Slot -> work item: workgroup ids go round-robin over the 8 XCDs (one L2 each); items are numbered K-split major and XCD x takes the
  // contiguous run [x * per, (x + 1) * per): the tiles that share a K range (the same dY rows for all taps / column tiles, the same X rows for
  // every Cout tile) are resident on ONE XCD together, and their re-reads hit its L2 instead of HBM (2.7 GB of reads on a 0.4 GB layer before).
  const int per = (g.tiles * g.split + 7) / 8;
  for (int slot = blockIdx.x; slot < 8 * per; slot += gridDim.x) {
  const int wi = (slot & 7) * per + (slot >> 3);
  if (wi >= g.tiles * g.split) continue;
  const int tile = wi % g.tiles, sp = wi / g.tiles;
  const int tm = tile / ntn, tn = tile - tm * ntn;
  const int co0 = tm * BM, j0 = tn * BN;
  const int ch_lo = sp * g.chunks_per_split;
  const int ch_hi = min(nchunks_all, ch_lo + g.chunks_per_split);

  // A pieces: v = tid + 256 u -> k row v / PA, co piece v % PA: byte offset inside dY of chunk 0, + mbase * Cout * 4 per chunk;
  // rows past Mpix fall out of the buffer's range
  unsigned aoff[NA];
#pragma unroll
  for (int u = 0; u < NA; ++u) {
    const int v = tid + kCfThreads * u; const int kr = v / PA, cq = v - kr * PA; const int co = co0 + 4 * cq;
    aoff[u] = co < g.Cout ? (unsigned)(kr * g.Cout + co) * 4u : kOob;
  }
  // XF: dy = cA * g + cB * xsrc + cD per output channel; a thread's A pieces keep their four channels for the whole work item
  f32x4v cfA[XF ? NA : 1], cfB[XF ? NA : 1], cfD[XF ? NA : 1];
  if (XF) {
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int v = tid + kCfThreads * u; const int cq = v % PA; const int co = co0 + 4 * cq;
      const bool ok = co < g.Cout;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        cfA[u][e] = ok ? coef[co + e] : 0.f; cfB[u][e] = ok ? coef[g.Cout + co + e] : 0.f; cfD[u][e] = ok ? coef[2 * g.Cout + co + e] : 0.f;
      }
    }
  }
  // B pieces.  DENSE (1x1 / stride 1: the source pixel of m is m): like A, on x.  Otherwise k row = wave + 4 u (wave-uniform),
  // columns j0 + 4 lane .. + 3 (inside one tap: Cin % 4 == 0), whose tap is a per-lane constant.
  unsigned boff[NB];
  const int tapT = SH ? (j0 >> g.lgCin) : 0;                   // SH: the ONE tap of this column tile (scalar)
  int shiftB = 0;
  if (SH) { const int rT = fdiv(tapT, g.dS); shiftB = (((rT - g.pad) * g.W + (tapT - rT * g.S - g.pad)) * g.Cin) * 4; }
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    const int v = tid + kCfThreads * u; const int kr = v / PB, jq = v - kr * PB; const int jj = j0 + 4 * jq;
    boff[u] = jj < g.Ng ? (unsigned)((SH == 2 ? 4 * kr : kr) * g.Cin + (SH ? (jj & (g.Cin - 1)) : jj)) * 4u : kOob;
  }
  // SH: the walk over pixel positions is SCALAR (position of the first row of the next chunk whose table entries get read: spix inside its image, swo
  // inside its output row); a piece adds its own row (constant) and reads the wrapped table ONE CHUNK AHEAD of the load that uses the entry, so the
  // LDS latency sits behind a whole chunk of MFMAs.  aval: invalid-tap bits of each A piece's row; bdel: "2 wo" bytes of each B piece's row.
  unsigned aval[SH ? NA : 1]; int bdel[SH == 2 ? NB : 1];
  int spix = 0, swo = 0;
  const int wstep = SH == 2 ? WBK - fdiv(WBK, g.dWo) * g.Wo : 0;  // WBK mod Wo
  if (SH) {
    const int m0 = ch_lo * WBK;
    spix = m0 - fdiv(m0, g.dHW) * g.HoWo;
#pragma unroll
    for (int u = 0; u < NA; ++u) aval[u] = (unsigned)vtab[spix + (tid + kCfThreads * u) / PA];
    spix += WBK; spix -= spix >= g.HoWo ? g.HoWo : 0;
    if (SH == 2) {
      swo = m0 - fdiv(m0, g.dWo) * g.Wo;
#pragma unroll
      for (int u = 0; u < NB; ++u) bdel[u] = wtab[swo + (tid + kCfThreads * u) / PB];
      swo += wstep; swo -= swo >= g.Wo ? g.Wo : 0;
    }
  }
  const int j = j0 + 4 * lane;
  const int tapL = j >> g.lgCin, ciL = j & (g.Cin - 1);
  const int rL = fdiv(tapL, g.dS);
  const int drL = rL - g.pad, dsL = tapL - rL * g.S - g.pad;
  const bool okL = j < g.Ng;
  constexpr int kSlots = 4;
  unsigned long long slotmask[SM ? kSlots : 1]; int sdr[SM ? kSlots : 1], sds[SM ? kSlots : 1];
  if (SM) {
    const int tap0 = j0 >> g.lgCin;                             // scalar: first tap of this column tile
#pragma unroll
    for (int q = 0; q < kSlots; ++q) {
      const int tq = tap0 + q; const int rq = fdiv(tq, g.dS);
      sdr[q] = rq - g.pad; sds[q] = tq - rq * g.S - g.pad;
      slotmask[q] = __builtin_amdgcn_ballot_w64(okL && tapL == tq);
    }
  }
  int laneoff = ((drL * g.W + dsL) * g.Cin + ciL) * 4;         // this lane's tap / channel relative to the pixel's (0, 0) tap, bytes
  asm volatile("" : "+v"(laneoff));                          // opaque: the compiler otherwise re-derives pixoff + laneoff from (h0 + drL, w0 + dsL)
                                                              // with a 64-bit multiply-add and a v_mul_lo per piece -- vector time the f32 MFMA pays for
  f32x16 acc[TM][TN];
#pragma unroll
  for (int it = 0; it < TM; ++it)
#pragma unroll
    for (int jt = 0; jt < TN; ++jt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[it][jt][r] = 0.f;

  f32x4v ra[NA], rb[NB], rx[XF ? NA : 1];
  auto load_chunk = [&](int ch) {
    const int mbase = ch * WBK;
    const unsigned abase = (unsigned)(mbase * g.Cout) * 4u;
    if (SH) {                                                   // a dY row whose tap leaves the image: the load itself returns zeros (offset bit 31 = out of range)
#pragma unroll
      for (int u = 0; u < NA; ++u) ra[u] = bload4(rs_dy, (aoff[u] + abase) | (((aval[u] >> tapT) & 1u) << 31));
#pragma unroll
      for (int u = 0; u < NA; ++u) aval[u] = (unsigned)vtab[spix + (tid + kCfThreads * u) / PA];     // chunks are loaded in order: the entries of the NEXT call
      spix += WBK; spix -= spix >= g.HoWo ? g.HoWo : 0;
    } else {
#pragma unroll
      for (int u = 0; u < NA; ++u) ra[u] = bload4(rs_dy, aoff[u] + abase);
    }
    if (XF) {
#pragma unroll
      for (int u = 0; u < NA; ++u) rx[u] = bload4(rs_x2, aoff[u] + abase);
    }
    if (DENSE) {
      const unsigned bbase = (unsigned)((SH == 2 ? 4 * mbase : mbase) * g.Cin) * 4u + (unsigned)shiftB;
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        if (SH == 2) rb[u] = bload4(rs_x, boff[u] + bbase - (unsigned)bdel[u]);
        else rb[u] = bload4(rs_x, boff[u] + bbase);
      }
      if (SH == 2) {
#pragma unroll
        for (int u = 0; u < NB; ++u) bdel[u] = wtab[swo + (tid + kCfThreads * u) / PB];
        swo += wstep; swo -= swo >= g.Wo ? g.Wo : 0;
      }
    } else {
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int m = mbase + wave + 4 * u;                     // scalar
        const bool live = m < g.Mpix;
        const int mm = live ? m : 0;
        const int t2 = fdiv(mm, g.dWo); const int wo = mm - t2 * g.Wo; const int n = fdiv(t2, g.dHo); const int ho = t2 - n * g.Ho;
        const int h0 = ho * g.stride, w0 = wo * g.stride;       // all scalar up to here
        const int pixoff = ((n * g.H + h0) * g.W + w0) * g.Cin * 4;
        if (SM) {
          unsigned long long vm = 0ull;
#pragma unroll
          for (int q = 0; q < kSlots; ++q) {
            const bool in = (unsigned)(h0 + sdr[q]) < (unsigned)g.H && (unsigned)(w0 + sds[q]) < (unsigned)g.W;
            vm |= in ? slotmask[q] : 0ull;
          }
          vm = live ? vm : 0ull;
          unsigned off;                                         // lanes of vm: pixoff + laneoff, the others: out of range
          asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(off) : "v"(kOob), "v"((unsigned)(pixoff + laneoff)), "s"(vm));
          rb[u] = bload4(rs_x, off);
        } else {
          const bool ok = live && okL && (unsigned)(h0 + drL) < (unsigned)g.H && (unsigned)(w0 + dsL) < (unsigned)g.W;
          rb[u] = bload4(rs_x, ok ? (unsigned)(pixoff + laneoff) : kOob);
        }
      }
    }
  };
  auto store_chunk = [&](int buf) {
    char* base = (char*)smem + buf * (SA + SB) * 4;
    if (XF) {                                                   // (rows past Mpix become cD: their X rows are zero, the products vanish)
#pragma unroll
      for (int u = 0; u < NA; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) ra[u][e] = __builtin_fmaf(cfA[u][e], ra[u][e], __builtin_fmaf(cfB[u][e], rx[u][e], cfD[u][e]));
    }
#pragma unroll
    for (int u = 0; u < NA; ++u) *(f32x4v*)(base + (tid + kCfThreads * u) * 16) = ra[u];
#pragma unroll
    for (int u = 0; u < NB; ++u) *(f32x4v*)(base + SA * 4 + (tid + kCfThreads * u) * 16) = rb[u];
  };
  if (ch_lo < ch_hi) {
    load_chunk(ch_lo);
    __syncthreads();                                            // the previous work item's last reads of buffer 0 are done
    store_chunk(0);
    __syncthreads();
    for (int ch = ch_lo; ch < ch_hi; ++ch) {
      const int buf = (ch - ch_lo) & 1;
      const float* sA = smem + buf * (SA + SB);
      if (ch + 1 < ch_hi) load_chunk(ch + 1);
      mma_chunk<false, false, BM, BN, TM, TN, WBK>(sA, sA + SA, wm0, wn0, lane, acc);
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
            const int jj = j0 + wn0 + jt * 32 + l31;
            if (g.dCin == g.Cin) {
              if (jj < g.Ng) atomicAdd(dw + (int64_t)co * g.Ng + jj, acc[it][jt][r]);
            } else {                                            // the stem: the padded 4th input channel has no slot in dw
              const int tp = jj >> g.lgCin, ci = jj & (g.Cin - 1);
              if (jj < g.Ng && ci < g.dCin) atomicAdd(dw + ((int64_t)co * g.RS + tp) * g.dCin + ci, acc[it][jt][r]);
            }
          }
        }
      }
  }
  }   // work items
}


// Scratch of the balanced kernel, one per stream (launches on a stream are ordered; two streams run side by side): counters first, slots behind.
constexpr int kSkWgs = 512;               // workgroups of a balanced launch: the resident slots of the chip (256 CUs x 2)
constexpr int kSkMaxTiles = 16384;
struct SkScratch { float* slots; unsigned int* counters; };
static std::mutex g_sk_mu;
static std::map<std::pair<int, void*>, SkScratch> g_sk;   // (device, stream)
static inline std::pair<int, void*> sk_key(hipStream_t st) { int d = 0; (void)hipGetDevice(&d); return {d, (void*)st}; }
static inline int64_t sk_scratch_bytes() { return (int64_t)kSkMaxTiles * 4 + (int64_t)kSkWgs * 2 * 128 * 128 * 4; }
static inline bool sk_lookup(hipStream_t st, SkScratch* out) {
  std::lock_guard<std::mutex> lk(g_sk_mu);
  auto it = g_sk.find(sk_key(st));
  if (it == g_sk.end()) return false;
  *out = it->second; return true;
}
// LEC_CF_SK: 0 never, 1 (default) where the tile count leaves the last round of workgroup slots under LEC_CF_SK_FILL (default 0.92) full, 2 wherever
// the kernel applies (tests).
// The entry points' `schedule` argument (LEC_SCHEDULE_*) decides per call; LEC_SCHEDULE_DEFAULT (-1) defers to the environment.
static inline int sk_mode(int schedule) {
  return schedule >= 0 ? (schedule > 2 ? 2 : schedule) : tuning().cf_sk;
}
static inline double sk_fill() { return tuning().cf_sk_fill; }
static inline int sk_min_chunks() { return tuning().cf_sk_min_chunks; }

template <bool B_KC, bool STATS, int FUSE = 0>
static int launch_act(const float* src, const float* wgt, float* dst, const ActGeo& g, float* part, int* nparts, hipStream_t st,
                      const ActFuse& fz = ActFuse{}, int part_rows = kCfMaxPart, int schedule = -1) {
  // column tile: 128 wide unless the layer has 64 output channels
  const bool narrow = g.Cd <= 64;
  const int BM = 128, BN = narrow ? 64 : 128;
  const int mtiles = (g.Mg + BM - 1) / BM, ntiles = (g.Cd + BN - 1) / BN;
  if constexpr ((FUSE & 1) == 0) {
    // the balanced form: 128 x 128 tiles of a dense destination, whole-tap chunks (see conv_f32_act_sk_kernel)
    const int mode = sk_mode(schedule);
    const long long tiles = (long long)mtiles * ntiles;
    const int nchunks = (g.Kg + kCfBK - 1) / kCfBK;
    SkScratch sc;
    if (mode != 0 && !narrow && g.dst_st == 1 && g.Cs % kCfBK == 0 && g.na * g.nb <= 32 && nchunks > 0 && tiles <= kSkMaxTiles &&
        (!(STATS || (FUSE & 2)) || mtiles <= part_rows) && sk_lookup(st, &sc)) {
      // workgroups: the chip's 512 slots, fewer when that would leave a workgroup under LEC_CF_SK_MIN_CHUNKS (8) chunks of work (small batches: the
      // reference's 10-image evaluation chunks give layer3 / layer4 16 - 128 tiles of 32 - 144 chunks each; cut along K they occupy the chip)
      const long long iters = tiles * nchunks;
      long long gmax = mode == 2 ? iters : iters / sk_min_chunks();
      if (gmax < 1) gmax = 1;
      // (LEC_CF_SK_WGS: workgroups of a balanced launch.  512 = every resident slot of the chip; 256 = ONE per CU, so that the balanced launches of two
      // concurrent passes sit side by side -- 2 x 74 KB of LDS, 2 x 190 registers per SIMD lane -- instead of queueing behind each other)
      const int kw = tuning().cf_sk_wgs < kSkWgs ? tuning().cf_sk_wgs : kSkWgs;
      const int G = (int)(gmax < kw ? gmax : kw);
      const double rounds = (double)tiles / kSkWgs;
      const double fill = rounds / (double)((tiles + kSkWgs - 1) / kSkWgs);
      // ... taken where the tile walk's last round is poorly filled AND the cut gives more parallel work than whole tiles would (tiles < 512), or
      // the same 512 workgroups an even share (tiles > 512)
      if (mode == 2 || (fill < sk_fill() && (tiles >= kSkWgs ? G == kw : (long long)G * 4 >= tiles * 5))) {
        const size_t lds = (size_t)2 * (BM * kCfLdk + (B_KC ? BN * kCfLdk : kCfBK * BN)) * 4;
        SkArgs sk{sc.slots, sc.counters, ntiles, (int)tiles};
        ActGeo gg = g; gg.xcd_per = 0;
        hipLaunchKernelGGL((conv_f32_act_sk_kernel<B_KC, 2, 2, STATS, FUSE>), dim3(G), dim3(kCfThreads), lds, st, src, wgt, dst, gg, part, fz, sk);
        if (nparts) *nparts = mtiles;
        LEC_CHECK_LAUNCH("conv_f32_act_sk_kernel");
        return LEC_OK;
      }
    }
  }
  int gx = mtiles;
  if (STATS || (FUSE & 2)) { const int cap = kCfMaxPart; if (gx > cap) gx = cap; }
  else { const int cap = 2048 / (ntiles > 8 ? 8 : ntiles); if (gx > cap) gx = cap; }
  // (Small launches of ANOTHER stream, round 4, tools/microbench/small_launch_under_load.py: next to this tile-walk kernel they are placed as fast as on an idle chip (a fill 4 us,
  // a one-workgroup kernel 8); next to the balanced form below -- 512 workgroups resident for the whole kernel -- they wait 60 - 500 us each.  Multi-stream steps take the tile walk.)
  if (gx < 1) gx = 1;
  // XCD-contiguous tile runs (LEC_CF_XCD=1; default off): the grid becomes a multiple of 8 so that a workgroup's slots stay on its XCD.
  // MEASURED (round 3, same box, alternating runs, 512 images): forward / data gradient of nine layer shapes identical to +-1 % with and
  // without (3x3 128 -> 128 @28: 961 / 966 us forward, 961 / 960 us data gradient; 256 -> 256 @14: 1012 / 1013, 1019 / 1019), bench step 131.3 - 131.6 ms
  // with, 130.9 without: the re-reads the round-robin order causes are served by the Infinity Cache at no cost to the matrix pipe.
  const int cf_xcd = tuning().cf_xcd;
  ActGeo gg = g;
  gg.xcd_per = 0;
  if (cf_xcd && mtiles >= 64) {
    gg.xcd_per = (mtiles + 7) / 8;
    gx = (gx + 7) / 8 * 8;
    const int cap = (STATS || (FUSE & 2)) ? kCfMaxPart : 2048 / (ntiles > 8 ? 8 : ntiles);
    if (gx > cap) gx = cap / 8 * 8;
    if (gx > 8 * gg.xcd_per) gx = 8 * gg.xcd_per;
  }
  size_t lds = (size_t)2 * (BM * kCfLdk + (B_KC ? BN * kCfLdk : kCfBK * BN)) * 4;      // 74 / 55 KB: two workgroups per CU
  if (tuning().cf_lds_pad > 0) lds += (size_t)tuning().cf_lds_pad;                     // experiments: force one workgroup per CU
  const bool tapv = g.Cs % kCfBK != 0;                          // source channels narrower than a chunk (the stem)
  LEC_CHECK_ARG(tapv || g.na * g.nb <= 32, "conv_f32: more than 32 taps per launch need the per-piece tap path");
  if constexpr (FUSE == 4) {
    if (tapv) {                                                 // the stem with an eval-mode BatchNorm behind it
      if (narrow) hipLaunchKernelGGL((conv_f32_act_kernel<B_KC, 2, 2, 2, 1, STATS, true, 4>), dim3(gx, ntiles), dim3(kCfThreads), lds, st, src, wgt, dst, gg, part, fz);
      else hipLaunchKernelGGL((conv_f32_act_kernel<B_KC, 2, 2, 2, 2, STATS, true, 4>), dim3(gx, ntiles), dim3(kCfThreads), lds, st, src, wgt, dst, gg, part, fz);
      if (nparts) *nparts = gx;
      LEC_CHECK_LAUNCH("conv_f32_act_kernel");
      return LEC_OK;
    }
  }
  if (FUSE != 0) {
    LEC_CHECK_ARG(!tapv, "conv_f32: the fused modes need source channels that are a multiple of the K chunk (%d)", kCfBK);
    if (narrow) hipLaunchKernelGGL((conv_f32_act_kernel<B_KC, 2, 2, 2, 1, STATS, false, FUSE>), dim3(gx, ntiles), dim3(kCfThreads), lds, st, src, wgt, dst, gg, part, fz);
    else hipLaunchKernelGGL((conv_f32_act_kernel<B_KC, 2, 2, 2, 2, STATS, false, FUSE>), dim3(gx, ntiles), dim3(kCfThreads), lds, st, src, wgt, dst, gg, part, fz);
  } else if (tapv) {
    if (narrow) hipLaunchKernelGGL((conv_f32_act_kernel<B_KC, 2, 2, 2, 1, STATS, true>), dim3(gx, ntiles), dim3(kCfThreads), lds, st, src, wgt, dst, gg, part, fz);
    else hipLaunchKernelGGL((conv_f32_act_kernel<B_KC, 2, 2, 2, 2, STATS, true>), dim3(gx, ntiles), dim3(kCfThreads), lds, st, src, wgt, dst, gg, part, fz);
  } else {
    if (narrow) hipLaunchKernelGGL((conv_f32_act_kernel<B_KC, 2, 2, 2, 1, STATS, false>), dim3(gx, ntiles), dim3(kCfThreads), lds, st, src, wgt, dst, gg, part, fz);
    else hipLaunchKernelGGL((conv_f32_act_kernel<B_KC, 2, 2, 2, 2, STATS, false>), dim3(gx, ntiles), dim3(kCfThreads), lds, st, src, wgt, dst, gg, part, fz);
  }
  if (nparts) *nparts = gx;
  LEC_CHECK_LAUNCH("conv_f32_act_kernel");
  return LEC_OK;
}



// Batches whose tensors reach 2 GiB (32-bit byte offsets, conv_check) run as several launches over GROUPS of images: a convolution is independent per
// image, so the entry points below hand each group its own base pointers (forward / data gradient: outputs of disjoint images; weight gradient: atomics
// into the same slot; statistics / fold partials: consecutive rows of the caller's buffer, the BatchNorm finalize sums whatever rows it is given).
// fp32 ResNet-50 at 224 x 224: 668 rows per launch -- the reference's own separate forwards of a step (B K = 1 280 / 2 560 rows at B = 256: oe_h.py:980-985,
// 1003-1009, `reference_exact_batches`) used to fall back to the library silently.
static inline int conv_group_images(int N, int64_t in_img_bytes, int64_t out_img_bytes) {
  const int64_t lim = (1ll << 31) - 1;
  const int64_t big = in_img_bytes > out_img_bytes ? in_img_bytes : out_img_bytes;
  const int64_t gmax = big > 0 ? lim / big : N;
  return (int)(gmax < 1 ? 0 : (gmax < N ? gmax : N));
}
}  // namespace lec

// Scratch of the balanced forward / data-gradient kernel for the launches of ONE stream: `buf` holds lec_conv_f32_scratch_bytes() bytes, ZEROED
// by the caller, and stays alive until it is unregistered (buf = null) -- also across replays of a graph captured meanwhile.  Without a
// registered scratch the stream's launches use the tile-walk kernel (same results to the last few bits: the K sum is split differently).
// Which form a call takes is its own `schedule` argument: LEC_SCHEDULE_TILE_WALK (0) never the balanced form, LEC_SCHEDULE_AUTO (1) where it pays,
// LEC_SCHEDULE_BALANCED (2) wherever it applies, LEC_SCHEDULE_DEFAULT (-1) what LEC_CF_SK in the environment says (default 1).
extern "C" int64_t lec_conv_f32_scratch_bytes(void) { return lec::sk_scratch_bytes(); }
extern "C" int lec_conv_f32_scratch(lec_stream_t stream, void* buf, int64_t bytes) {
  using namespace lec;
  LEC_CHECK_ARG(!buf || bytes >= sk_scratch_bytes(), "conv_f32_scratch: %lld bytes, need %lld", (long long)bytes, (long long)sk_scratch_bytes());
  LEC_CHECK_ARG(((uintptr_t)buf & 15) == 0, "conv_f32_scratch: buffer must be 16-byte aligned");
  std::lock_guard<std::mutex> lk(g_sk_mu);
  if (!buf) { g_sk.erase(sk_key((hipStream_t)stream)); return LEC_OK; }
  g_sk[sk_key((hipStream_t)stream)] = SkScratch{(float*)((char*)buf + (int64_t)kSkMaxTiles * 4), (unsigned int*)buf};
  return LEC_OK;
}

static int conv_f32_fwd_one(const float* x, const float* w, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                           float* y, float* partials, int64_t partials_bytes, int* n_partials, int schedule, lec_stream_t stream) {
  using namespace lec;
  if (int rc = conv_check("conv_f32_fwd", N, H, W, Cin, Cout, R, S, stride, pad)) return rc;
  LEC_CHECK_ARG(x && w && y, "conv_f32_fwd: null pointer");
  const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
  ActGeo g;
  g.zfill = 0;
  g.Mg = N * Ho * Wo; g.Hm = Ho; g.Wm = Wo; g.Hs = H; g.Ws = W; g.Cs = Cin; g.lgCs = ilog2_exact(Cin); g.sst = stride;
  g.oh0 = -pad; g.ow0 = -pad; g.sg = 1; g.na = R; g.nb = S; g.r0 = 0; g.rstep = 1; g.s0 = 0; g.sstep = 1; g.S = S; g.RS = R * S;
  g.Cd = Cout; g.Cin = Cin; g.Hd = Ho; g.Wd = Wo; g.dst_st = 1; g.dph = 0; g.dpw = 0; g.Kg = R * S * Cin;
  g.src_bytes = (uint32_t)((int64_t)N * H * W * Cin * 4); g.wgt_bytes = (uint32_t)((int64_t)Cout * R * S * Cin * 4);
  g.dst_bytes = (uint32_t)((int64_t)g.Mg * Cout * 4);
  g.dWm = make_fastdiv(g.Wm); g.dHm = make_fastdiv(g.Hm); g.dnb = make_fastdiv(g.nb);
  if (partials) {
    LEC_CHECK_ARG(n_partials && partials_bytes >= (int64_t)kCfMaxPart * 2 * Cout * (int64_t)sizeof(float), "conv_f32_fwd: partials buffer too small");
    const int64_t rows = partials_bytes / ((int64_t)2 * Cout * (int64_t)sizeof(float));
    return launch_act<true, true>(x, w, y, g, partials, n_partials, (hipStream_t)stream, ActFuse{}, (int)(rows < kCfMaxRows ? rows : kCfMaxRows), schedule);
  }
  return launch_act<true, false>(x, w, y, g, nullptr, nullptr, (hipStream_t)stream, ActFuse{}, kCfMaxPart, schedule);
}

// Forward with an eval-mode BatchNorm (+ residual, + ReLU) in the epilogue: y = [relu](conv(x, w) * scale[c] + shift[c] [+ res]) -- see ActFuse AFF.
// scale / shift: Cout floats each (gamma / sqrt(running_var + eps), beta - running_mean * scale); res: null or a tensor of y's shape.
static int conv_f32_fwd_affine_one(const float* x, const float* w, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                                  float* y, const float* scale, const float* shift, const float* res, int relu, int schedule, lec_stream_t stream) {
  using namespace lec;
  if (int rc = conv_check("conv_f32_fwd_affine", N, H, W, Cin, Cout, R, S, stride, pad)) return rc;
  LEC_CHECK_ARG(x && w && y && scale && shift, "conv_f32_fwd_affine: null pointer");
  const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
  ActGeo g;
  g.zfill = 0;
  g.Mg = N * Ho * Wo; g.Hm = Ho; g.Wm = Wo; g.Hs = H; g.Ws = W; g.Cs = Cin; g.lgCs = ilog2_exact(Cin); g.sst = stride;
  g.oh0 = -pad; g.ow0 = -pad; g.sg = 1; g.na = R; g.nb = S; g.r0 = 0; g.rstep = 1; g.s0 = 0; g.sstep = 1; g.S = S; g.RS = R * S;
  g.Cd = Cout; g.Cin = Cin; g.Hd = Ho; g.Wd = Wo; g.dst_st = 1; g.dph = 0; g.dpw = 0; g.Kg = R * S * Cin;
  g.src_bytes = (uint32_t)((int64_t)N * H * W * Cin * 4); g.wgt_bytes = (uint32_t)((int64_t)Cout * R * S * Cin * 4);
  g.dst_bytes = (uint32_t)((int64_t)g.Mg * Cout * 4);
  g.dWm = make_fastdiv(g.Wm); g.dHm = make_fastdiv(g.Hm); g.dnb = make_fastdiv(g.nb);
  ActFuse fz{};
  fz.scale = scale; fz.shift = shift; fz.res = res; fz.relu = relu;
  return launch_act<true, false, 4>(x, w, y, g, nullptr, nullptr, (hipStream_t)stream, fz, kCfMaxPart, schedule);
}

static int conv_f32_dgrad_one(const float* dy, const float* w, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                             float* dx, int schedule, lec_stream_t stream) {
  using namespace lec;
  if (int rc = conv_check("conv_f32_dgrad", N, H, W, Cin, Cout, R, S, stride, pad)) return rc;
  LEC_CHECK_ARG(dy && w && dx, "conv_f32_dgrad: null pointer");
  const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
  // a 1x1 / stride-2 layer with an even input grid: only the (0, 0) parity class has a tap; its launch also writes the zeros of the other three
  const bool one_launch = stride == 2 && R == 1 && S == 1 && pad == 0 && H % 2 == 0 && W % 2 == 0;
  // every other strided layer whose chunks lie inside one tap: its parity classes as ONE grid (conv_f32_act_classes_kernel).  LEC_DGRAD_CLASSES=0: one launch per class
  const int dg_classes = tuning().dgrad_classes;
  const bool merged = dg_classes && stride == 2 && !one_launch && Cout % kCfBK == 0;
  ActGeoSet gs; int ncls = 0;
  for (int ph = 0; ph < (one_launch ? 1 : stride); ++ph) {
    for (int pw = 0; pw < (one_launch ? 1 : stride); ++pw) {
      ActGeo g;
      g.zfill = one_launch ? 1 : 0;
      g.Hm = (H - ph + stride - 1) / stride; g.Wm = (W - pw + stride - 1) / stride;
      if (g.Hm <= 0 || g.Wm <= 0) continue;
      g.Mg = N * g.Hm * g.Wm; g.Hs = Ho; g.Ws = Wo; g.Cs = Cout; g.lgCs = ilog2_exact(Cout); g.sst = 1;
      g.r0 = (ph + pad) % stride; g.s0 = (pw + pad) % stride; g.rstep = stride; g.sstep = stride;
      g.na = g.r0 < R ? (R - g.r0 + stride - 1) / stride : 0; g.nb = g.s0 < S ? (S - g.s0 + stride - 1) / stride : 0;
      g.oh0 = (ph + pad - g.r0) / stride; g.ow0 = (pw + pad - g.s0) / stride; g.sg = -1;
      g.S = S; g.RS = R * S; g.Cd = Cin; g.Cin = Cin; g.Hd = H; g.Wd = W; g.dst_st = stride; g.dph = ph; g.dpw = pw;
      g.Kg = g.na * g.nb * Cout;
      g.src_bytes = (uint32_t)((int64_t)N * Ho * Wo * Cout * 4); g.wgt_bytes = (uint32_t)((int64_t)Cout * R * S * Cin * 4);
      g.dst_bytes = (uint32_t)((int64_t)N * H * W * Cin * 4);
      g.dWm = make_fastdiv(g.Wm); g.dHm = make_fastdiv(g.Hm); g.dnb = make_fastdiv(g.nb);   // (Kg = 0: the loop is empty, zeros are stored)
      if (merged) { g.xcd_per = 0; gs.g[ncls++] = g; continue; }
      if (int rc = launch_act<false, false>(dy, w, dx, g, nullptr, nullptr, (hipStream_t)stream, ActFuse{}, kCfMaxPart, schedule)) return rc;
    }
  }
  if (merged && ncls > 0) {
    for (int a = 1; a < ncls; ++a)                              // longest K first (insertion sort of at most four)
      for (int b = a; b > 0 && gs.g[b].Kg > gs.g[b - 1].Kg; --b) { const ActGeo t = gs.g[b]; gs.g[b] = gs.g[b - 1]; gs.g[b - 1] = t; }
    for (int a = ncls; a < 4; ++a) { gs.g[a] = gs.g[0]; gs.g[a].Mg = 0; }
    const bool narrow = Cin <= 64;
    const int BM = 128, BN = narrow ? 64 : 128;
    const int ntiles = (Cin + BN - 1) / BN;
    int gx = 1;
    for (int a = 0; a < ncls; ++a) { LEC_CHECK_ARG(gs.g[a].na * gs.g[a].nb <= 32, "conv_f32_dgrad: more than 32 taps per class"); const int mt = (gs.g[a].Mg + BM - 1) / BM; if (mt > gx) gx = mt; }
    const int cap = 2048 / (ntiles > 8 ? 8 : ntiles);
    if (gx > cap) gx = cap;
    const size_t lds = (size_t)2 * (BM * kCfLdk + kCfBK * BN) * 4;
    if (narrow) hipLaunchKernelGGL((conv_f32_act_classes_kernel<1>), dim3(gx, ntiles, ncls), dim3(kCfThreads), lds, (hipStream_t)stream, dy, w, dx, gs);
    else hipLaunchKernelGGL((conv_f32_act_classes_kernel<2>), dim3(gx, ntiles, ncls), dim3(kCfThreads), lds, (hipStream_t)stream, dy, w, dx, gs);
    LEC_CHECK_LAUNCH("conv_f32_act_classes_kernel");
  }
  return LEC_OK;
}

// Data gradient of a STRIDE-1 layer with BatchNorm pieces fused in (see ActFuse):
//   xsrc / coef  (both or neither; 1x1, pad 0 only): dy is not materialised -- the kernel forms it from g = `dy` and the BatchNorm input
//                xsrc with the per-channel coefficients coef[3][Cout] (lec_bn_bwd_coeffs_f32) while loading;
//   xbn .. partials (all or none): the result is not dx but g = mask * (dx + dres), and the partial sums of pass 1 of the backward of
//                the BatchNorm whose output this layer consumed are left in `partials` (n_partials rows of [2][Cin]).
static int conv_f32_dgrad_fused_one(const float* dy, const float* w, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                                   float* dx, const float* xsrc, const float* coef, const float* dres, const float* xbn, const uint8_t* mask,
                                   const float* mean, const float* invstd, float* partials, int64_t partials_bytes, int* n_partials,
                                   int schedule, lec_stream_t stream) {
  using namespace lec;
  if (int rc = conv_check("conv_f32_dgrad_fused", N, H, W, Cin, Cout, R, S, stride, pad)) return rc;
  LEC_CHECK_ARG(dy && w && dx, "conv_f32_dgrad_fused: null pointer");
  LEC_CHECK_ARG(stride == 1, "conv_f32_dgrad_fused: stride-1 layers only (a strided data gradient is several launches)");
  const bool xf = xsrc || coef, fold = xbn || mean || invstd || partials;
  LEC_CHECK_ARG(xf || fold, "conv_f32_dgrad_fused: nothing to fuse (use lec_conv_f32_dgrad)");
  LEC_CHECK_ARG(!xf || (xsrc && coef && R == 1 && S == 1 && pad == 0), "conv_f32_dgrad_fused: the on-load form needs xsrc, coef and a 1x1 / pad 0 layer");
  LEC_CHECK_ARG(!fold || (xbn && mean && invstd && partials && n_partials && Cin % 8 == 0), "conv_f32_dgrad_fused: the fold needs xbn, mean, invstd, partials, n_partials (Cin %% 8 == 0)");
  LEC_CHECK_ARG(!fold || partials_bytes >= (int64_t)kCfMaxPart * 2 * Cin * (int64_t)sizeof(float), "conv_f32_dgrad_fused: partials buffer too small");
  LEC_CHECK_ARG(Cout % kCfBK == 0, "conv_f32_dgrad_fused: Cout must be a multiple of %d", kCfBK);
  const int Ho = H + 2 * pad - R + 1, Wo = W + 2 * pad - S + 1;
  ActGeo g;
  g.zfill = 0;
  g.Hm = H; g.Wm = W; g.Mg = N * H * W; g.Hs = Ho; g.Ws = Wo; g.Cs = Cout; g.lgCs = ilog2_exact(Cout); g.sst = 1;
  g.r0 = 0; g.s0 = 0; g.rstep = 1; g.sstep = 1; g.na = R; g.nb = S; g.oh0 = pad; g.ow0 = pad; g.sg = -1;   // lec_conv_f32_dgrad's formulas at stride 1
  g.S = S; g.RS = R * S; g.Cd = Cin; g.Cin = Cin; g.Hd = H; g.Wd = W; g.dst_st = 1; g.dph = 0; g.dpw = 0;
  g.Kg = R * S * Cout;
  g.src_bytes = (uint32_t)((int64_t)N * Ho * Wo * Cout * 4); g.wgt_bytes = (uint32_t)((int64_t)Cout * R * S * Cin * 4);
  g.dst_bytes = (uint32_t)((int64_t)N * H * W * Cin * 4);
  g.dWm = make_fastdiv(g.Wm); g.dHm = make_fastdiv(g.Hm); g.dnb = make_fastdiv(g.nb);
  ActFuse fz{};
  fz.xsrc = xsrc; fz.coef = coef; fz.dres = dres; fz.xbn = xbn; fz.mask = mask; fz.mean = mean; fz.invstd = invstd;
  fz.mask_bytes = (uint32_t)((int64_t)N * H * W * (Cin / 8));
  hipStream_t st = (hipStream_t)stream;
  const int64_t rows = fold ? partials_bytes / ((int64_t)2 * Cin * (int64_t)sizeof(float)) : 0;
  if (xf && fold) return launch_act<false, false, 3>(dy, w, dx, g, partials, n_partials, st, fz);
  if (fold) return launch_act<false, false, 2>(dy, w, dx, g, partials, n_partials, st, fz, (int)(rows < kCfMaxRows ? rows : kCfMaxRows), schedule);
  return launch_act<false, false, 1>(dy, w, dx, g, nullptr, nullptr, st, fz);
}

extern "C" int lec_conv_f32_wgrad(const float* dy, const float* x, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                                  float* dw, lec_stream_t stream) {
  return lec_conv_f32_wgrad_fused(dy, x, N, H, W, Cin, Cout, R, S, stride, pad, dw, nullptr, nullptr, stream);
}

// Weight gradient; xsrc / coef (both or neither; 1x1 / stride 1 / pad 0 layers): dy is formed on load from g = `dy`, the BatchNorm input
// xsrc and coef[3][Cout] exactly as in lec_conv_f32_dgrad_fused, so pass 2 of that BatchNorm's backward never runs as a kernel.
static int conv_f32_wgrad_impl(const float* dy, const float* x, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                               float* dw, const float* xsrc, const float* coef, int dCin, lec_stream_t stream);

extern "C" int lec_conv_f32_wgrad_fused(const float* dy, const float* x, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                                        float* dw, const float* xsrc, const float* coef, lec_stream_t stream) {
  return conv_f32_wgrad_impl(dy, x, N, H, W, Cin, Cout, R, S, stride, pad, dw, xsrc, coef, Cin, stream);
}

// The stem (3 input channels): x is [N, H, W, 4] with a zero 4th channel, dw is the layer's own [Cout][R][S][3] gradient slot -- added
// into with float atomics like every other weight gradient (several backward passes of a step may run concurrently).
extern "C" int lec_conv_f32_wgrad_c3(const float* dy, const float* x4, int N, int H, int W, int Cout, int R, int S, int stride, int pad,
                                     float* dw3, lec_stream_t stream) {
  if (Cout == 64 && R == 7 && S == 7 && stride == 2 && pad == 3 && dy && x4 && dw3 && lec_conv_f32_stem_supported(N, H, W))
    return lec::conv_f32_stem_wgrad_launch(dy, x4, N, H, W, dw3, (void*)stream);      // torchvision's stem at the image sizes its own kernel serves
  return conv_f32_wgrad_impl(dy, x4, N, H, W, 4, Cout, R, S, stride, pad, dw3, nullptr, nullptr, 3, stream);
}

static int conv_f32_wgrad_one(const float* dy, const float* x, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                              float* dw, const float* xsrc, const float* coef, int dCin, lec_stream_t stream) {
  using namespace lec;
  if (int rc = conv_check("conv_f32_wgrad", N, H, W, Cin, Cout, R, S, stride, pad)) return rc;
  LEC_CHECK_ARG(dy && x && dw, "conv_f32_wgrad: null pointer");
  const bool xf = xsrc || coef;
  LEC_CHECK_ARG(!xf || (xsrc && coef && R == 1 && S == 1 && stride == 1 && pad == 0), "conv_f32_wgrad_fused: the on-load form needs xsrc, coef and a 1x1 / stride 1 / pad 0 layer");
  WgGeo g;
  g.Ho = (H + 2 * pad - R) / stride + 1; g.Wo = (W + 2 * pad - S) / stride + 1; g.Mpix = N * g.Ho * g.Wo;
  g.H = H; g.W = W; g.Cin = Cin; g.lgCin = ilog2_exact(Cin); g.Cout = Cout; g.S = S; g.RS = R * S; g.stride = stride; g.pad = pad;
  g.Ng = R * S * Cin; g.dCin = dCin;
  g.dy_bytes = (uint32_t)((int64_t)g.Mpix * Cout * 4); g.x_bytes = (uint32_t)((int64_t)N * H * W * Cin * 4);
  g.dWo = make_fastdiv(g.Wo); g.dHo = make_fastdiv(g.Ho); g.dS = make_fastdiv(S);
  g.HoWo = g.Ho * g.Wo; g.dHW = make_fastdiv(g.HoWo);
  const bool dense = R == 1 && S == 1 && stride == 1 && pad == 0;
  // tile (BM over Cout) x (BN over R*S*Cin), chunk width WBK: gathered layers 64 x 256; dense 1x1 layers whatever fits their shape
  int BM = 64, BN = 256, WBK = kWgBK;
  const int wg_dense_tile = tuning().wg_dense_tile;
  // dense 1x1 layers: 128 x 128 wherever it fits (round 4: 2 - 4 % under the 64 x 256 tile on all eleven shapes of ResNet-50, e.g. 256 -> 128 @56 910 -> 881 us,
  // 512 -> 128 @28 453 -> 435, 1024 -> 512 @14 811 -> 781).  LEC_WGRAD_DENSE_TILE: 0 = the round-3 rule, 2 = 128 x 256 (experiment).
  if (dense && !xf && wg_dense_tile == 2 && Cout % 128 == 0 && g.Ng % 256 == 0) { BM = 128; BN = 256; }
  else if (dense && wg_dense_tile != 0 && Cout % 128 == 0 && g.Ng % 128 == 0) { BM = 128; BN = 128; }
  else if (dense) {
    if (g.Ng >= 256) { BM = 64; BN = 256; }
    else if (g.Ng >= 128) { BM = 128; BN = 128; }
    else if (Cout >= 256) { BM = 256; BN = 64; }
    else { BM = 64; BN = 64; WBK = 32; }
  }
  // Gathered layers whose output channels are a multiple of 128: a 128 x 256 tile where it pays -- the gathered operand's decode, bounds
  // tests, loads and LDS traffic are per B tile, so twice the rows halve them per MFMA (and with 128 output channels x is gathered once
  // instead of twice).  Same box, 512 images: 3x3 128 -> 128 @28 1391 -> 1075 us, stride-2 @56 1398 -> 1078; 3x3 256 -> 256 @14 1240 -> 1216;
  // strided 1x1 (downsample) 920 -> 870 / 898 -> 869; 3x3 512 -> 512 @7 1126 -> 1154 (worse: left on the 64-row tile).  223 VGPRs instead
  // of 128: beside this variant no BatchNorm wave fits a SIMD, so the step gains less than the kernels (144.4 -> 144.0 ms).
  // LEC_WGRAD_BM128 = 0: off, 2: every eligible layer.
  const int wg_bm128 = tuning().wg_bm128;
  // Shifted-dense form (see the kernel): stride-1 layers whose output grid is their input grid, tile inside one tap.  LEC_WGRAD_SHIFT=0: off.
  const int wg_shift = tuning().wg_shift;
  const bool same1 = stride == 1 && g.Ho == H && g.Wo == W && R * S > 1, half2 = stride == 2 && H == 2 * g.Ho && W == 2 * g.Wo;
  const int wg_shift64 = tuning().wg_shift64;       // the 64 -> 64 3x3 layers on a 64 x 64 tile: 1376 - 1412 -> 1103 - 1122 us @56 (library 1290 - 1305)
  const bool shifted = wg_shift && !dense && !xf && (same1 || (half2 && wg_shift != 3)) && R * S <= 16 && (Cin >= 128 || (wg_shift64 && Cin == 64 && Cout == 64 && stride == 1)) && dCin == Cin
                       && g.HoWo >= 32 && g.Wo >= 2 && Cout % 64 == 0 && (Cout % 128 == 0 || Cin >= 256 || Cin == 64);      // (the column tile must lie inside one tap)
  if (shifted) {
    if (Cin == 64) { BM = 64; BN = 64; WBK = 32; }
    else if (Cout % 128 == 0) { BM = 128; BN = 128; } else { BM = 64; BN = 256; }     // 128 x 128: 928 / 904 / 902 us on the 3x3 layers @28 / 14 / 7 against 957 / 1104 / 1133 (64 x 256), 952 / 934 / 930 (128 x 256)
  }
  const bool big = !shifted && !dense && Cout % 128 == 0 && (wg_bm128 == 2 || (wg_bm128 == 1 && (R * S == 1 || Cout <= 256)));
  if (big) BM = 128;
  const int tiles = ((Cout + BM - 1) / BM) * ((g.Ng + BN - 1) / BN);
  const int nchunks = (g.Mpix + WBK - 1) / WBK;
  const int wg_target = tuning().wg_items;
  // K split: work items = tiles x split.  The items of a launch should fill the resident workgroup slots a WHOLE number of times: with
  // split = ceil(target / tiles) (round 2) 18 tiles gave 18 x 57 = 1 026 items for 1 024 slots -- two items alone in a third round, a fifth
  // of the launch -- and 5 tiles 1 025.  Round DOWN (LEC_WGRAD_SPLIT_FLOOR=0: the old rule): at most `target` items, and `target` is a
  // multiple of the slots of every variant (512 for the 223-register 128 x 256 tile, 1 024 for the others).
  const int wg_floor = tuning().wg_split_floor;
  int split = wg_floor ? wg_target / tiles : (wg_target + tiles - 1) / tiles;
  if (split > nchunks) split = nchunks;
  if (split < 1) split = 1;
  g.chunks_per_split = (nchunks + split - 1) / split;
  split = (nchunks + g.chunks_per_split - 1) / g.chunks_per_split;
  g.tiles = tiles; g.split = split;
  size_t lds = (size_t)2 * WBK * (BM + BN) * 4 + (shifted ? (size_t)(((g.HoWo + WBK + 1) * 2 + (g.Wo + WBK) * 4 + 15) / 16 * 16) : 0);
  // Residency knob (LEC_WGRAD_LDS_PAD = extra LDS bytes): the tiles need 32 - 40 KB, so four workgroups share a CU and the main stream's
  // HBM-bound BatchNorm kernels (this kernel runs on the side stream) find no registers to land on.  14336 extra bytes keep it at two workgroups
  // per CU: BatchNorm 61.6 -> 52.2 ms inside the step, these kernels 44 -> 47.5 ms, the step 157.5 -> 155.8 ms (same-box A/B, 1 %); left off by
  // default: within the box-to-box spread, and it stretches every convolution's in-step duration.
  const int wg_lds_pad = tuning().wg_lds_pad;
  if (wg_lds_pad > 0) lds += (size_t)wg_lds_pad;
  const int wg_cap = tuning().wg_wgs;
  const int total = 8 * ((tiles * split + 7) / 8);              // slots (see the kernel's slot -> item map)
  const dim3 grid(total < wg_cap ? total : wg_cap), blk(kCfThreads);
  hipStream_t st = (hipStream_t)stream;
  const float* nof = nullptr;
  // scalar bounds masks: a 256-column tile must span at most 4 taps (Cin >= 64; the stem's 4 channels keep the vector test)
  const int wg_sm = tuning().wg_smask;
  const bool sm = !dense && Cin >= 64 && wg_sm != 0;
  if (xf) {
    LEC_CHECK_ARG(dense, "conv_f32_wgrad_fused: dense layers only");
    if (BM == 64 && BN == 256) hipLaunchKernelGGL((conv_f32_wgrad_kernel<1, 4, 2, 2, kWgBK, true, true>), grid, blk, lds, st, dy, x, dw, g, xsrc, coef);
    else if (BM == 128) hipLaunchKernelGGL((conv_f32_wgrad_kernel<2, 2, 2, 2, kWgBK, true, true>), grid, blk, lds, st, dy, x, dw, g, xsrc, coef);
    else if (BM == 256) hipLaunchKernelGGL((conv_f32_wgrad_kernel<4, 1, 2, 2, kWgBK, true, true>), grid, blk, lds, st, dy, x, dw, g, xsrc, coef);
    else hipLaunchKernelGGL((conv_f32_wgrad_kernel<2, 2, 1, 1, 32, true, true>), grid, blk, lds, st, dy, x, dw, g, xsrc, coef);
  }
  else if (shifted && stride == 2 && BM == 128) hipLaunchKernelGGL((conv_f32_wgrad_kernel<2, 2, 2, 2, kWgBK, true, false, false, 2>), grid, blk, lds, st, dy, x, dw, g, nof, nof);
  else if (shifted && stride == 2) hipLaunchKernelGGL((conv_f32_wgrad_kernel<1, 4, 2, 2, kWgBK, true, false, false, 2>), grid, blk, lds, st, dy, x, dw, g, nof, nof);
  else if (shifted && BN == 64) hipLaunchKernelGGL((conv_f32_wgrad_kernel<2, 2, 1, 1, 32, true, false, false, 1>), grid, blk, lds, st, dy, x, dw, g, nof, nof);
  else if (shifted && BM == 128) hipLaunchKernelGGL((conv_f32_wgrad_kernel<2, 2, 2, 2, kWgBK, true, false, false, 1>), grid, blk, lds, st, dy, x, dw, g, nof, nof);
  else if (shifted) hipLaunchKernelGGL((conv_f32_wgrad_kernel<1, 4, 2, 2, kWgBK, true, false, false, 1>), grid, blk, lds, st, dy, x, dw, g, nof, nof);
  else if (big && sm) hipLaunchKernelGGL((conv_f32_wgrad_kernel<2, 2, 2, 4, kWgBK, false, false, true>), grid, blk, lds, st, dy, x, dw, g, nof, nof);
  else if (big) hipLaunchKernelGGL((conv_f32_wgrad_kernel<2, 2, 2, 4, kWgBK, false>), grid, blk, lds, st, dy, x, dw, g, nof, nof);
  else if (!dense && sm) hipLaunchKernelGGL((conv_f32_wgrad_kernel<1, 4, 2, 2, kWgBK, false, false, true>), grid, blk, lds, st, dy, x, dw, g, nof, nof);
  else if (!dense) hipLaunchKernelGGL((conv_f32_wgrad_kernel<1, 4, 2, 2, kWgBK, false>), grid, blk, lds, st, dy, x, dw, g, nof, nof);
  else if (BM == 128 && BN == 256) hipLaunchKernelGGL((conv_f32_wgrad_kernel<2, 2, 2, 4, kWgBK, true>), grid, blk, lds, st, dy, x, dw, g, nof, nof);
  else if (BM == 64 && BN == 256) hipLaunchKernelGGL((conv_f32_wgrad_kernel<1, 4, 2, 2, kWgBK, true>), grid, blk, lds, st, dy, x, dw, g, nof, nof);
  else if (BM == 128) hipLaunchKernelGGL((conv_f32_wgrad_kernel<2, 2, 2, 2, kWgBK, true>), grid, blk, lds, st, dy, x, dw, g, nof, nof);
  else if (BM == 256) hipLaunchKernelGGL((conv_f32_wgrad_kernel<4, 1, 2, 2, kWgBK, true>), grid, blk, lds, st, dy, x, dw, g, nof, nof);
  else hipLaunchKernelGGL((conv_f32_wgrad_kernel<2, 2, 1, 1, 32, true>), grid, blk, lds, st, dy, x, dw, g, nof, nof);
  LEC_CHECK_LAUNCH("conv_f32_wgrad_kernel");
  return LEC_OK;
}

// ---- exported entry points: one launch, or one per group of images when a tensor of the whole batch would reach 2 GiB
#define LEC_CONV_GROUPS(who, in_c, out_c)                                                                                         \
  const int64_t Ho_ = (H + 2 * pad - R) / stride + 1, Wo_ = (W + 2 * pad - S) / stride + 1;                                       \
  const int64_t in_img = (int64_t)H * W * (in_c) * 4, out_img = Ho_ * Wo_ * (int64_t)(out_c) * 4;                                 \
  const int G_ = N > 0 && H > 0 && W > 0 && Ho_ > 0 && Wo_ > 0 ? lec::conv_group_images(N, in_img, out_img) : N;                  \
  LEC_CHECK_ARG(G_ > 0 || N <= 0, who ": one image of this layer reaches 2 GiB")

extern "C" int lec_conv_f32_fwd(const float* x, const float* w, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                                float* y, float* partials, int64_t partials_bytes, int* n_partials, int schedule, lec_stream_t stream) {
  LEC_CONV_GROUPS("conv_f32_fwd", Cin, Cout);
  if (G_ >= N) return conv_f32_fwd_one(x, w, N, H, W, Cin, Cout, R, S, stride, pad, y, partials, partials_bytes, n_partials, schedule, stream);
  int rows = 0;
  for (int n0 = 0; n0 < N; n0 += G_) {
    const int n = N - n0 < G_ ? N - n0 : G_;
    int k = 0;
    const int64_t used = (int64_t)rows * 2 * Cout * (int64_t)sizeof(float);
    if (int rc = conv_f32_fwd_one(x + n0 * (in_img / 4), w, n, H, W, Cin, Cout, R, S, stride, pad, y + n0 * (out_img / 4),
                                  partials ? partials + (int64_t)rows * 2 * Cout : nullptr, partials ? partials_bytes - used : 0, partials ? &k : nullptr, schedule, stream)) return rc;
    rows += k;
  }
  if (n_partials) *n_partials = rows;
  return LEC_OK;
}

extern "C" int lec_conv_f32_fwd_affine(const float* x, const float* w, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                                       float* y, const float* scale, const float* shift, const float* res, int relu, int schedule, lec_stream_t stream) {
  LEC_CONV_GROUPS("conv_f32_fwd_affine", Cin, Cout);
  for (int n0 = 0; n0 < N; n0 += (G_ > 0 ? G_ : N)) {
    const int n = N - n0 < G_ ? N - n0 : G_;
    if (int rc = conv_f32_fwd_affine_one(x + n0 * (in_img / 4), w, n, H, W, Cin, Cout, R, S, stride, pad, y + n0 * (out_img / 4), scale, shift,
                                         res ? res + n0 * (out_img / 4) : nullptr, relu, schedule, stream)) return rc;
  }
  return N > 0 ? LEC_OK : conv_f32_fwd_affine_one(x, w, N, H, W, Cin, Cout, R, S, stride, pad, y, scale, shift, res, relu, schedule, stream);
}

extern "C" int lec_conv_f32_dgrad(const float* dy, const float* w, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                                  float* dx, int schedule, lec_stream_t stream) {
  LEC_CONV_GROUPS("conv_f32_dgrad", Cin, Cout);
  for (int n0 = 0; n0 < N; n0 += (G_ > 0 ? G_ : N)) {
    const int n = N - n0 < G_ ? N - n0 : G_;
    if (int rc = conv_f32_dgrad_one(dy + n0 * (out_img / 4), w, n, H, W, Cin, Cout, R, S, stride, pad, dx + n0 * (in_img / 4), schedule, stream)) return rc;
  }
  return N > 0 ? LEC_OK : conv_f32_dgrad_one(dy, w, N, H, W, Cin, Cout, R, S, stride, pad, dx, schedule, stream);
}

extern "C" int lec_conv_f32_dgrad_fused(const float* dy, const float* w, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                                        float* dx, const float* xsrc, const float* coef, const float* dres, const float* xbn, const uint8_t* mask,
                                        const float* mean, const float* invstd, float* partials, int64_t partials_bytes, int* n_partials,
                                        int schedule, lec_stream_t stream) {
  LEC_CONV_GROUPS("conv_f32_dgrad_fused", Cin, Cout);
  if (G_ >= N) return conv_f32_dgrad_fused_one(dy, w, N, H, W, Cin, Cout, R, S, stride, pad, dx, xsrc, coef, dres, xbn, mask, mean, invstd, partials, partials_bytes, n_partials, schedule, stream);
  int rows = 0;
  const int64_t mask_img = (int64_t)H * W * (Cin / 8);
  for (int n0 = 0; n0 < N; n0 += G_) {
    const int n = N - n0 < G_ ? N - n0 : G_;
    int k = 0;
    const int64_t used = (int64_t)rows * 2 * Cin * (int64_t)sizeof(float);
    if (int rc = conv_f32_dgrad_fused_one(dy + n0 * (out_img / 4), w, n, H, W, Cin, Cout, R, S, stride, pad, dx + n0 * (in_img / 4), xsrc ? xsrc + n0 * (out_img / 4) : nullptr, coef,
                                          dres ? dres + n0 * (in_img / 4) : nullptr, xbn ? xbn + n0 * (in_img / 4) : nullptr, mask ? mask + n0 * mask_img : nullptr, mean, invstd,
                                          partials ? partials + (int64_t)rows * 2 * Cin : nullptr, partials ? partials_bytes - used : 0, partials ? &k : nullptr, schedule, stream)) return rc;
    rows += k;
  }
  if (n_partials) *n_partials = rows;
  return LEC_OK;
}

static int conv_f32_wgrad_impl(const float* dy, const float* x, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                               float* dw, const float* xsrc, const float* coef, int dCin, lec_stream_t stream) {
  LEC_CONV_GROUPS("conv_f32_wgrad", Cin, Cout);
  for (int n0 = 0; n0 < N; n0 += (G_ > 0 ? G_ : N)) {
    const int n = N - n0 < G_ ? N - n0 : G_;
    if (int rc = conv_f32_wgrad_one(dy + n0 * (out_img / 4), x + n0 * (in_img / 4), n, H, W, Cin, Cout, R, S, stride, pad, dw, xsrc ? xsrc + n0 * (out_img / 4) : nullptr, coef, dCin, stream)) return rc;
  }
  return N > 0 ? LEC_OK : conv_f32_wgrad_one(dy, x, N, H, W, Cin, Cout, R, S, stride, pad, dw, xsrc, coef, dCin, stream);
}
