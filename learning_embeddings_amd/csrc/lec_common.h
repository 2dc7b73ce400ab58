// Shared host/device helpers for liblecone.so (gfx950 only: 64-wide wavefronts are assumed everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lecone.h"

namespace lec {

void set_error(const char* fmt, ...);                      // abi.cpp (thread-local message for lec_last_error)
int hip_fail(hipError_t e, const char* what);               // records + returns LEC_E_HIP

#define LEC_CHECK_ARG(cond, ...)                                                                   \
  do { if (!(cond)) { ::lec::set_error(__VA_ARGS__); return LEC_E_ARG; } } while (0)
#define LEC_CHECK_LAUNCH(name)                                                                     \
  do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return ::lec::hip_fail(e__, name); } while (0)

// reference constants (oe_h.py:59,66,75,79,83,826-827)
__host__ __device__ inline float inner_radius_f(float K) {
  double k = (double)K;                                     // the reference evaluates this in python float64
  return (float)(2.0 * k / (1.0 + sqrt(1.0 + 4.0 * k * k)));
}
inline float inner_radius_h_f(float K) {                    // Embedder.arctanh on the float32 radius (oe_h.py:106-110)
  float x = inner_radius_f(K);
  const float lo = (float)(-1.0 + 1e-5), hi = (float)(1.0 - 1e-5);
  x = x < lo ? lo : (x > hi ? hi : x);
  return 0.5f * (logf(1.0f + x) - logf(1.0f - x));
}

#if defined(__HIPCC__)
constexpr int kWave = 64;

// DPP lane exchange inside a 16-lane row (no LDS crossbar round trip, unlike ds_bpermute behind __shfl_xor)
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

// all-reduce (sum) inside aligned groups of T lanes; every lane ends with the bit-identical total.
// Steps inside a 16-lane row are DPP butterflies: quad_perm[1,0,3,2] (xor 1), quad_perm[2,3,0,1] (xor 2), then
// row_half_mirror / row_mirror -- once every lane of a quad (of an 8-group) already holds that group's sum, the mirror
// pairs each group with its sibling exactly as xor 4 (xor 8) would.  Wider groups finish with ds_bpermute butterflies.
template <int T>
__device__ __forceinline__ float group_sum(float v) {
  if (T >= 2) v += dpp_move<0xB1>(v);
  if (T >= 4) v += dpp_move<0x4E>(v);
  if (T >= 8) v += dpp_move<0x141>(v);
  if (T >= 16) v += dpp_move<0x140>(v);
  if (T >= 32) v += __shfl_xor(v, 16, kWave);
  if (T >= 64) v += __shfl_xor(v, 32, kWave);
  return v;
}
template <int T>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int m = T >> 1; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, kWave));
  return v;
}

__device__ __forceinline__ float clampf_nanprop(float a, float lo, float hi) {   // torch.clamp keeps NaN
  return a < lo ? lo : (a > hi ? hi : a);
}

// ---------------------------------------------------------------------------------------------------------
// Hyperbolic entailment-cone energy from the four row statistics (oe_h.py:817-833), float32 in the reference's
// operation order, plus the coefficients of its gradient:
//     dE/dx = cxx * x + cxy * y ,   dE/dy = cxy * x + cyy * y
// (E depends on x, y only through |x|, |y|, |x-y| and <x,y>; autograd's chain through those four collapses to this.)
// ---------------------------------------------------------------------------------------------------------
struct ConeEval { float E, cxx, cxy, cyy; };
struct ConeFwd { float E, xn, yn, dist, xn2, yn2, s, rad, den, a, pa, ac, pc, diff; };

__device__ __forceinline__ ConeFwd cone_forward(float xx, float yy, float s, float dd, float K) {
  const float lo = (float)(-1.0 + 1e-5), hi = (float)(1.0 - 1e-5);
  ConeFwd f;
  f.xn = sqrtf(xx); f.yn = sqrtf(yy); f.dist = sqrtf(dd); f.s = s;
  f.xn2 = f.xn * f.xn; f.yn2 = f.yn * f.yn;
  float num = s * (1.0f + f.xn2) - f.xn2 * (1.0f + f.yn2);
  float xy = f.xn * f.yn;
  f.rad = 1.0f + xy * xy - 2.0f * s;
  f.den = f.xn * f.dist * sqrtf(f.rad);
  f.a = num / f.den;                                                              // oe_h.py:823
  f.pa = K * (1.0f - f.xn2) / f.xn;
  f.ac = clampf_nanprop(f.a, lo, hi); f.pc = clampf_nanprop(f.pa, lo, hi);
  f.diff = acosf(f.ac) - asinf(f.pc);                                             // :826-827
  f.E = f.diff < 0.0f ? 0.0f : f.diff;                                            // :833 (NaN stays NaN)
  return f;
}

// gradient coefficients from the forward intermediates.  Divisions here use reciprocals (1-2 ulp): the coefficients feed
// gradients whose parity bar is relative 1e-3; the energy itself keeps correctly rounded arithmetic.
__device__ __forceinline__ void cone_grad_coeffs(const ConeFwd& f, float K, float& cxx, float& cxy, float& cyy) {
  const float lo = (float)(-1.0 + 1e-5), hi = (float)(1.0 - 1e-5);
  cxx = cxy = cyy = 0.0f;
  const bool live = f.diff >= 0.0f;
  const bool a_in = live && (f.a >= lo) && (f.a <= hi);
  const bool p_in = live && (f.pa >= lo) && (f.pa <= hi);
  float g_s = 0.f, g_d = 0.f, g_xn = 0.f, g_yn = 0.f;
  const float r_xn = __frcp_rn(f.xn);
  if (a_in) {
    const float dth = -__frsqrt_rn(1.0f - f.ac * f.ac);
    const float r_den = __frcp_rn(f.den), a_rad = f.a * __frcp_rn(f.rad);
    g_s = dth * ((1.0f + f.xn2) * r_den + a_rad);
    g_d = dth * (-f.a * __frcp_rn(f.dist));
    g_xn = dth * ((2.0f * f.xn * f.s - 2.0f * f.xn * (1.0f + f.yn2)) * r_den - f.a * r_xn - a_rad * (f.xn * f.yn2));
    g_yn = dth * (-2.0f * f.xn2 * f.yn * r_den - a_rad * (f.xn2 * f.yn));
  }
  if (p_in) {
    const float dps = __frsqrt_rn(1.0f - f.pc * f.pc);
    g_xn += dps * (K * (1.0f + f.xn2) * (r_xn * r_xn));                            // -dpsi/dxn, dpa/dxn = -K(1+xn^2)/xn^2
  }
  if (a_in || p_in) {
    const float gd_over = a_in ? g_d * __frcp_rn(f.dist) : 0.0f;
    cxx = g_xn * r_xn + gd_over;
    cxy = g_s - gd_over;
    cyy = (a_in ? g_yn * __frcp_rn(f.yn) : 0.0f) + gd_over;
  }
}

template <bool GRAD>
__device__ __forceinline__ ConeEval cone_eval(float xx, float yy, float s, float dd, float K) {
  ConeFwd f = cone_forward(xx, yy, s, dd, K);
  ConeEval r; r.E = f.E; r.cxx = r.cxy = r.cyy = 0.0f;
  if (GRAD) cone_grad_coeffs(f, K, r.cxx, r.cxy, r.cyy);
  return r;
}

// ---------------------------------------------------------------------------------------------------------
// Euclidean entailment cone (network/oe.py:721-739) from |x|^2, |y-x|^2 and u = <x, y-x>:
//     theta = -u / (max(|x|,eps) max(|y-x|,eps))     (the two F.normalize calls, eps = 1e-12)
//     psi   = -sqrt(1 - K^2/|x|^2) ,  E = max(theta - psi, 0)
// Its gradient has the same shape as the hyperbolic one (E depends on x, y through |x|, |y-x|, u only):
//     dE/dx = cxx x + cxy y ,  dE/dy = cxy x + cyy y .
// ---------------------------------------------------------------------------------------------------------
template <bool GRAD>
__device__ __forceinline__ ConeEval euc_cone_eval(float xx, float dd, float u, float K) {
  const float eps = 1e-12f;
  const float xn = sqrtf(xx), dn = sqrtf(dd);
  const float xc = fmaxf(xn, eps), dc = fmaxf(dn, eps);
  const float KK = (float)((double)K * (double)K);          // python float K*K, then a float32 tensor op (oe.py:737)
  const float theta = -(u / (xc * dc));
  const float psi = -sqrtf(1.0f - KK / (xn * xn));
  const float diff = theta - psi;
  ConeEval r; r.E = diff < 0.0f ? 0.0f : diff;              // NaN stays NaN (torch.clamp)
  r.cxx = r.cxy = r.cyy = 0.0f;
  if (GRAD && diff >= 0.0f) {
    const float A = -1.0f / (xc * dc);                                              // d theta / d u
    const float Bx = (xn >= eps ? u / (xc * xc * dc) : 0.0f) - KK / (xn * xn * xn * psi);   // d theta/d|x| - d psi/d|x|
    const float cd = dn >= eps ? u / (xc * dc * dc) / dc : 0.0f;                    // (d theta / d|y-x|) / |y-x|
    r.cxx = -2.0f * A + Bx / xn + cd; r.cxy = A - cd; r.cyy = cd;
  }
  return r;
}

// ---------------------------------------------------------------------------------------------------------
// Deterministic scalar reduction across blocks (one launch): every block publishes one partial; the block whose
// ticket is last sums all partials in a fixed order and writes `out`.  Agent-scope release/acquire hand-off
// (per-CU L1s and per-XCD L2s are not coherent): publish = store -> vmcnt(0) -> release fence -> vmcnt(0) ->
// relaxed agent ticket; last arriver = acquire fence -> vmcnt(0) -> barrier -> plain loads.
// `counter` must be zero at launch: the workspace is zeroed once when it is allocated and the last arriver re-arms
// it, so launches that share a workspace must be stream-ordered (one in flight at a time).
// Must be called by ALL threads of the block (contains __syncthreads).  blockDim.x multiple of 64, <= 1024.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void block_publish_and_finalize(float wave_value /*valid in lane 0 of each wave*/,
                                                           float* partials, unsigned int* counter, float* out,
                                                           float scale) {
  __shared__ float s_wave[16];
  __shared__ int s_last;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  if (lane == 0) s_wave[wave] = wave_value;
  __syncthreads();
  if (threadIdx.x == 0) {
    float acc = 0.0f;
    for (int w = 0; w < nwave; ++w) acc += s_wave[w];
    partials[blockIdx.x] = acc;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned int prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int last = (prev == gridDim.x - 1) ? 1 : 0;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    s_last = last;
  }
  __syncthreads();
  if (s_last && wave == 0) {
    float acc = 0.0f;
    for (unsigned int i = lane; i < gridDim.x; i += 64) acc += partials[i];       // fixed order per lane
    acc = group_sum<64>(acc);                                                      // fixed butterfly
    if (lane == 0) {
      out[0] = acc * scale;
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm for the next (stream-ordered) launch
    }
  }
}

// The same reduction with ONE returning atomic per block and no partial store, release, acquire or summation loop (round 5; the ticket form above costs the fused
// loss 4.4 us of a 15.6 us wave at config 5: profiles/r05_cone_timeline.md).  A block's partial p >= 0 enters a 64-bit word as an INTEGER: bits 63..52 count the
// blocks, bits 51..0 hold llrint(p * scale) -- integer addition is associative, so the total does not depend on the order the blocks arrive in, and the block that
// finds count == gridDim.x - 1 in the value its own add returned knows the whole sum without reading anything else.  The caller picks `scale` = 2^F from an upper
// bound of the total so that 52 bits cannot overflow (the quantisation, 2^-F per block, is far below one float rounding of the result).  A partial that is NaN,
// infinite, negative or above the bound (the reference's loss is NaN / inf there too) sets bit 0 of `flag` first; the result is then NaN.
// `acc` and `flag` (8-byte aligned, consecutive) are zero at launch; the last block re-arms them.  gridDim.x < 4096.  Call with all threads of the block.
__device__ __forceinline__ void block_publish_fixed_point(float wave_value /*valid in lane 0 of each wave*/, unsigned long long* acc, unsigned int* flag,
                                                          float* out, double scale, double inv_scale, float bound) {
  __shared__ float s_wave_fx[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  if (lane == 0) s_wave_fx[wave] = wave_value;
  __syncthreads();
  if (threadIdx.x != 0) return;
  float p = 0.0f;
  for (int w = 0; w < nwave; ++w) p += s_wave_fx[w];
  const bool ok = p >= 0.0f && p <= bound;                                        // (false for NaN)
  // Ordering of the flag hand-off: hardware -- the bad block's fetch_or is a RETURNING agent-scope atomic and the wave waits for it (vmcnt(0)) before it
  // issues its ticket, and the last block's flag load is issued only after its own ticket has RETURNED (the branch below depends on the returned value; both are
  // agent-scope accesses performed at the coherence point, past the CU's L1); compiler -- the signal fences keep the three accesses in this program order (a relaxed
  // atomic may otherwise be moved across another address's atomic).  An agent-scope acq_rel on the ticket would say the same in the memory model's own words, at the
  // price of an L2 write-back + L1 invalidate in EVERY block (1.7 - 3.5 us each next to a 17 us launch): not taken.
  if (!ok) {
    __hip_atomic_fetch_or(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // returning: performed before the ticket below is issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
  }
  const unsigned long long mask = (1ull << 52) - 1ull;
  const unsigned long long add = (1ull << 52) | (ok ? ((unsigned long long)__double2ll_rn((double)p * scale) & mask) : 0ull);
  const unsigned long long prev = __hip_atomic_fetch_add(acc, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if ((prev >> 52) == (unsigned long long)(gridDim.x - 1)) {
    const unsigned long long total = (prev & mask) + (add & mask);
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    const unsigned int bad = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    out[0] = bad ? __builtin_nanf("") : (float)((double)total * inv_scale);
    __hip_atomic_store(acc, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // re-arm for the next (stream-ordered) launch
    if (bad) __hip_atomic_store(flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
#endif  // __HIPCC__

}  // namespace lec
