// Microbenchmark (round 5): what does it cost to finalize BatchNorm statistics in the TAIL of the kernel that produced the partial rows -- a two-level,
// fixed-order reduction behind agent-scope tickets -- instead of in a separate launch?  Built as a shared library, driven by fin_tail.py next to a
// convolution running on another stream.  `stub_kernel` stands for a producer's epilogue: gx workgroups per 128-channel column block, each leaves one
// partial row [sum | sum of squares] of its 128 channels.
//   mode 0: rows only (the separate lec_bn_fwd_finalize launch follows)
//   mode 1: + tail: groups of 16 workgroups -> the group's last arriver sums its 16 rows (double) into a group row -> the column block's last group sums
//           the <= 32 group rows, turns them into mean / invstd / scale / shift and re-arms the counters.
#include <hip/hip_runtime.h>
#include <stdint.h>

struct TailArgs {
  float* part;            // [gx][2][C]
  double* grows;          // [ntiles][32][2][128]
  unsigned* cnt_group;    // [ntiles][32]
  unsigned* cnt_tile;     // [ntiles]
  float* out;             // [4][C]: mean, invstd, scale, shift
  const float* gamma; const float* beta;
  int gx, C, mode; float count;
};

__device__ __forceinline__ bool last_arriver(unsigned* counter, unsigned expected) {
  __shared__ int s_last;
  __syncthreads();
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = prev == expected - 1;
    if (last) { __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
    s_last = last;
  }
  __syncthreads();
  return s_last != 0;
}

extern "C" __global__ __launch_bounds__(256) void stub_kernel(TailArgs a) {
  const int nt = blockIdx.x / a.gx, bx = blockIdx.x % a.gx;
  const int tid = threadIdx.x, stat = tid >> 7, ch = tid & 127;
  const int c = nt * 128 + ch;
  // the producer's partial row (any deterministic value)
  const float v = stat == 0 ? 0.001f * (float)((bx * 131 + c * 7) % 1000 - 500) : 1.0f + 0.001f * (float)((bx * 17 + c) % 1000);
  __builtin_nontemporal_store(v, a.part + ((int64_t)bx * 2 + stat) * a.C + c);
  if (a.mode == 0) return;
  const int G = 16, ng = (a.gx + G - 1) / G, g = bx / G;
  const int gsize = (g == ng - 1) ? a.gx - g * G : G;
  if (!last_arriver(a.cnt_group + nt * 32 + g, (unsigned)gsize)) return;
  // level 1: this group's rows, fixed order, one batch of loads
  float r[16];
#pragma unroll
  for (int i = 0; i < G; ++i) r[i] = i < gsize ? __builtin_nontemporal_load(a.part + ((int64_t)(g * G + i) * 2 + stat) * a.C + c) : 0.0f;
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < G; ++i) s += (double)r[i];
  double* grow = a.grows + (((int64_t)nt * 32 + g) * 2 + stat) * 128 + ch;
  __builtin_nontemporal_store(s, grow);
  if (!last_arriver(a.cnt_tile + nt, (unsigned)ng)) return;
  // level 2: <= 32 group rows -> the column block's statistics
  __shared__ double sh[2][128];
  double rr[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) rr[i] = i < ng ? __builtin_nontemporal_load(a.grows + (((int64_t)nt * 32 + i) * 2 + stat) * 128 + ch) : 0.0;
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < 32; ++i) t += rr[i];
  sh[stat][ch] = t;
  __syncthreads();
  if (stat == 0) {
    const double mean = sh[0][ch] / (double)a.count;
    double var = sh[1][ch] / (double)a.count - mean * mean; if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + 1e-5));
    const float sc = a.gamma[c] * invstd;
    a.out[c] = (float)mean; a.out[a.C + c] = invstd; a.out[2 * a.C + c] = sc; a.out[3 * a.C + c] = a.beta[c] - (float)mean * sc;
  }
}

extern "C" int fin_tail_launch(float* part, double* grows, unsigned* cnt_group, unsigned* cnt_tile, float* out, const float* gamma, const float* beta,
                               int gx, int C, int mode, float count, void* stream) {
  TailArgs a{part, grows, cnt_group, cnt_tile, out, gamma, beta, gx, C, mode, count};
  hipLaunchKernelGGL(stub_kernel, dim3(gx * (C / 128)), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
