// What a streaming pass can reach on this GPU by mix of streams (per thread: two contiguous-per-wave 16-byte accesses per array and row, as bn.hip's EF32):
// 1 read; 2 reads; 2 reads + 1 write; 1 read + 1 write -- each with plain and non-temporal accesses, and 1 / 2 / 4 rows in flight per thread.
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench/stream_mix.hip -o tools/microbench/stream_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ f4 ld(const f4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(f4* p, f4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
template <int NR, int NW, bool NT, int U>
__global__ __launch_bounds__(256) void k(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ o, float* __restrict__ sink, long rows, int CV) {
  const int cv = threadIdx.x % CV, rg = threadIdx.x / CV, RPI = 256 / CV;
  const long stride = (long)gridDim.x * RPI;
  f4 acc = {0, 0, 0, 0};
  for (long r0 = (long)blockIdx.x * RPI + rg; r0 < rows; r0 += U * stride) {
    f4 x0[U], x1[U], y0[U], y1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long r = r0 + u * stride; const bool ok = r < rows; const long i0 = (ok ? r : 0) * 2 * CV + cv, i1 = i0 + CV;
      x0[u] = ld<NT>(a + i0); x1[u] = ld<NT>(a + i1);
      if (NR > 1) { y0[u] = ld<NT>(b + i0); y1[u] = ld<NT>(b + i1); }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long r = r0 + u * stride; if (r >= rows) continue; const long i0 = r * 2 * CV + cv, i1 = i0 + CV;
      f4 v0 = x0[u] * 1.5f, v1 = x1[u] * 1.5f;
      if (NR > 1) { v0 += y0[u]; v1 += y1[u]; }
      if (NW) { st<NT>(o + i0, v0); st<NT>(o + i1, v1); } else acc += v0 + v1;
    }
  }
  if (!NW && acc[0] == 12345.f) sink[0] = acc[1];
}
template <int NR, int NW, bool NT, int U> void run(const char* name, f4* a, f4* b, f4* o, float* sink, long rows, int CV, size_t n) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((k<NR, NW, NT, U>), dim3(2048), dim3(256), 0, 0, a, b, o, sink, rows, CV);
  (void)hipEventRecord(e0);
  for (int it = 0; it < 10; ++it) hipLaunchKernelGGL((k<NR, NW, NT, U>), dim3(2048), dim3(256), 0, 0, a, b, o, sink, rows, CV);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-8s nt=%d rows-in-flight=%d: %7.1f us  %.2f TB/s\n", name, (int)NT, U, ms * 100, (double)(NR + NW) * n / (ms / 10 * 1e-3) / 1e12);
}
int main() {
  const int C = 256, CV = C / 8; const long rows = 512L * 56 * 56; const size_t n = (size_t)rows * C * 4;
  f4 *a, *b, *o; float* sink; (void)hipMalloc(&a, n); (void)hipMalloc(&b, n); (void)hipMalloc(&o, n); (void)hipMalloc(&sink, 64); (void)hipMemset(a, 0, n); (void)hipMemset(b, 0, n);
#define ALLU(NR, NW, NT, NAME) run<NR, NW, NT, 1>(NAME, a, b, o, sink, rows, CV, n); run<NR, NW, NT, 2>(NAME, a, b, o, sink, rows, CV, n); run<NR, NW, NT, 4>(NAME, a, b, o, sink, rows, CV, n);
  ALLU(1, 0, false, "1R") ALLU(1, 0, true, "1R") ALLU(2, 0, false, "2R") ALLU(2, 0, true, "2R")
  ALLU(2, 1, false, "2R+1W") ALLU(2, 1, true, "2R+1W") ALLU(1, 1, false, "1R+1W") ALLU(1, 1, true, "1R+1W")
  return 0;
}
