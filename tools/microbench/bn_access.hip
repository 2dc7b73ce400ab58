// Does the lane -> address pattern of a thread's two 16-byte accesses matter for a 2-read / 1-write streaming pass (the BatchNorm backward apply)?
//   A: lane i owns bytes [32 i, 32 i + 32): two loads at a 32-byte lane stride (the fp32 BatchNorm kernels' EF32 layout)
//   B: lane i owns [16 i, 16 i + 16) and [half + 16 i, half + 16 i + 16): each wave-instruction is 1 KiB contiguous
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench/bn_access.hip -o tools/microbench/bn_access
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ o, long rows, int CV) {
  // a row = CV 32-byte vectors = 2 CV f4; thread (cv, rg) as in bn.hip
  const int cv = threadIdx.x % CV, rg = threadIdx.x / CV, RPI = 256 / CV;
  for (long r = (long)blockIdx.x * RPI + rg; r < rows; r += (long)gridDim.x * RPI) {
    const long base = r * 2 * CV;
    long i0, i1;
    if (MODE == 0) { i0 = base + 2 * cv; i1 = i0 + 1; } else { i0 = base + cv; i1 = base + CV + cv; }
    const f4 x0 = __builtin_nontemporal_load(a + i0), x1 = __builtin_nontemporal_load(a + i1);
    const f4 y0 = __builtin_nontemporal_load(b + i0), y1 = __builtin_nontemporal_load(b + i1);
    __builtin_nontemporal_store(x0 * 1.5f + y0, o + i0); __builtin_nontemporal_store(x1 * 1.5f + y1, o + i1);
  }
}
int main() {
  const int C = 256, CV = C / 8; const long rows = 512L * 56 * 56; const size_t n = (size_t)rows * C * 4;
  f4 *a, *b, *o; hipMalloc(&a, n); hipMalloc(&b, n); hipMalloc(&o, n); hipMemset(a, 0, n); hipMemset(b, 0, n);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) for (int blocks : {1024, 2048, 4096}) {
    for (int it = 0; it < 3; ++it) { if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, a, b, o, rows, CV); else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, a, b, o, rows, CV); }
    hipEventRecord(e0);
    for (int it = 0; it < 10; ++it) { if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, a, b, o, rows, CV); else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, a, b, o, rows, CV); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %c blocks %d: %.1f us  %.2f TB/s\n", mode ? 'B' : 'A', blocks, ms * 100, 3.0 * n / (ms / 10 * 1e-3) / 1e12);
  }
  return 0;
}
