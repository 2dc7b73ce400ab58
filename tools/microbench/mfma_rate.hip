// Micro-benchmark: issue rate of v_mfma_f32_32x32x16_bf16 on gfx950 with 1 / 2 / 4 independent accumulator chains per
// wave, operands in registers (no memory).   hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, long long* cycles) {
  bf16x8_t a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3f80 + threadIdx.x % 3); b[j] = (short)(0x3f80 + threadIdx.x % 5); }
  f32x16_t acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int q = 0; q < 16; ++q) acc[c][q] = 0.0f;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
  }
  long long t1 = clock64();
  float s = 0.0f;
  for (int c = 0; c < CHAINS; ++c) for (int q = 0; q < 16; ++q) s += acc[c][q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int CHAINS>
void run(int blocks, int threads) {
  float* out; long long* cyc; hipMalloc(&out, blocks * threads * 4); hipMalloc(&cyc, 8);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_loop<CHAINS>, dim3(blocks), dim3(threads), 0, 0, out, 100, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(mfma_loop<CHAINS>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n_mfma = (double)iters * CHAINS;
  const double waves = (double)blocks * threads / 64;
  printf("chains %d, %4d blocks x %3d threads: %.1f ns per MFMA per wave, %.1f clock64 ticks per MFMA, chip %.0f TFLOP/s\n", CHAINS, blocks, threads,
         ms * 1e6 / n_mfma, (double)c / n_mfma, n_mfma * waves * 32768.0 / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(cyc);
}

int main() {
  run<1>(256, 256); run<2>(256, 256); run<4>(256, 256);
  run<2>(512, 256); run<4>(512, 256); run<2>(256, 512); run<4>(1024, 256);
  return 0;
}
