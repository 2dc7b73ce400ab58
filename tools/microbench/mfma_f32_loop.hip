// Micro-benchmark: where does an LDS-fed v_mfma_f32_32x32x2_f32 loop of the conv_f32 shape (64 x 64 per wave, K chunks of 32) lose
// matrix-pipe time?  Levels: 0 registers only; 1 + fragment reads from LDS (8 ds_read_b128 per 32 MFMAs); 2 + 8 ds_write_b128 and a
// barrier per chunk; 3 + 8 global_load_dwordx4 per chunk (L2-resident source).  One or two workgroups per CU.
//   hipcc --offload-arch=gfx950 -O3 mfma_f32_loop.hip -o mfma_f32_loop && ./mfma_f32_loop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int LDK = 36;

template <int LEVEL>
__global__ __launch_bounds__(256, 2) void loop(const float* __restrict__ src, float* out, int chunks) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
  for (int i = tid; i < 2 * 2 * 128 * LDK; i += 256) smem[i] = (float)((i * 7) % 5) * 0.25f;
  __syncthreads();
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  f32x4v rg[8];
  const float* gp = src + (size_t)(blockIdx.x % 64) * 8192 + tid * 4;
  float ra[2][4] = {{1, 2, 3, 4}, {1, 2, 3, 4}}, rb[2][4] = {{1, 2, 3, 4}, {4, 3, 2, 1}};
  for (int ch = 0; ch < chunks; ++ch) {
    const int buf = ch & 1;
    const float* sA = smem + buf * 2 * 128 * LDK; const float* sB = sA + 128 * LDK;
    if (LEVEL >= 3) {
#pragma unroll
      for (int u = 0; u < 8; ++u) rg[u] = *(const f32x4v*)(gp + ((ch * 8 + u) % 8) * 1024);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (LEVEL >= 1) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const f32x4v va = *(const f32x4v*)(sA + (wm0 + it * 32 + l31) * LDK + 8 * q + 4 * h);
          const f32x4v vb = *(const f32x4v*)(sB + (wn0 + it * 32 + l31) * LDK + 8 * q + 4 * h);
          for (int t = 0; t < 4; ++t) { ra[it][t] = va[t]; rb[it][t] = vb[t]; }
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
          for (int jt = 0; jt < 2; ++jt) acc[it][jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[it][t], rb[jt][t], acc[it][jt], 0, 0, 0);
    }
    if (LEVEL >= 2) {
      float* d = smem + (buf ^ 1) * 2 * 128 * LDK;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        f32x4v v = {1.f, 2.f, 3.f, 4.f};
        if (LEVEL >= 3) v = rg[u];
        *(f32x4v*)(d + ((tid >> 3) + 32 * u) * LDK + 4 * (tid & 7)) = v;
      }
      __syncthreads();
    }
  }
  float s = 0.f;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int LEVEL>
void run(int blocks, size_t lds, const float* src, float* out) {
  const int chunks = 400;
  hipLaunchKernelGGL(loop<LEVEL>, dim3(blocks), dim3(256), lds, 0, src, out, 10);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(loop<LEVEL>, dim3(blocks), dim3(256), lds, 0, src, out, chunks);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * chunks * 64 * 4096.0;
  printf("level %d, %4d workgroups, %6zu B LDS each: %7.3f ms  %6.1f TFLOP/s (%.1f %% of 157.3)\n", LEVEL, blocks, lds, ms, flops / ms / 1e9, flops / ms / 1e9 / 1.573);
}

int main() {
  float *src, *out; hipMalloc(&src, 64 * 8192 * 4 + 65536); hipMemset(src, 0, 64 * 8192 * 4 + 65536); hipMalloc(&out, 4096 * 256 * 4);
  const size_t l2 = 2 * 2 * 128 * LDK * 4, l1 = 100000;      // 73.7 KB: two per CU; 100 KB: one per CU
  for (int pass = 0; pass < 2; ++pass) {
    const size_t lds = pass ? l1 : l2; const int blocks = pass ? 256 * 4 : 512 * 4;
    run<0>(blocks, lds, src, out); run<1>(blocks, lds, src, out); run<2>(blocks, lds, src, out); run<3>(blocks, lds, src, out);
  }
  return 0;
}
