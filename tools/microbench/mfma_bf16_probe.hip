// How v_mfma_f32_32x32x16_bf16 sums its 16 products and the accumulator: one wave, one instruction, hand-picked operands.
// build: hipcc -O2 --offload-arch=gfx950 tools/microbench/mfma_bf16_probe.hip -o tools/microbench/mfma_bf16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(const float* a16, const float* b16, float c0, float* out) {
  const int lane = threadIdx.x, i = lane & 31, kb = lane >> 5;
  bf16x8 a, b;
  for (int t = 0; t < 8; ++t) {
    const float av = i == 0 ? a16[8 * kb + t] : 0.f, bv = i == 0 ? b16[8 * kb + t] : 0.f;
    a[t] = (short)(__builtin_bit_cast(unsigned, av) >> 16); b[t] = (short)(__builtin_bit_cast(unsigned, bv) >> 16);
  }
  f32x16 c; for (int r = 0; r < 16; ++r) c[r] = 0.f;
  if (lane == 0) c[0] = c0;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  if (lane == 0) out[0] = c[0];
}
static float run(const float* a, const float* b, float c0) {
  static float *da = nullptr, *db, *dout;
  if (!da) { hipMalloc(&da, 64); hipMalloc(&db, 64); hipMalloc(&dout, 4); }
  hipMemcpy(da, a, 64, hipMemcpyHostToDevice); hipMemcpy(db, b, 64, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, c0, dout);
  float o; hipMemcpy(&o, dout, 4, hipMemcpyDeviceToHost); return o;
}
int main() {
  float a[16], b[16];
  for (int k = 0; k < 16; ++k) b[k] = 1.f;
  for (int e : {8, 16, 20, 23, 24, 25, 30}) {
    memset(a, 0, sizeof a); a[0] = 1.f; a[1] = ldexpf(1.f, -e);
    printf("1*1 + 2^-%d*1, C=0      : 1 + %.6e (exact %.6e)\n", e, (double)run(a, b, 0.f) - 1.0, ldexp(1.0, -e));
    memset(a, 0, sizeof a); a[0] = 1.f; a[9] = ldexpf(1.f, -e);
    printf("  (small in other k half) : 1 + %.6e\n", (double)run(a, b, 0.f) - 1.0);
    memset(a, 0, sizeof a); a[1] = ldexpf(1.f, -e);
    printf("  C=1 + 2^-%d*1           : 1 + %.6e\n", e, (double)run(a, b, 1.f) - 1.0);
  }
  for (int e : {20, 24, 26}) {
    for (int k = 0; k < 16; ++k) a[k] = ldexpf(1.f, -e); a[0] = 1.f;
    printf("1 + 15 x 2^-%d, C=0      : 1 + %.6e (exact %.6e)\n", e, (double)run(a, b, 0.f) - 1.0, 15 * ldexp(1.0, -e));
    a[0] = ldexpf(1.f, -e);
    printf("C=1 + 16 x 2^-%d         : 1 + %.6e (exact %.6e)\n", e, (double)run(a, b, 1.f) - 1.0, 16 * ldexp(1.0, -e));
  }
  // full-width products
  memset(a, 0, sizeof a); memset(b, 0, sizeof b);
  a[0] = 1.f + ldexpf(1.f, -7); b[0] = 1.f + ldexpf(1.f, -7);
  printf("(1+2^-7)^2 = %.10f exact %.10f\n", run(a, b, 0.f), (1 + ldexp(1.0, -7)) * (1 + ldexp(1.0, -7)));
  printf("C=256 + (1+2^-7)^2 = %.10f exact %.10f\n", run(a, b, 256.f), 256 + (1 + ldexp(1.0, -7)) * (1 + ldexp(1.0, -7)));
  a[1] = 3.f; b[1] = 1.f;
  printf("(1+2^-7)^2 + 3 = %.10f exact %.10f\n", run(a, b, 0.f), 3 + (1 + ldexp(1.0, -7)) * (1 + ldexp(1.0, -7)));
  a[1] = 1024.f;
  printf("(1+2^-7)^2 + 1024 = %.10f exact %.10f\n", run(a, b, 0.f), 1024 + (1 + ldexp(1.0, -7)) * (1 + ldexp(1.0, -7)));
  memset(a, 0, sizeof a); memset(b, 0, sizeof b);
  a[0] = 1.f; a[1] = ldexpf(1.f, -20); b[1] = 1.f;
  printf("[1, s] . [0, 1] / s = %.6f\n", run(a, b, 0.f) / ldexp(1.0, -20));
  b[0] = 1.f; b[1] = 0.f;
  printf("[1, s] . [1, 0] = %.9f\n", run(a, b, 0.f));
  a[0] = 0.f; b[1] = 1.f;
  printf("[0, s] . [1, 1] / s = %.6f\n", run(a, b, 0.f) / ldexp(1.0, -20));
  return 0;
}
