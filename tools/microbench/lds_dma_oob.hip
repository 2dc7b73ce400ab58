#include <hip/hip_runtime.h>
#include <cstdio>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* src, unsigned nbytes, unsigned* out, const unsigned* offs) {
  __shared__ __attribute__((aligned(16))) unsigned s[64 * 4 * 2];
  for (int i = threadIdx.x; i < 512; i += 64) s[i] = 0xdeadbeefu;
  __syncthreads();
  rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), (short)0, (int)nbytes, 0x00020000);
  // wave-uniform LDS base, lane i lands at base + 16 i
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)s, 16, (int)offs[threadIdx.x], 0, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(s + 256), 16, (int)offs[threadIdx.x], 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = s[i];
}
int main() {
  unsigned *src, *out, *offs; unsigned h[1024], ho[512], hoff[64];
  for (int i = 0; i < 1024; ++i) h[i] = i;
  for (int i = 0; i < 64; ++i) hoff[i] = (i % 5 == 0) ? 0x80000000u : (unsigned)(16 * ((i * 7) % 60));
  hipMalloc(&src, 4096); hipMalloc(&out, 2048); hipMalloc(&offs, 256);
  hipMemcpy(src, h, 4096, hipMemcpyHostToDevice); hipMemcpy(offs, hoff, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, 1024u, out, offs);
  hipMemcpy(ho, out, 2048, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) {
    unsigned exp0 = hoff[i] == 0x80000000u ? 0u : hoff[i] / 4;
    for (int b = 0; b < 2; ++b) {
      unsigned got = ho[b * 256 + 4 * i];
      if (i < 12 && b == 0) printf("lane %d off %x got %x %x %x %x\n", i, hoff[i], ho[4*i], ho[4*i+1], ho[4*i+2], ho[4*i+3]);
      if (hoff[i] != 0x80000000u && got != exp0) bad++;
      if (hoff[i] == 0x80000000u && got != 0) { bad += 1000; }
    }
  }
  printf("bad=%d (>=1000: OOB lanes did not get zeros)\n", bad);
  return 0;
}
