// v_mad_u64_u32 issue behaviour on gfx950: independent accumulators vs back-to-back dependent chains, at 1/2/4/8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_mad tools/ubench_mad.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64; typedef unsigned int u32;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
#define ITER 2048
// DEP = number of consecutive mads on the same accumulator before moving to the next of 16 accumulators
template <int DEP> __global__ void k(u64* out, u32 seed) {
  u64 a[16]; u32 x[8], d[8];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x + i;
  for (int i = 0; i < 8; ++i) { x[i] = seed * (i + 3) + threadIdx.x; d[i] = seed * (i + 11) ^ threadIdx.x; }
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
#pragma unroll
      for (int j = 0; j < DEP; ++j)
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[g]) : "v"(x[(g + j) & 7]), "v"(d[(g * 3 + j) & 7]) : "vcc");
    }
  }
  u64 r = 0; for (int i = 0; i < 16; ++i) r ^= a[i];
  if (r == 0x123456789ull) out[threadIdx.x] = r;
}
// same with an SGPR pair as the (unused) carry-out, as hipcc emits it
template <int DEP> __global__ void ks(u64* out, u32 seed) {
  u64 a[16]; u32 x[8], d[8];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x + i;
  for (int i = 0; i < 8; ++i) { x[i] = seed * (i + 3) + threadIdx.x; d[i] = seed * (i + 11) ^ threadIdx.x; }
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
#pragma unroll
      for (int j = 0; j < DEP; ++j)
        asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(a[g]) : "v"(x[(g + j) & 7]), "v"(d[(g * 3 + j) & 7]) : "s20", "s21");
    }
  }
  u64 r = 0; for (int i = 0; i < 16; ++i) r ^= a[i];
  if (r == 0x123456789ull) out[threadIdx.x] = r;
}
int main() {
  CK(hipSetDevice(0));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  u64* d; CK(hipMalloc(&d, 1 << 20));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#define RUN(K, DEP, WPS) { int threads = 256 * (WPS > 4 ? 4 : WPS); int blocks = p.multiProcessorCount * (WPS > 4 ? WPS / 4 : 1); K<DEP><<<blocks, threads>>>(d, 1); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); \
    for (int r = 0; r < 3; ++r) K<DEP><<<blocks, threads>>>(d, r + 2); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); \
    double mads = 3.0 * blocks * threads * (double)ITER * 16 * DEP; printf("%-4s dep=%d waves/SIMD=%d : %8.2f Glane-mads/s  -> %.2f cycles per wave-mad per SIMD at 2.0 GHz\n", #K, DEP, WPS, mads / ms / 1e6, 1024.0 * 2.0e9 * 64 / (mads / ms * 1e3)); }
  RUN(k, 1, 1) RUN(k, 1, 2) RUN(k, 1, 4) RUN(k, 1, 8)
  RUN(k, 2, 4) RUN(k, 3, 4) RUN(k, 4, 4) RUN(k, 8, 4) RUN(k, 8, 1)
  RUN(ks, 1, 4) RUN(ks, 3, 4)
  return 0;
}
