// Rate of the int8 matrix-core instructions on gfx950 (dense, register operands), as a basis for moving the key-switch dot product
// (30-bit operands split into four bytes) onto them -- DESIGN.md section 8 (1).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_mfma_i8 tools/ubench_mfma_i8.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
#define ITER 4096
// K = 16: v_mfma_i32_32x32x16_i8 (8 bytes of A and B per lane);  K = 32: v_mfma_i32_32x32x32_i8 (16 bytes per lane, gfx950)
template <int K> __global__ void __launch_bounds__(256) k(int* out, int seed) {
  v16i c[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) c[i][j] = 0;
  const long a8 = (long)(seed * 0x01010101u + threadIdx.x) * 0x100000001l, b8 = (long)(seed * 0x03030303u ^ threadIdx.x) * 0x100000001l;
  const v4i a16 = {(int)a8, (int)(a8 >> 32), (int)a8 ^ 0x5a5a5a5a, (int)(a8 >> 32) + 3}, b16 = {(int)b8, (int)(b8 >> 32), (int)b8 ^ 0x3c3c3c3c, (int)(b8 >> 32) + 7};
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (K == 16) c[i] = __builtin_amdgcn_mfma_i32_32x32x16_i8(a8, b8, c[i], 0, 0, 0);
      else c[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a16, b16, c[i], 0, 0, 0);
    }
  }
  int r = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) r ^= c[i][j];
  if (r == 0x12345678) out[threadIdx.x] = r;
}
template <int K> int run(int* d, int waves_per_simd, int ncu, int secs_hint) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = ncu * waves_per_simd;                  // 256 threads = one wave per SIMD
  k<K><<<blocks, 256>>>(d, 1); CK(hipDeviceSynchronize());
  const int reps = secs_hint;
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) k<K><<<blocks, 256>>>(d, 2 + r);
  hipEventRecord(e1); CK(hipEventSynchronize(e1));
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double macs = (double)reps * blocks * 4 /*waves*/ * ITER * 4 * 32.0 * 32.0 * K;
  printf("v_mfma_i32_32x32x%d_i8  waves/SIMD=%d : %.0f TOPS (2 ops per MAC)  = %.1f T byte-MACs/s  -> %.1f T 30-bit multiply-adds/s as 4 x 4 bytes (VALU v_mad_u64_u32: 33)   [%.1f ms]\n",
         K, waves_per_simd, 2 * macs / (ms * 1e-3) / 1e12, macs / (ms * 1e-3) / 1e12, macs / 16 / (ms * 1e-3) / 1e12, ms);
  return 0;
}
int main(int argc, char** argv) {
  int* d; CK(hipMalloc(&d, 4096));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int ncu = p.multiProcessorCount, reps = argc > 1 ? atoi(argv[1]) : 20;
  printf("device CUs=%d clock=%d kHz\n", ncu, p.clockRate);
  for (int w : {1, 2, 4}) { if (run<16>(d, w, ncu, reps)) return 1; if (run<32>(d, w, ncu, reps)) return 1; }
  return 0;
}
