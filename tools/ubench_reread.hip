// What does it cost to touch a tile of a streamed array TWICE from the same workgroup, one tile-time apart?
//   mode 0: every 128-byte line read once, whole (the reference rate)
//   mode 1: the first 64 bytes of every line of the tile, then the second 64 bytes (a tile = `tile` bytes of lines)
//   mode 2: the whole tile read twice in a row (full lines both times)
// One persistent workgroup per CU (x `wpc`), tiles handed out round robin; every tile also writes `tile * wfrac / 8` bytes (full lines)
// so that the L2 sees the write stream of the real kernel.  Reported: tiles' bytes / time (each byte counted once).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_reread tools/ubench_reread.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32;
typedef u32 v4u __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
template <int MODE>
__global__ void __launch_bounds__(512) k(const v4u* __restrict__ src, v4u* __restrict__ dst, long ntiles, int tile16 /* 16-byte words per tile */, int w16 /* 16-byte words written per tile */) {
  v4u acc = {0, 0, 0, 0};
  for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const v4u* s = src + t * tile16;
    if (MODE == 0) {
      for (int i = threadIdx.x; i < tile16; i += 512) { const v4u v = __builtin_nontemporal_load(s + i); acc ^= v; }
    } else if (MODE == 1) {
      for (int h = 0; h < 2; ++h)
        for (int i = threadIdx.x; i < tile16 / 2; i += 512) { const int line = i >> 2, w = i & 3; const v4u v = s[line * 8 + h * 4 + w]; acc ^= v; }
    } else {
      for (int h = 0; h < 2; ++h)
        for (int i = threadIdx.x; i < tile16; i += 512) { const v4u v = s[i]; acc ^= v; acc += h; }
    }
    v4u* d = dst + t * w16;
    for (int i = threadIdx.x; i < w16; i += 512) __builtin_nontemporal_store(acc, d + i);
  }
}
template <int MODE> int run(const v4u* s, v4u* d, long bytes, int tile, int wpc, int ncu) {
  const long ntiles = bytes / tile;
  const int tile16 = tile / 16, w16 = tile16 * 3 / 8;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<ncu * wpc, 512>>>(s, d, ntiles, tile16, w16); CK(hipDeviceSynchronize());
  const int reps = 3;
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) k<MODE><<<ncu * wpc, 512>>>(s, d, ntiles, tile16, w16);
  hipEventRecord(e1); CK(hipEventSynchronize(e1));
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("mode %d tile %6d B wg/cu %d : %.3f ms, %.0f GB/s of distinct bytes (read + written)\n", MODE, tile, wpc, ms / reps, (double)ntiles * (tile + w16 * 16.0) / (ms / reps * 1e-3) / 1e9);
  return 0;
}
int main() {
  const long bytes = 8l << 30;
  v4u *s, *d; CK(hipMalloc(&s, bytes)); CK(hipMalloc(&d, bytes / 2));
  CK(hipMemset(s, 1, bytes));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  for (int tile : {135168, 270336, 67584})
    for (int wpc : {1, 2}) {
      if (run<0>(s, d, bytes, tile, wpc, p.multiProcessorCount)) return 1;
      if (run<1>(s, d, bytes, tile, wpc, p.multiProcessorCount)) return 1;
      if (run<2>(s, d, bytes, tile, wpc, p.multiProcessorCount)) return 1;
    }
  return 0;
}
