// How much HBM bandwidth survives when rows are scattered / gathered in segments of g bytes (g = 32 ... 256)?  The question behind the
// layout of the digit rows between the forward transform (a workgroup owns a row) and the key-switch dot product (a workgroup owns a
// block of coefficients of many rows): one of the two has to touch memory in pieces shorter than a row, and a 128-byte line written
// or read in halves by DIFFERENT workgroups may cost twice.
//   scatter: unit u reads its 64 KiB row contiguously and writes segment t to dst[(t * U + u) * g]   (the neighbours of a segment
//            belong to units u - 1, u + 1: other workgroups, other XCDs unless `pair` puts u and u ^ 1 on the same XCD)
//   gather:  the reverse (segments read, row written contiguously)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_seg tools/ubench_seg.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32;
typedef u32 v4u __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
constexpr long ROW = 65536;            // bytes per unit
template <int G, bool SCATTER, bool PAIR, bool NT>
__global__ void __launch_bounds__(256) k(const v4u* __restrict__ src, v4u* __restrict__ dst, long U) {
  const u32 b = blockIdx.x;
  const long u = PAIR ? ((long)((b >> 4) * 8 + (b & 7)) * 2 + ((b >> 3) & 1)) : (long)b;
  constexpr int LPS = G / 16;          // lanes per segment
  const long nseg = ROW / G;
#pragma unroll 4
  for (long i = threadIdx.x; i < ROW / 16; i += 256) {
    const long t = i / LPS, w = i % LPS;
    const long lin = u * (ROW / 16) + i, seg = (t * U + u) * LPS + w;
    if (SCATTER) { const v4u v = NT ? __builtin_nontemporal_load(src + lin) : src[lin]; if (NT) __builtin_nontemporal_store(v, dst + seg); else dst[seg] = v; }
    else { const v4u v = NT ? __builtin_nontemporal_load(src + seg) : src[seg]; if (NT) __builtin_nontemporal_store(v, dst + lin); else dst[lin] = v; }
  }
  (void)nseg;
}
template <int G, bool SCATTER, bool PAIR, bool NT> int run(const v4u* s, v4u* d, long U) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<G, SCATTER, PAIR, NT><<<(unsigned)U, 256>>>(s, d, U); CK(hipDeviceSynchronize());
  const int reps = 5;
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) k<G, SCATTER, PAIR, NT><<<(unsigned)U, 256>>>(s, d, U);
  hipEventRecord(e1); CK(hipEventSynchronize(e1));
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-7s g=%3d pair=%d nt=%d : %.3f ms per pass, %.0f GB/s (read + write)\n", SCATTER ? "scatter" : "gather", G, (int)PAIR, (int)NT, ms / reps, 2.0 * U * ROW / (ms / reps * 1e-3) / 1e9);
  return 0;
}
int main() {
  const long U = 65536;                // 4 GiB each way
  v4u *s, *d; CK(hipMalloc(&s, U * ROW)); CK(hipMalloc(&d, U * ROW));
  CK(hipMemset(s, 1, U * ROW)); CK(hipMemset(d, 2, U * ROW));
#define ALL(G) \
  if (run<G, true, false, false>(s, d, U)) return 1; if (run<G, true, true, false>(s, d, U)) return 1; if (run<G, true, false, true>(s, d, U)) return 1; if (run<G, true, true, true>(s, d, U)) return 1; \
  if (run<G, false, false, false>(s, d, U)) return 1; if (run<G, false, true, false>(s, d, U)) return 1; if (run<G, false, false, true>(s, d, U)) return 1;
  ALL(256) ALL(128) ALL(64) ALL(32)
  return 0;
}
