// Prototype of DESIGN.md section 8 (1) on the device: the key-switch dot product of a tile (8 ciphertexts x 66 columns -> 30 (limb, row)
// outputs per coefficient, one auxiliary prime) as signed-int8 matrix-core products with the lane-local recombination, checked against a
// direct host computation and timed.  Operands come pre-arranged in the instruction's layout (the key table would be stored that way at
// upload; the digit tile would be transposed on its way into LDS) -- this measures the arithmetic, not the data path.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_dot_tile tools/mfma_dot_tile.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned long long u64; typedef long long i64; typedef unsigned int u32;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
constexpr int NCOL = 66, CT = 8, NC = 30, KS = 3 /* k steps of 32 */, NJ = 4 /* byte planes of the key = column blocks */;
// A: [coef][KS][64 lanes] v4i;  B: [coef][KS][NJ][64 lanes] v4i;  S: [coef][NJ][32] int (column sums);  out: [coef][CT][32] u32
__global__ void __launch_bounds__(256) tile_kernel(const v4i* __restrict__ A, const v4i* __restrict__ B, const int* __restrict__ S, u32* __restrict__ out, int ncoef, u32 p, u32 r32) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwave = (gridDim.x * blockDim.x) >> 6;
  const int c = lane & 31, h = lane >> 5;
  for (int e = wave; e < ncoef; e += nwave) {
    v16i acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const v4i a = A[((i64)e * KS + s) * 64 + lane];
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, B[(((i64)e * KS + s) * NJ + j) * 64 + lane], acc[j], 0, 0, 0);
    }
    int sj[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) sj[j] = 128 * S[((i64)e * NJ + j) * 32 + c];
#pragma unroll
    for (int q = 0; q < 4; ++q) {                       // ciphertext 4 h + q of the tile; plane i of the digits is register 4 i + q
      i64 t[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) t[i + j] += acc[j][4 * i + q] + sj[j];
      const i64 lo = t[0] + (t[1] << 8) + (t[2] << 16) + (t[3] << 24);       // |lo| < 2^50
      i64 hi = t[4] + (t[5] << 8) + (t[6] << 16);                            // |hi| < 2^42, weight 2^32
      hi %= (i64)p; if (hi < 0) hi += p;
      i64 v = (hi * (i64)r32 + lo) % (i64)p;                                 // (prototype: plain 64-bit remainders)
      if (v < 0) v += p;
      out[((i64)e * CT + 4 * h + q) * 32 + c] = (u32)v;
    }
  }
}
int main(int argc, char** argv) {
  const int ncoef = argc > 1 ? atoi(argv[1]) : 1 << 16;                      // coefficient positions x tiles
  const u32 p = (1u << 30) - 3 * (1u << 15) + 1, r32 = (u32)((1ull << 32) % p);
  std::vector<u32> D((size_t)ncoef * CT * NCOL), K((size_t)ncoef * NCOL * NC);
  srand(99);
  for (auto& x : D) x = (u32)(((u64)rand() << 16 ^ rand()) % p);
  for (auto& x : K) x = (u32)(((u64)rand() << 16 ^ rand()) % p);
  for (int t = 0; t < CT * NCOL; ++t) D[t] = p - 1;                          // first coefficient: extreme values
  for (int t = 0; t < NCOL * NC; ++t) K[t] = p - 1;
  std::vector<int8_t> hA((size_t)ncoef * KS * 64 * 16, 0), hB((size_t)ncoef * KS * NJ * 64 * 16, 0);
  std::vector<int> hS((size_t)ncoef * NJ * 32, 0);
  for (int e = 0; e < ncoef; ++e) {
    for (int k = 0; k < NCOL; ++k) {
      const int s = k / 32, half = (k % 32) / 16, t = k % 16;
      for (int ct = 0; ct < CT; ++ct)
        for (int i = 0; i < 4; ++i) hA[((((size_t)e * KS + s) * 64) + (half * 32 + i * 8 + ct)) * 16 + t] = (int8_t)((int)((D[((size_t)e * CT + ct) * NCOL + k] >> (8 * i)) & 255) - 128);
      for (int c = 0; c < NC; ++c) {
        i64 x = K[((size_t)e * NCOL + k) * NC + c];
        for (int j = 0; j < NJ; ++j) {
          i64 b = x & 255; if (j < NJ - 1 && b >= 128) b -= 256;
          x = (x - b) >> 8;
          hB[(((((size_t)e * KS + s) * NJ + j) * 64) + (half * 32 + c)) * 16 + t] = (int8_t)b;
          hS[((size_t)e * NJ + j) * 32 + c] += (int)b;
        }
      }
    }
  }
  void *dA, *dB; int* dS; u32* dO;
  CK(hipMalloc(&dA, hA.size())); CK(hipMalloc(&dB, hB.size())); CK(hipMalloc(&dS, hS.size() * 4)); CK(hipMalloc(&dO, (size_t)ncoef * CT * 32 * 4));
  CK(hipMemcpy(dA, hA.data(), hA.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dS, hS.data(), hS.size() * 4, hipMemcpyHostToDevice));
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const int blocks = pr.multiProcessorCount * 2;
  tile_kernel<<<blocks, 256>>>((const v4i*)dA, (const v4i*)dB, dS, dO, ncoef, p, r32); CK(hipDeviceSynchronize());
  std::vector<u32> out((size_t)ncoef * CT * 32);
  CK(hipMemcpy(out.data(), dO, out.size() * 4, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (int e = 0; e < ncoef && e < 4096; ++e)
    for (int ct = 0; ct < CT; ++ct)
      for (int c = 0; c < NC; ++c) {
        unsigned __int128 acc = 0;
        for (int k = 0; k < NCOL; ++k) acc += (u64)D[((size_t)e * CT + ct) * NCOL + k] * K[((size_t)e * NCOL + k) * NC + c];
        bad += (u32)(acc % p) != out[((size_t)e * CT + ct) * 32 + c];
      }
  printf("checked %d coefficient-tiles against the direct dot product: %zu mismatches\n", ncoef < 4096 ? ncoef : 4096, bad);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 20;
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) tile_kernel<<<blocks, 256>>>((const v4i*)dA, (const v4i*)dB, dS, dO, ncoef, p, r32);
  hipEventRecord(e1); CK(hipEventSynchronize(e1));
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mads = (double)ncoef * CT * NC * NCOL;               // per launch (ms / reps below is per launch too)
  printf("%d coefficient-tiles in %.3f ms: %.1f T useful 30-bit multiply-adds/s with %.1f TB/s of operand bytes streamed from HBM (12 KB of key bytes per coefficient-tile of 8 ciphertexts);\n"
         "dot32_kernel2: 1.33e11 per launch of 1024 in 10 ms = 13.3 T/s with its key words from L2\n", ncoef, ms / reps, mads / (ms / reps * 1e-3) / 1e12,
         ((double)hA.size() + hB.size()) / (ms / reps * 1e-3) / 1e12);
  return bad != 0;
}
