// Operand / accumulator layout of the int8 matrix-core instructions on gfx950, checked against a host product: which byte of which lane
// is A[i][k] / B[k][j], and which (lane, register) holds C[i][j].  The basis for laying out the key table and the digit tile of the
// key-switch dot product for these instructions.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_i8_layout tools/mfma_i8_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
// every lane hands over its raw operand bytes; the host decides what they mean
__global__ void k16(const long* a, const long* b, int* out) {
  v16i c; for (int j = 0; j < 16; ++j) c[j] = 0;
  c = __builtin_amdgcn_mfma_i32_32x32x16_i8(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
  for (int j = 0; j < 16; ++j) out[threadIdx.x * 16 + j] = c[j];
}
__global__ void k32(const v4i* a, const v4i* b, int* out) {
  v16i c; for (int j = 0; j < 16; ++j) c[j] = 0;
  c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
  for (int j = 0; j < 16; ++j) out[threadIdx.x * 16 + j] = c[j];
}
// hypothesis: lane l, byte t of its operand  <->  k = kmap(l, t),  row / column = l % 32
static int kmap(int K, int hyp, int l, int t) {
  const int half = l / 32, per = K / 2;                       // bytes per lane
  if (hyp == 0) return half * per + t;                        // contiguous block of K/2 per half wave
  return (t / 8) * 16 + half * 8 + (t % 8);                   // K = 32 as two K = 16 steps: 8-byte groups interleaved between the half waves
}
template <int K> int check() {
  const int per = K / 2;
  std::vector<int8_t> A(32 * K), B(K * 32);
  srand(1234 + K);
  for (auto& x : A) x = (int8_t)(rand() % 256 - 128);
  for (auto& x : B) x = (int8_t)(rand() % 256 - 128);
  std::vector<int> C(32 * 32);
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { int s = 0; for (int k = 0; k < K; ++k) s += (int)A[i * K + k] * (int)B[k * 32 + j]; C[i * 32 + j] = s; }
  void *da, *db; int* dout;
  CK(hipMalloc(&da, 64 * per)); CK(hipMalloc(&db, 64 * per)); CK(hipMalloc(&dout, 64 * 16 * 4));
  for (int hyp = 0; hyp < (K == 32 ? 2 : 1); ++hyp) {
    std::vector<int8_t> ha(64 * per), hb(64 * per);
    for (int l = 0; l < 64; ++l) for (int t = 0; t < per; ++t) { const int k = kmap(K, hyp, l, t); ha[l * per + t] = A[(l % 32) * K + k]; hb[l * per + t] = B[k * 32 + (l % 32)]; }
    CK(hipMemcpy(da, ha.data(), ha.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), hb.size(), hipMemcpyHostToDevice));
    if (K == 16) k16<<<1, 64>>>((const long*)da, (const long*)db, dout); else k32<<<1, 64>>>((const v4i*)da, (const v4i*)db, dout);
    std::vector<int> out(64 * 16);
    CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
    int ok = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) { const int i = 8 * (r / 4) + 4 * (l / 32) + (r % 4), j = l % 32; ok += out[l * 16 + r] == C[i * 32 + j]; }
    printf("v_mfma_i32_32x32x%d_i8  operand bytes: lane l, byte t <-> k = %s, row/col = l %% 32;  C[i][j] in lane (j + 32 * ((i / 4) %% 2)), register 4 * (i / 8) + i %% 4:  %d of 1024 entries match%s\n",
           K, hyp == 0 ? (K == 16 ? "8 * (l / 32) + t" : "16 * (l / 32) + t") : "16 * (t / 8) + 8 * (l / 32) + t % 8", ok, ok == 1024 ? "  <-- the layout" : "");
  }
  return 0;
}
int main() { if (check<16>()) return 1; return check<32>(); }
