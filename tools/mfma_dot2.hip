// Prototype of the key-switch dot product on the int8 matrix cores, second design.
//
//   O[ct][col][a][j] = sum_k D[ct][k][a][j] * K[a][col][k][j]  mod p_a        66 columns k, 30 outputs col = (limb, key row), 4 primes, n = 2^14
//
// The contraction runs over (k, byte plane of D): a digit word W (any 32-bit value, lazy residues are fine) is its own four operand
// bytes -- flipped to signed with one XOR (W ^ 0x80808080 = the signed bytes of W - 0x80808080) -- and the key side carries the
// plane's weight:  B[(k, bp)][(bq, col)] = balanced byte bq of centred(K[k][col] * 256^bp mod p).  Per coefficient and 32 ciphertexts:
// 9 depth blocks x 4 key byte planes = 36 v_mfma_i32_32x32x32_i8, and only FOUR partial sums per output (|.| < 2^23, exact):
//   V = a0 + a1 2^8 + a2 2^16 + a3 2^24 = sum_k (W_k - 0x80808080) K_k  (mod p),   out = V + 0x80808080 sum_k K_k   (mod p).
// Tile = 16 coefficients x 32 ciphertexts; a wave owns two coefficients (128 accumulator registers); the 66 columns stream through a
// two-deep LDS ring in 9 phases of 8 columns; one persistent workgroup per CU; the two 16-coefficient halves of a 128-byte line are
// taken back to back by the same workgroup and their outputs leave together as whole lines.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_dot2 tools/mfma_dot2.hip        Run: tools/mfma_dot2 [ciphertexts = 1024] [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>
#ifndef ABL
#define ABL 0
#endif
#ifndef NT
#define NT 1
#endif
#ifndef PIN
#define PIN 0
#endif
typedef unsigned int u32;
typedef unsigned long long u64;
typedef long long i64;
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef u32 v4u __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)

constexpr int NCOL = 66, NS = 9, COLS = 30, NLB = 15, LOGN = 14, N = 1 << LOGN, NSL = N / 32;
constexpr int JS = 260;                  // words between coefficients of a ring phase: [16 j][32 ct][8 k] + 4
constexpr int RING = 16 * JS;            // words per phase
constexpr int HS = 32 * COLS + 1;        // words between coefficients of the held / staged outputs [16 j][32 ct][30 col] + 1
struct Consts { u32 p[4], mu50[4] /* floor(2^50 / p) */; };

__host__ __device__ inline u32 hash32(u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return (u32)x; }

// D [a][S][ct * 66 + k][32 j]: lazy residues below 4p;  K [a][col][k][n] below p
__global__ void init_kernel(u32* D, u32* K, int CT, Consts c) {
  const i64 nD = (i64)4 * N * CT * NCOL, nK = (i64)4 * COLS * NCOL * N;
  for (i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x; g < nD + nK; g += (i64)gridDim.x * blockDim.x) {
    if (g < nD) { const int a = (int)(g / ((i64)N * CT * NCOL)); D[g] = (u32)(((u64)hash32(g * 2 + 1) * (4ull * c.p[a])) >> 32); }
    else { const i64 h = g - nD; const int a = (int)(h / ((i64)COLS * NCOL * N)); K[h] = (u32)(((u64)hash32(h * 2) * c.p[a]) >> 32); }
  }
}
__device__ inline int balanced_byte(int c, int j) {      // c = sum_j b_j 256^j, b_0..b_2 in [-128, 127], b_3 the rest
  int b = 0;
  for (int jj = 0; jj <= j; ++jj) { b = jj < 3 ? ((c + 128) & 255) - 128 : c; c = (c - b) >> 8; }
  return b;
}
// Bt [a][j][s][bq][lane] x 16 bytes;  corr [a][j][32]
__global__ void table_kernel(const u32* K, v4i* Bt, u32* corr, Consts c) {
  const i64 nB = (i64)4 * N * NS * 4 * 64, nC = (i64)4 * N * 32;
  for (i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x; g < nB + nC; g += (i64)gridDim.x * blockDim.x) {
    if (g < nB) {
      const int lane = (int)(g & 63), bq = (int)((g >> 6) & 3);
      const i64 q = g >> 8;
      const int s = (int)(q % NS);
      const i64 aj = q / NS, j = aj % N;
      const int a = (int)(aj / N), col = lane & 31, kh = lane >> 5;
      const u64 p = c.p[a];
      u32 w[4] = {0, 0, 0, 0};
      for (int t = 0; t < 16; ++t) {
        const int k = 8 * s + 4 * kh + (t >> 2), bp = t & 3;
        int b = 0;
        if (k < NCOL && col < COLS) {
          const u64 kv = K[(((i64)a * COLS + col) * NCOL + k) * N + j];
          const u64 m = (kv << (8 * bp)) % p;
          const int cen = m > p / 2 ? (int)((i64)m - (i64)p) : (int)m;
          b = balanced_byte(cen, bq);
        }
        w[t >> 2] |= (u32)(b & 255) << (8 * (t & 3));
      }
      Bt[g] = v4i{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
    } else {
      const i64 h = g - nB;
      const int col = (int)(h & 31);
      const i64 aj = h >> 5, j = aj % N;
      const int a = (int)(aj / N);
      const u64 p = c.p[a];
      u64 sum = 0;
      if (col < COLS) for (int k = 0; k < NCOL; ++k) sum += K[(((i64)a * COLS + col) * NCOL + k) * N + j];
      corr[h] = (u32)(((sum % p) * (0x80808080ull % p)) % p);
    }
  }
}
// out [ct][r][l][a][n]  (col = 2 l + r), plain reference for the ciphertexts listed in cts
__global__ void ref_kernel(const u32* D, const u32* K, u32* ref, const int* cts, int ncts, int CT, Consts c) {
  const i64 total = (i64)ncts * COLS * 4 * N;
  for (i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (i64)gridDim.x * blockDim.x) {
    const int j = (int)(g % N);
    i64 q = g / N;
    const int a = (int)(q % 4); q /= 4;
    const int col = (int)(q % COLS), ci = (int)(q / COLS), ct = cts[ci];
    const u64 p = c.p[a];
    u64 acc = 0;
    for (int k = 0; k < NCOL; ++k) {
      const u64 d = D[((((i64)a * NSL + (j >> 5)) * CT + ct) * NCOL + k) * 32 + (j & 31)];
      acc = (acc + (d % p) * K[(((i64)a * COLS + col) * NCOL + k) * N + j]) % p;
    }
    ref[g] = (u32)acc;
  }
}

__global__ void __launch_bounds__(512, 1) dot_mfma2_kernel(const u32* __restrict__ D, const v4i* __restrict__ Bt, const u32* __restrict__ corr, u32* __restrict__ out, int CT, Consts cs) {
  extern __shared__ __attribute__((aligned(16))) u32 lds[];
  u32* const ring = lds;                       // [2][RING]
  u32* const held = lds + 2 * RING;            // [16][HS]   outputs of the first half tile
  u32* const stage = held + 16 * HS;           // [16][HS]   outputs of the second
  const u32 tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int G = CT >> 5;
  const int nitems = (4 * NSL / 8) * G;   // per XCD: blocks q = a * NSL + S with q = xcd mod 8, times the ciphertext groups
  // loader: thread -> (ciphertext, column within the phase, four coefficients)
  u32 offA[2], ldsA[2];
  bool k2[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const u32 idx = tid + 512 * i, ct_l = idx >> 5, kk = (idx >> 2) & 7, j4 = idx & 3;
    offA[i] = (ct_l * NCOL + kk) * 32 + 4 * j4;
    ldsA[i] = (4 * j4) * JS + ct_l * 8 + kk;
    k2[i] = kk < 2;
  }
  const u32 aoff = (lane & 31) * 8 + (lane >> 5) * 4;      // A operand of the lane inside a coefficient's [32 ct][8 k] block
  const int col = (int)(lane & 31);
  struct Item { const u32* dA; const v4i* bB; const u32* cC; int a, S, g; };
  auto item_of = [&](int it) {
    Item r;
    const int ql = it / G;
    r.g = it - ql * G;
    const int q = ql * 8 + xcd;
    r.a = q / NSL; r.S = q % NSL;
    r.dA = D + (((i64)q * CT + r.g * 32) * NCOL) * 32;
    r.bB = Bt + ((i64)r.a * N + r.S * 32 + 2 * w) * (NS * 4 * 64) + lane;
    r.cC = corr + ((i64)r.a * N + r.S * 32 + 2 * w) * 32 + col;
    return r;
  };
  if (slot >= nitems) return;
  Item cur = item_of(slot), nxt = cur;
  v4u st[3][2];
  v4i Bb[2][4];
  v16i acc[2][4];
  u32 pr = 0, mu = 0, cc2[2] = {0, 0};
#if PIN
#define SB() __builtin_amdgcn_sched_barrier(0)
#else
#define SB()
#endif
  auto issueA = [&](auto SET, const u32* base, auto PH) {
    constexpr int set = decltype(SET)::value, ph = decltype(PH)::value, h = ph / NS, s = ph % NS;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const u32* q = base + offA[i] + 256 * s + 16 * h;
#if ABL & 2
      st[set][i] = v4u{(u32)(size_t)q, tid, 3u, 4u};
#else
#if NT
      if (s < NS - 1 || k2[i]) st[set][i] = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(q)); else st[set][i] = v4u{0, 0, 0, 0};
#else
      if (s < NS - 1 || k2[i]) st[set][i] = *reinterpret_cast<const v4u*>(q); else st[set][i] = v4u{0, 0, 0, 0};
#endif
#endif
    }
  };
  auto writeA = [&](u32* rg, auto SET) {
    constexpr int set = decltype(SET)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) rg[ldsA[i] + e * JS] = st[set][i][e] ^ 0x80808080u;
  };
  auto issueB = [&](auto X, const v4i* base, auto PH) {
    constexpr int x = decltype(X)::value, ph = decltype(PH)::value, h = ph / NS, s = ph % NS;
#pragma unroll
    for (int bq = 0; bq < 4; ++bq) {
#if ABL & 1
      Bb[x][bq] = v4i{(int)(size_t)base, (int)tid + ph, bq, x};
#else
      Bb[x][bq] = base[((i64)(h * 16 + x) * NS * 4 + s * 4 + bq) * 64];
#endif
    }
  };
#define IC(v) std::integral_constant<int, (v)>{}
  auto step = [&](auto PH) {
    constexpr int ph = decltype(PH)::value, h = ph / NS, s = ph % NS;
    // a. digit words three phases ahead
    if constexpr (ph + 3 < 2 * NS) issueA(IC(ph % 3), cur.dA, IC(ph + 3)); else issueA(IC(ph % 3), nxt.dA, IC(ph + 3 - 2 * NS));
    SB();
    // b. the next phase into the other half of the ring
    writeA(ring + ((ph + 1) & 1) * RING, IC((ph + 1) % 3));
    SB();
    // c. this phase; the key operands of the next phase are requested into the registers the matrix instructions have just read
    const u32* rg = ring + (ph & 1) * RING;
    auto slot_x = [&](auto X) {
      constexpr int x = decltype(X)::value;
      const v4i av = *reinterpret_cast<const v4i*>(rg + (2 * w + x) * JS + aoff);
#pragma unroll
      for (int bq = 0; bq < 4; ++bq) {
        if constexpr (s == 0) { const v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; acc[x][bq] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, Bb[x][bq], z, 0, 0, 0); }
        else acc[x][bq] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, Bb[x][bq], acc[x][bq], 0, 0, 0);
      }
      SB();
      if constexpr (ph + 1 < 2 * NS) issueB(X, cur.bB, IC(ph + 1)); else issueB(X, nxt.bB, IC(0));
      SB();
    };
    slot_x(IC(0)); slot_x(IC(1));
    if constexpr (s == NS - 2) { cc2[0] = cur.cC[(h * 16) * 32]; cc2[1] = cur.cC[(h * 16 + 1) * 32]; }
    if constexpr (s == NS - 1) {
      // d. four partial sums per output -> residue below 2p
      u32* dst = (h == 0 ? held : stage) + (2 * w) * HS + 4 * (int)(lane >> 5) * COLS + col;
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const u64 C64 = ((u64)pr << 18) + cc2[x];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ct = 8 * (r >> 2) + (r & 3);                 // + 4 (lane >> 5)
          const int u = acc[x][0][r] + acc[x][1][r] * 256, v = acc[x][2][r] + acc[x][3][r] * 256;
          const u64 T = (u64)((i64)v * 65536 + (i64)u + (i64)C64);
          const u32 qh = __umulhi((u32)(T >> 18), mu);
          const u32 res = (u32)T - qh * pr;
          if (col < COLS) dst[x * HS + ct * COLS] = res;
        }
      }
    }
    __syncthreads();
    if constexpr (ph == 2 * NS - 1) {
      // e. the 960 rows of the tile pair, 128 bytes each: lanes 0..31 key row 0, lanes 32..63 key row 1 of (ciphertext 4 w + i, limb l)
      const u32 jj = lane & 31, r = lane >> 5;
      const u32* src = (jj < 16 ? held : stage) + (jj & 15) * HS + (4 * w) * COLS + r;
      u32* o = out + (((((i64)cur.g * 32 + 4 * w) * 2 + r) * NLB * 4 + cur.a) << LOGN) + cur.S * 32 + jj;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int l = 0; l < NLB; ++l) {
          const u32 v = src[i * COLS + 2 * l];
#if ABL & 4
          if (v == 0x12345678u)
#endif
          __builtin_nontemporal_store(v, o + ((i64)(i * 2 * NLB + l) * 4 << LOGN));
        }
    }
  };
  // prologue
  issueB(IC(0), cur.bB, IC(0)); issueB(IC(1), cur.bB, IC(0));
  issueA(IC(0), cur.dA, IC(0)); issueA(IC(1), cur.dA, IC(1)); issueA(IC(2), cur.dA, IC(2));
  writeA(ring, IC(0));
  __syncthreads();
  for (int it = slot; it < nitems; it += nslot) {
    nxt = it + nslot < nitems ? item_of(it + nslot) : cur;
    pr = cs.p[cur.a]; mu = cs.mu50[cur.a];
    step(IC(0)); step(IC(1)); step(IC(2)); step(IC(3)); step(IC(4)); step(IC(5)); step(IC(6)); step(IC(7)); step(IC(8));
    step(IC(9)); step(IC(10)); step(IC(11)); step(IC(12)); step(IC(13)); step(IC(14)); step(IC(15)); step(IC(16)); step(IC(17));
    cur = nxt;
  }
}

static bool is_prime(u64 n) { if (n < 2) return false; for (u64 d = 2; d * d <= n; ++d) if (n % d == 0) return false; return true; }
int main(int argc, char** argv) {
  const int CT = argc > 1 ? atoi(argv[1]) : 1024, reps = argc > 2 ? atoi(argv[2]) : 5;
  if (CT % 32) { printf("ciphertexts: a multiple of 32\n"); return 1; }
  Consts cs;
  int found = 0;
  for (u64 k = ((u64)1 << 15) - 1; k > 0 && found < 4; --k) { const u64 cand = (k << 15) + 1; if (cand < ((u64)1 << 30) && is_prime(cand)) { cs.p[found] = (u32)cand; cs.mu50[found] = (u32)(((u64)1 << 50) / cand); ++found; } }
  printf("primes %u %u %u %u, %d ciphertexts\n", cs.p[0], cs.p[1], cs.p[2], cs.p[3], CT);
  const size_t nD = (size_t)4 * N * CT * NCOL, nK = (size_t)4 * COLS * NCOL * N, nB = (size_t)4 * N * NS * 4 * 64, nC = (size_t)4 * N * 32, nO = (size_t)CT * COLS * 4 * N;
  u32 *D, *K, *corr, *out, *ref; v4i* Bt; int* dcts;
  CK(hipMalloc(&D, nD * 4)); CK(hipMalloc(&K, nK * 4)); CK(hipMalloc(&Bt, nB * 16)); CK(hipMalloc(&corr, nC * 4)); CK(hipMalloc(&out, nO * 4));
  const int cts[6] = {0, 1, 31, 32 % CT, CT / 2 + 5, CT - 1};
  CK(hipMalloc(&ref, (size_t)6 * COLS * 4 * N * 4)); CK(hipMalloc(&dcts, sizeof cts)); CK(hipMemcpy(dcts, cts, sizeof cts, hipMemcpyHostToDevice));
  init_kernel<<<8192, 256>>>(D, K, CT, cs); CK(hipDeviceSynchronize());
  table_kernel<<<8192, 256>>>(K, Bt, corr, cs); CK(hipDeviceSynchronize());
  ref_kernel<<<8192, 256>>>(D, K, ref, dcts, 6, CT, cs); CK(hipDeviceSynchronize());
  CK(hipMemset(out, 0xff, nO * 4));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  const size_t shmem = (size_t)(2 * RING + 32 * HS) * 4;
  CK(hipFuncSetAttribute((const void*)dot_mfma2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  printf("LDS %zu bytes, grid %d\n", shmem, ncu);
  dot_mfma2_kernel<<<ncu, 512, shmem>>>(D, Bt, corr, out, CT, cs); CK(hipDeviceSynchronize());
  // check
  std::vector<u32> ho((size_t)COLS * 4 * N), hr((size_t)6 * COLS * 4 * N);
  CK(hipMemcpy(hr.data(), ref, hr.size() * 4, hipMemcpyDeviceToHost));
  long bad = 0, lazy = 0;
  for (int ci = 0; ci < 6; ++ci) {
    CK(hipMemcpy(ho.data(), out + (size_t)cts[ci] * COLS * 4 * N, ho.size() * 4, hipMemcpyDeviceToHost));
    for (int cc = 0; cc < COLS; ++cc) for (int a = 0; a < 4; ++a) for (int j = 0; j < N; ++j) {
      const u32 got = ho[((size_t)((cc & 1) * NLB + (cc >> 1)) * 4 + a) * N + j], want = hr[(((size_t)ci * COLS + cc) * 4 + a) * N + j];
      if (got >= 2 * cs.p[a] || got % cs.p[a] != want) { if (bad < 5) printf("MISMATCH ct %d col %d prime %d j %d: got %u want %u\n", cts[ci], cc, a, j, got, want); ++bad; }
      else if (got >= cs.p[a]) ++lazy;
    }
  }
  printf("check: %ld mismatches of %zu (%ld values in [p, 2p))\n", bad, (size_t)6 * COLS * 4 * N, lazy);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) dot_mfma2_kernel<<<ncu, 512, shmem>>>(D, Bt, corr, out, CT, cs);
  hipEventRecord(e1); CK(hipEventSynchronize(e1));
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)nD * 4 + (double)nO * 4 + (double)nB * 16 + (double)nC * 4;
  printf("dot_mfma2: %.3f ms per launch of %d ciphertexts; %.1f GB moved at least -> %.0f GB/s; %.1f T multiply-adds/s\n", ms / reps, CT, bytes / 1e9, bytes / (ms / reps * 1e-3) / 1e9,
         (double)CT * COLS * NCOL * 4 * N / (ms / reps * 1e-3) / 1e12);
  return bad != 0;
}
