// Prototype of the key-switch dot product on the int8 matrix cores, second design, hand-pipelined, NW = 8 waves x 2 coefficients or 16 waves x 1:
// the digit words go global -> LDS by DMA (global_load_lds_dwordx4) into a four-phase ring, the key operands into a register double
// buffer, every load and every s_waitcnt vmcnt(N) is written out (the compiler drains or re-orders a pipeline it schedules itself).
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
#ifndef LAY16
#define LAY16 0
#endif
#ifndef HOT
#define HOT 0
#endif
#ifndef NWAVES
#define NWAVES 16
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
constexpr int PHB = 16384, RINGN = 4;    // bytes per ring phase [8 k][2 ct halves][16 ct][4 granules of 4 coefficients], phases in the ring
constexpr int HS = 32 * COLS + 1;        // words between coefficients of the held outputs [16 j][32 ct][30 col] + 1
constexpr int HS2 = 32 * (COLS / 2) + 1; // ... of the staged half [16 j][32 ct][15 col] + 1
constexpr int LDS_HELD = RINGN * PHB, LDS_STAGE = LDS_HELD + 16 * HS * 4, LDS_TOTAL = LDS_STAGE + 16 * HS2 * 4;
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
#if LAY16
      const u64 d = D[((((i64)a * 2 * NSL + (j >> 4)) * CT + ct) * NCOL + k) * 16 + (j & 15)];
#else
      const u64 d = D[((((i64)a * NSL + (j >> 5)) * CT + ct) * NCOL + k) * 32 + (j & 31)];
#endif
      acc = (acc + (d % p) * K[(((i64)a * COLS + col) * NCOL + k) * N + j]) % p;
    }
    ref[g] = (u32)acc;
  }
}



__device__ __forceinline__ void dma16(u32 voff, const void* sbase, u32 lds_dst) {      // 64 lanes x 16 bytes -> LDS [lds_dst, + 1024)
  u32 keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(sbase) : "memory");
}
template <int IMM> __device__ __forceinline__ v4i ldB(u32 voff, const void* sbase) {
  v4i r;
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(r) : "v"(voff), "s"(sbase), "n"(IMM));
  return r;
}
__device__ __forceinline__ u32 ldC(u32 voff, const void* sbase) {
  u32 r;
  asm volatile("global_load_dword %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sbase));
  return r;
}
template <class T> __device__ __forceinline__ const T* uni(const T* p) {      // a pointer the compiler must keep in scalar registers
  const u64 v = (u64)p;
  return (const T*)(((u64)(u32)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)v));
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// NW waves; SL = 16 / NW coefficients per wave; PPW = 16 / NW DMA pieces per wave and phase
template <int NW>
__global__ void __launch_bounds__(NW * 64, 1) dot_mfma4_kernel(const u32* __restrict__ D, const v4i* __restrict__ Bt, const u32* __restrict__ corr, u32* __restrict__ out, const u32* __restrict__ zpage, int CT, Consts cs) {
  constexpr int SL = 16 / NW, PPW = 16 / NW;
  constexpr int NB = SL * 4 /* key operand loads per phase */, NC = SL /* offset-term loads at s = 7 */;
  extern __shared__ __attribute__((aligned(16))) u32 lds[];
  const u32 lds0 = (u32)(size_t)lds;
  u32* const held = lds + LDS_HELD / 4;
  u32* const stage = lds + LDS_STAGE / 4;
  const u32 tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int G = CT >> 5;
  const int nitems = (4 * NSL / 8) * G;
  // DMA pieces: piece pi = w * PPW + i brings column k = pi >> 1 of the phase for the ciphertexts 16 (pi & 1) + r
  u32 dmaoff[PPW];
  {
    const u32 r = lane >> 2, jx = lane & 3, j4 = jx ^ ((r >> 2) & 3);
#pragma unroll
    for (int i = 0; i < PPW; ++i) { const int pi = w * PPW + i; dmaoff[i] = ((16 * (pi & 1) + r) * NCOL + (pi >> 1)) * (LAY16 ? 64 : 128) + j4 * 16; }
  }
  const u32 l16 = lane * 16;
  const u32 ct_l = lane & 31, kh = lane >> 5;
  const int j0 = w * SL;                                      // first coefficient of the wave inside the tile
  const u32 rdoff = ((4 * kh * 2 + (ct_l >> 4)) * 64 + (ct_l & 15) * 4 + ((u32)(j0 >> 2) ^ ((ct_l >> 2) & 3))) * 16 + (j0 & 3) * 4;
  const int col = (int)(lane & 31);
  struct Item { const char* dA; const char* bB; const char* cC; int a, S, g; };
  auto item_of = [&](int it) {
    Item r;
    const int ql = it / G;
    r.g = it - ql * G;
    const int q = ql * 8 + xcd;
    r.a = q / NSL; r.S = q % NSL;
#if LAY16
    r.dA = uni((const char*)(D + (((i64)2 * q * CT + r.g * 32) * NCOL) * 16));
#else
    r.dA = uni((const char*)(D + (((i64)q * CT + r.g * 32) * NCOL) * 32));
#endif
    r.bB = uni((const char*)(Bt + ((i64)r.a * N + r.S * 32 + j0) * (NS * 4 * 64)));
    r.cC = uni((const char*)(corr + ((i64)r.a * N + r.S * 32 + j0) * 32));
    return r;
  };
  if (slot >= nitems) return;
  Item cur = item_of(slot), nxt = cur;
  v4i Bb[2][SL][4];
  v16i acc[SL][4];
  u32 res[SL][16];
  u32 pr = 0, mu = 0, cc2[SL], pcb = 0;
#pragma unroll
  for (int x = 0; x < SL; ++x) cc2[x] = 0;
#define IC(v) std::integral_constant<int, (v)>{}
  auto issueA = [&](u32 ringslot, const char* abase, auto PH) {
    constexpr int ph = decltype(PH)::value, h = ph / NS, s = ph % NS;
    const u32 dst = lds0 + ringslot * PHB + w * PPW * 1024;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const bool real = s < NS - 1 || ((w * PPW + i) >> 1) < 2;            // the last phase holds columns 64, 65 only
      if (real && !(HOT & 2)) {
#if LAY16
        const char* ab = abase + s * 512 + (i64)h * CT * NCOL * 64;
#else
        const char* ab = abase + s * 1024 + h * 64;
#endif
        dma16(dmaoff[i], ab, dst + i * 1024);
      } else dma16(l16, zpage, dst + i * 1024);
    }
  };
  auto issueB = [&](auto SET, const char* bbase, auto PH) {
    constexpr int set = decltype(SET)::value, ph = decltype(PH)::value, h = ph / NS, s = ph % NS;
#pragma unroll
    for (int x = 0; x < SL; ++x) {
      const char* sb = (HOT & 1) ? (const char*)Bt + (x * NS * 4 + s * 4) * 1024 : bbase + ((h * 16 + x) * NS * 4 + s * 4) * 1024;
      Bb[set][x][0] = ldB<0>(l16, sb); Bb[set][x][1] = ldB<1024>(l16, sb); Bb[set][x][2] = ldB<2048>(l16, sb); Bb[set][x][3] = ldB<3072>(l16, sb);
    }
  };
  auto waitB = [&](auto SET, auto WITHC, auto SLC) {
    constexpr int set = decltype(SET)::value;
    constexpr int NWAIT = PPW + NB + PPW;                      // younger requests: the pieces of the previous phase, this phase's key operands and pieces
    if constexpr (decltype(SLC)::value == 2) {
      if constexpr (decltype(WITHC)::value) asm volatile("s_waitcnt vmcnt(%10)" : "+v"(Bb[set][0][0]), "+v"(Bb[set][0][1]), "+v"(Bb[set][0][2]), "+v"(Bb[set][0][3]), "+v"(Bb[set][1][0]), "+v"(Bb[set][1][1]), "+v"(Bb[set][1][2]), "+v"(Bb[set][1][3]), "+v"(cc2[0]), "+v"(cc2[SL - 1]) : "n"(NWAIT) : "memory");
      else asm volatile("s_waitcnt vmcnt(%8)" : "+v"(Bb[set][0][0]), "+v"(Bb[set][0][1]), "+v"(Bb[set][0][2]), "+v"(Bb[set][0][3]), "+v"(Bb[set][1][0]), "+v"(Bb[set][1][1]), "+v"(Bb[set][1][2]), "+v"(Bb[set][1][3]) : "n"(NWAIT) : "memory");
    } else {
      if constexpr (decltype(WITHC)::value) asm volatile("s_waitcnt vmcnt(%5)" : "+v"(Bb[set][0][0]), "+v"(Bb[set][0][1]), "+v"(Bb[set][0][2]), "+v"(Bb[set][0][3]), "+v"(cc2[0]) : "n"(NWAIT) : "memory");
      else asm volatile("s_waitcnt vmcnt(%4)" : "+v"(Bb[set][0][0]), "+v"(Bb[set][0][1]), "+v"(Bb[set][0][2]), "+v"(Bb[set][0][3]) : "n"(NWAIT) : "memory");
    }
  };
  auto step = [&](auto PH, auto SLC) {
    constexpr int ph = decltype(PH)::value, h = ph / NS, s = ph % NS, set = ph & 1;
    // a. key operands of the next phase, the offset terms one phase before they are used, digit words three phases ahead
    if constexpr (ph + 1 < 2 * NS) issueB(IC((ph + 1) & 1), cur.bB, IC(ph + 1)); else issueB(IC((ph + 1) & 1), nxt.bB, IC(0));
    if constexpr (s == NS - 2) {
#pragma unroll
      for (int x = 0; x < SL; ++x) cc2[x] = ldC((u32)col * 4, cur.cC + (h * 16 + x) * 128);
    }
    if constexpr (ph + 3 < 2 * NS) issueA((pcb + ph + 3) & 3, cur.dA, IC(ph + 3)); else issueA((pcb + ph + 3) & 3, nxt.dA, IC(ph + 3 - 2 * NS));
    // b. the A operands of this phase
    const char* rb = (const char*)lds + ((pcb + ph) & 3) * PHB + rdoff;
    v4i av[SL];
    if constexpr (decltype(SLC)::value == 2) {
      uint2 g[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) g[i] = *reinterpret_cast<const uint2*>(rb + i * 2048);
      av[0] = v4i{(int)(g[0].x ^ 0x80808080u), (int)(g[1].x ^ 0x80808080u), (int)(g[2].x ^ 0x80808080u), (int)(g[3].x ^ 0x80808080u)};
      av[SL - 1] = v4i{(int)(g[0].y ^ 0x80808080u), (int)(g[1].y ^ 0x80808080u), (int)(g[2].y ^ 0x80808080u), (int)(g[3].y ^ 0x80808080u)};
    } else {
      u32 g[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) g[i] = *reinterpret_cast<const u32*>(rb + i * 2048);
      av[0] = v4i{(int)(g[0] ^ 0x80808080u), (int)(g[1] ^ 0x80808080u), (int)(g[2] ^ 0x80808080u), (int)(g[3] ^ 0x80808080u)};
    }
    // c. the key operands requested one phase ago (the offset terms, requested right behind them, come with the same wait)
    waitB(IC(set), std::integral_constant<bool, s == NS - 1>{}, IC(SL));
#pragma unroll
    for (int x = 0; x < SL; ++x)
#pragma unroll
      for (int bq = 0; bq < 4; ++bq) {
        if constexpr (s == 0) { const v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; acc[x][bq] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[x], Bb[set][x][bq], z, 0, 0, 0); }
        else acc[x][bq] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[x], Bb[set][x][bq], acc[x][bq], 0, 0, 0);
      }
    if constexpr (s == NS - 1) {
      // d. four partial sums per output -> residue below 2p.  First half tile: into `held`; second: kept, the columns below 15 staged now
#pragma unroll
      for (int x = 0; x < SL; ++x) {
        const u64 C64 = ((u64)pr << 18) + cc2[x];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int u = acc[x][0][r] + acc[x][1][r] * 256, v = acc[x][2][r] + acc[x][3][r] * 256;
          const u64 T = (u64)((i64)v * 65536 + (i64)u + (i64)C64);
          const u32 qh = __umulhi((u32)(T >> 18), mu);
          const u32 rv = (u32)T - qh * pr;
          const int ct = 8 * (r >> 2) + (r & 3);                 // + 4 (lane >> 5)
          if constexpr (h == 0) { if (col < COLS) held[(j0 + x) * HS + (ct + 4 * (int)(lane >> 5)) * COLS + col] = rv; }
          else { res[x][r] = rv; if (col < COLS / 2) stage[(j0 + x) * HS2 + (ct + 4 * (int)(lane >> 5)) * (COLS / 2) + col] = rv; }
        }
      }
    }
    // e. this wave's pieces of the next phase (requested two phases ago) have landed; then everybody's have
    wait_vm<NB + PPW + NB + PPW>();
    __syncthreads();
    if constexpr (ph == 2 * NS - 1) {
      // f. the 960 rows of the tile pair, 128 bytes each, in two passes of 15 columns: a wave takes 32 / NW ciphertexts, its lower lanes the
      //    even ones; coefficients 0..15 from `held`, 16..31 from the staged half
      constexpr int CPW = 32 / NW;                               // ciphertexts per wave: 4 or 2
      const u32 jj = lane & 31, r2 = lane >> 5;
      const int ctb = CPW * w + (CPW / 2) * (int)r2;
      u32* const obase = out + (((i64)cur.g * 32 + ctb) * 2 * NLB * 4 + cur.a << LOGN) + cur.S * 32 + jj;
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
          __syncthreads();
#pragma unroll
          for (int x = 0; x < SL; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int ct = 8 * (r >> 2) + (r & 3);
              if (col >= COLS / 2 && col < COLS) stage[(j0 + x) * HS2 + (ct + 4 * (int)(lane >> 5)) * (COLS / 2) + col - COLS / 2] = res[x][r];
            }
          __syncthreads();
        }
        const u32* src = jj < 16 ? held + jj * HS + ctb * COLS + pass * (COLS / 2) : stage + (jj - 16) * HS2 + ctb * (COLS / 2);
        const u32 sct = jj < 16 ? COLS : COLS / 2;
#pragma unroll
        for (int il = 0; il < CPW / 2; ++il)
#pragma unroll
          for (int c = 0; c < COLS / 2; ++c) {
            const int cc = pass * (COLS / 2) + c;
            const u32 v = src[il * sct + c];
            if (!(HOT & 4) || v == 0x12345678u) __builtin_nontemporal_store(v, obase + ((i64)((il * 2 + (cc & 1)) * NLB + (cc >> 1)) * 4 << LOGN));
          }
      }
    }
  };
  // prologue: phases 0, 1, 2 and the key operands of phase 0
  issueB(IC(0), cur.bB, IC(0));
  issueA(0, cur.dA, IC(0)); issueA(1, cur.dA, IC(1)); issueA(2, cur.dA, IC(2));
  wait_vm<0>();
  __syncthreads();
  for (int it = slot; it < nitems; it += nslot) {
    nxt = it + nslot < nitems ? item_of(it + nslot) : cur;
    pr = cs.p[cur.a]; mu = cs.mu50[cur.a];
    step(IC(0), IC(SL)); step(IC(1), IC(SL)); step(IC(2), IC(SL)); step(IC(3), IC(SL)); step(IC(4), IC(SL)); step(IC(5), IC(SL)); step(IC(6), IC(SL)); step(IC(7), IC(SL)); step(IC(8), IC(SL));
    step(IC(9), IC(SL)); step(IC(10), IC(SL)); step(IC(11), IC(SL)); step(IC(12), IC(SL)); step(IC(13), IC(SL)); step(IC(14), IC(SL)); step(IC(15), IC(SL)); step(IC(16), IC(SL)); step(IC(17), IC(SL));
    cur = nxt;
    pcb = (pcb + 2 * NS) & 3;
  }
  wait_vm<0>();
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
  const size_t shmem = LDS_TOTAL;
  u32* zpage; CK(hipMalloc(&zpage, 4096)); CK(hipMemset(zpage, 0, 4096));
  CK(hipFuncSetAttribute((const void*)dot_mfma4_kernel<NWAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  printf("LDS %zu bytes, grid %d\n", shmem, ncu);
  dot_mfma4_kernel<NWAVES><<<ncu, NWAVES * 64, shmem>>>(D, Bt, corr, out, zpage, CT, cs); CK(hipDeviceSynchronize());
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
  for (int r = 0; r < reps; ++r) dot_mfma4_kernel<NWAVES><<<ncu, NWAVES * 64, shmem>>>(D, Bt, corr, out, zpage, CT, cs);
  hipEventRecord(e1); CK(hipEventSynchronize(e1));
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)nD * 4 + (double)nO * 4 + (double)nB * 16 + (double)nC * 4;
  printf("dot_mfma4: %.3f ms per launch of %d ciphertexts; %.1f GB moved at least -> %.0f GB/s; %.1f T multiply-adds/s\n", ms / reps, CT, bytes / 1e9, bytes / (ms / reps * 1e-3) / 1e9,
         (double)CT * COLS * NCOL * 4 * N / (ms / reps * 1e-3) / 1e12);
  return bad != 0;
}
