// kernels_ntt.hip -- negacyclic number-theoretic transform for power-of-two m (n = phi(m) = m/2) on gfx950.
//
// Replaces Cmod::FFT / Cmod::iFFT (CModulus.cpp:90-107, 110-132) + tBluesteinFFT (bluestein.cpp:93-144) for
// m = 2^k: there Cmod::FFT is exactly  y[j] = sum_k a_k psi^{(2j+1)k} mod q,  psi = root^2, natural order in and out
// (SURVEY.md fact 5), and Cmod::iFFT its inverse (scatter to odd exponents, DFT with root^-1, /m, mod X^n+1).
//
// Arithmetic: 64-bit residues, Shoup/Harvey lazy butterflies (values kept in [0,4q) forward, [0,2q) inverse),
// canonical [0,q) on store.  No MFMA: this is integer NTT.
//
// Two implementations:
//   * ntt_*_tile: the tuned path, n = 2^11..2^14.  One workgroup per row, n/32 threads, 32 residues per thread in
//     registers, radix-32 register passes (5+5+rest stages) separated by two LDS exchanges; coalesced row loads/stores,
//     the bit-reversal permutation is absorbed by the exchange addressing.
//   * n = 2^15..2^17 (stress config, Bluestein convolution sizes): two passes, 2^(logn-14) tile sub-transforms per row and a
//     tail kernel for the remaining stages (ntt_*_tail below, TileBig in ntt_tile.inc).
//   * ntt_*_lds: generic radix-2 LDS path for every other power of two (up to 2^14 residues per block; rows above 2^17 run
//     their outer stages through global-memory stage kernels first).
#include "fhesi_internal.h"

#define NTT_LDS_MAX_LOG 14

__device__ __forceinline__ u32 brv_bits(u32 x, int bits) { return __brev(x) >> (32 - bits); }

// ------------------------------------------------------------------------------------------ lazy butterflies
// forward (Cooley-Tukey, Harvey): X,Y in [0,4q) -> [0,4q)
__device__ __forceinline__ void bfly_fwd(u64& X, u64& Y, u64 w, u64 wp, u64 q, u64 two_q) {
  u64 x = X >= two_q ? X - two_q : X;
  u64 t = d_shoup_lazy(Y, w, wp, q);
  X = x + t;
  Y = x - t + two_q;
}
// inverse (Gentleman-Sande): X,Y in [0,2q) -> [0,2q)
__device__ __forceinline__ void bfly_inv(u64& X, u64& Y, u64 w, u64 wp, u64 q, u64 two_q) {
  u64 s = X + Y;
  u64 d = X - Y + two_q;
  X = s >= two_q ? s - two_q : s;
  Y = d_shoup_lazy(d, w, wp, q);
}
__device__ __forceinline__ u64 norm4(u64 v, u64 q, u64 two_q) {
  if (v >= two_q) v -= two_q;
  if (v >= q) v -= q;
  return v;
}
__device__ __forceinline__ u64 norm2(u64 v, u64 q) { return v >= q ? v - q : v; }

// ------------------------------------------------------------------------------------------ generic LDS kernels
// One block handles one contiguous sub-block of 2^logb residues of a row of 2^logn; tw_mul = 2^(logn-logb) + subblock
// selects the twiddles of the remaining stages (twiddle index = mm_local * tw_mul + i_local).
template <bool BITREV>
__global__ void __launch_bounds__(1024) ntt_fwd_lds(u64* __restrict__ rows, int logn, int logb, int nslots, const int* __restrict__ prime_of_slot,
                                                     const PrimeConst* __restrict__ pcs, const Shoup2* __restrict__ tw_all, u64 skip_q) {
  extern __shared__ __attribute__((aligned(16))) u64 s[];
  const int nb = 1 << logb, T = blockDim.x, tid = threadIdx.x;
  const int sub_per_row = 1 << (logn - logb);
  const i64 row = blockIdx.x / sub_per_row;
  const int sb = blockIdx.x % sub_per_row;
  const int slot = (int)(row % nslots);
  const int prime = prime_of_slot ? prime_of_slot[slot] : slot;
  const PrimeConst pc = pcs[prime];
  if (skip_q && pc.q >= skip_q) return;       // this row belongs to the tile kernel
  const u64 q = pc.q, two_q = pc.two_q;
  const Shoup2* tw = tw_all + ((i64)prime << logn);
  u64* g = rows + (row << logn) + ((i64)sb << logb);
  const u32 tw_mul = (u32)sub_per_row + sb;

  for (int i = tid; i < nb; i += T) s[i] = g[i];
  __syncthreads();
  int logt = logb;
  for (int mm = 1; mm < nb; mm <<= 1) {
    --logt;
    const int t = 1 << logt;
    for (int b = tid; b < (nb >> 1); b += T) {
      const int i = b >> logt, j = ((i << 1) << logt) + (b & (t - 1));
      const Shoup2 w = tw[(u32)mm * tw_mul + i];
      u64 X = s[j], Y = s[j + t];
      bfly_fwd(X, Y, w.w, w.wp, q, two_q);
      s[j] = X;
      s[j + t] = Y;
    }
    __syncthreads();
  }
  if (BITREV) {   // full rows only (logb == logn): natural-order output y[j] = a[brv(j)]
    for (int j = tid; j < nb; j += T) g[j] = norm4(s[brv_bits(j, logb)], q, two_q);
  } else {
    for (int j = tid; j < nb; j += T) g[j] = norm4(s[j], q, two_q);
  }
}

template <bool BITREV>
__global__ void __launch_bounds__(1024) ntt_inv_lds(u64* __restrict__ rows, int logn, int logb, int nslots, const int* __restrict__ prime_of_slot,
                                                     const PrimeConst* __restrict__ pcs, const Shoup2* __restrict__ tw_all, u64 skip_q) {
  extern __shared__ __attribute__((aligned(16))) u64 s[];
  const int nb = 1 << logb, T = blockDim.x, tid = threadIdx.x;
  const int sub_per_row = 1 << (logn - logb);
  const i64 row = blockIdx.x / sub_per_row;
  const int sb = blockIdx.x % sub_per_row;
  const int slot = (int)(row % nslots);
  const int prime = prime_of_slot ? prime_of_slot[slot] : slot;
  const PrimeConst pc = pcs[prime];
  if (skip_q && pc.q >= skip_q) return;       // this row belongs to the tile kernel
  const u64 q = pc.q, two_q = pc.two_q;
  const Shoup2* tw = tw_all + ((i64)prime << logn);
  u64* g = rows + (row << logn) + ((i64)sb << logb);
  const u32 tw_mul = (u32)sub_per_row + sb;
  const bool full = (logb == logn);

  if (BITREV) { for (int j = tid; j < nb; j += T) s[brv_bits(j, logb)] = g[j]; }
  else        { for (int j = tid; j < nb; j += T) s[j] = g[j]; }
  __syncthreads();
  int logt = 0;
  for (int mm = nb; mm > 1; mm >>= 1) {
    const int h = mm >> 1, t = 1 << logt;
    const bool last = full && (h == 1);
    for (int b = tid; b < (nb >> 1); b += T) {
      const int i = b >> logt, j = ((i << 1) << logt) + (b & (t - 1));
      u64 X = s[j], Y = s[j + t];
      if (!last) {
        const Shoup2 w = tw[(u32)h * tw_mul + i];
        bfly_inv(X, Y, w.w, w.wp, q, two_q);
      } else {   // final stage folded with the 1/n scaling (the /m of CModulus.cpp:125)
        u64 sum = X + Y, d = X - Y + two_q;
        X = d_shoup_lazy(sum, pc.ninv, pc.ninv_sh, q);
        Y = d_shoup_lazy(d, pc.ninv_w, pc.ninv_w_sh, q);
      }
      s[j] = X;
      s[j + t] = Y;
    }
    __syncthreads();
    ++logt;
  }
  for (int j = tid; j < nb; j += T) g[j] = norm2(s[j], q);
}

// outer stages of rows larger than one LDS block: one butterfly per thread straight on HBM
__global__ void __launch_bounds__(256) ntt_fwd_global_stage(u64* __restrict__ rows, int logn, int stage /* mm = 2^stage */, int nslots,
                                                             const int* __restrict__ prime_of_slot, const PrimeConst* __restrict__ pcs,
                                                             const Shoup2* __restrict__ tw_all, i64 total_bfly) {
  i64 gid = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total_bfly) return;
  const i64 row = gid >> (logn - 1);
  const u32 b = (u32)(gid & ((1ll << (logn - 1)) - 1));
  const int slot = (int)(row % nslots);
  const int prime = prime_of_slot ? prime_of_slot[slot] : slot;
  const PrimeConst pc = pcs[prime];
  const int logt = logn - 1 - stage;
  const u32 t = 1u << logt, i = b >> logt, j = ((i << 1) << logt) + (b & (t - 1));
  const Shoup2 w = tw_all[((i64)prime << logn) + (1u << stage) + i];
  u64* g = rows + (row << logn);
  u64 X = g[j], Y = g[j + t];
  bfly_fwd(X, Y, w.w, w.wp, pc.q, pc.two_q);
  g[j] = X;       // stays lazy in [0,4q); the LDS pass normalises
  g[j + t] = Y;
}
__global__ void __launch_bounds__(256) ntt_inv_global_stage(u64* __restrict__ rows, int logn, int stage /* h = 2^stage */, int nslots,
                                                             const int* __restrict__ prime_of_slot, const PrimeConst* __restrict__ pcs,
                                                             const Shoup2* __restrict__ tw_all, i64 total_bfly) {
  i64 gid = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total_bfly) return;
  const i64 row = gid >> (logn - 1);
  const u32 b = (u32)(gid & ((1ll << (logn - 1)) - 1));
  const int slot = (int)(row % nslots);
  const int prime = prime_of_slot ? prime_of_slot[slot] : slot;
  const PrimeConst pc = pcs[prime];
  const int logt = logn - 1 - stage;
  const u32 t = 1u << logt, i = b >> logt, j = ((i << 1) << logt) + (b & (t - 1));
  u64* g = rows + (row << logn);
  u64 X = g[j], Y = g[j + t];
  if (stage > 0) {
    const Shoup2 w = tw_all[((i64)prime << logn) + (1u << stage) + i];
    bfly_inv(X, Y, w.w, w.wp, pc.q, pc.two_q);
    g[j] = norm2(X, pc.q);
    g[j + t] = norm2(Y, pc.q);
  } else {
    u64 sum = X + Y, d = X - Y + pc.two_q;
    g[j] = norm2(d_shoup_lazy(sum, pc.ninv, pc.ninv_sh, pc.q), pc.q);
    g[j + t] = norm2(d_shoup_lazy(d, pc.ninv_w, pc.ninv_w_sh, pc.q), pc.q);
  }
}
__global__ void __launch_bounds__(256) bitrev_rows(const u64* __restrict__ src, u64* __restrict__ dst, int logn, i64 total) {
  i64 gid = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const i64 row = gid >> logn;
  const u32 j = (u32)(gid & ((1ll << logn) - 1));
  dst[gid] = src[(row << logn) + brv_bits(j, logn)];
}

// ------------------------------------------------------------------------------------------ tuned register-tile kernels
#include "ntt_tile.inc"

// ------------------------------------------------------------------------------------------ tails of the two-pass transforms
// Rows of n = 2^(14+S0): the S0 stages the 2^14-point tile sub-transforms leave over.  One thread owns the 2^S0 residues
// {j1 + 2^14 h}; loads and stores are contiguous across j1, rows are taken prime-major so the tail twiddles stay in L2.
//
// Forward (decimation in time).  F(M,c) = negacyclic transform of a[i*(n/M)+c] with root psi^(n/M):
//   F(M,c)[j]       = F(M/2,c)[j] + psi_M^(2j+1) F(M/2,c+n/M)[j]
//   F(M,c)[j + M/2] = F(M/2,c)[j] - psi_M^(2j+1) F(M/2,c+n/M)[j]          j < M/2
// src holds F(2^14,k2) at [row][k2][j1]; stage s = 1..S0 builds M = 2^(14+s).  tail_tw: per prime, stage s at offset
// 2^14 (2^(s-1) - 1), entry j = psi_M^(2j+1), j < M/2.  src == dst is allowed (a thread reads what it writes).
template <int S0>
__global__ void __launch_bounds__(256) ntt_fwd_tail(const u64* src, u64* dst, i64 count, int nslots, int slot0,
                                                    const int* __restrict__ prime_of_slot, const PrimeConst* __restrict__ pcs,
                                                    const Shoup2* __restrict__ tail_tw) {
  constexpr int N2 = 1 << S0, LOGN = 14 + S0;
  const u32 rb = blockIdx.x >> 6;
  const u32 j1 = ((blockIdx.x & 63) << 8) | threadIdx.x;
  const int slot = (int)(rb / (u32)count) + slot0;
  const i64 row = (i64)(rb % (u32)count) * nslots + slot;
  const int prime = prime_of_slot ? prime_of_slot[slot] : slot;
  const u64 q = pcs[prime].q, two_q = pcs[prime].two_q;
  const Shoup2* __restrict__ tw = tail_tw + ((i64)prime << LOGN) + j1;
  const i64 base = (row << LOGN) + j1;
  u64 v[N2];
#pragma unroll
  for (int k = 0; k < N2; ++k) v[k] = src[base + ((i64)k << 14)];
#pragma unroll
  for (int s = 1; s <= S0; ++s) {
    const int C = 1 << (S0 - s + 1), H = 1 << (s - 1);     // v[h*C + c] = F(2^(13+s), c)[j1 + 2^14 h]
    u64 nv[N2];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const Shoup2 w = tw[(i64)(H - 1 + h) << 14];
#pragma unroll
      for (int c = 0; c < C / 2; ++c) {
        u64 X = v[h * C + c], Y = v[h * C + c + C / 2];
        bfly_fwd(X, Y, w.w, w.wp, q, two_q);
        nv[h * (C / 2) + c] = X;
        nv[(h + H) * (C / 2) + c] = Y;
      }
    }
#pragma unroll
    for (int k = 0; k < N2; ++k) v[k] = nv[k];
  }
#pragma unroll
  for (int h = 0; h < N2; ++h) dst[base + ((i64)h << 14)] = norm4(v[h], q, two_q);
}

// Inverse: stages S0-1 .. 0 of the Gentleman-Sande network (partner distance 2^(14+S0-1-s), twiddle tw_inv[2^s + block],
// uniform per thread); the 1/n scaling was folded into the sub-transforms' last stage.
template <int S0>
__global__ void __launch_bounds__(256) ntt_inv_tail(const u64* src, u64* dst, i64 count, int nslots,
                                                    const int* __restrict__ prime_of_slot, const PrimeConst* __restrict__ pcs,
                                                    const Shoup2* __restrict__ tw_all) {
  constexpr int N2 = 1 << S0, LOGN = 14 + S0;
  const u32 rb = blockIdx.x >> 6;
  const u32 j1 = ((blockIdx.x & 63) << 8) | threadIdx.x;
  const int slot = (int)(rb / (u32)count);
  const i64 row = (i64)(rb % (u32)count) * nslots + slot;
  const int prime = prime_of_slot ? prime_of_slot[slot] : slot;
  const u64 q = pcs[prime].q, two_q = pcs[prime].two_q;
  const Shoup2* __restrict__ tw = tw_all + ((i64)prime << LOGN);
  const i64 base = (row << LOGN) + j1;
  u64 v[N2];
#pragma unroll
  for (int h = 0; h < N2; ++h) v[h] = src[base + ((i64)h << 14)];
#pragma unroll
  for (int s = S0 - 1; s >= 0; --s) {
    const int dist = 1 << (S0 - 1 - s);
#pragma unroll
    for (int h = 0; h < N2; ++h) {
      if (h & dist) continue;
      const Shoup2 w = tw[(1 << s) + (h >> (S0 - s))];
      bfly_inv(v[h], v[h + dist], w.w, w.wp, q, two_q);
    }
  }
#pragma unroll
  for (int h = 0; h < N2; ++h) dst[base + ((i64)h << 14)] = norm2(v[h], q);
}

// Head of the order-free forward transform: stages 0 .. S0-1 of the Cooley-Tukey network on the natural-order row (partner
// distance 2^(14+S0-1-s), twiddle tw_fwd[2^s + block], uniform per thread), in place; values stay lazy in [0,4q) for the
// sub-transforms.  Mirror image of ntt_inv_tail.
template <int S0>
__global__ void __launch_bounds__(256) ntt_fwd_head(u64* rows, i64 count, int nslots, const int* __restrict__ prime_of_slot, const PrimeConst* __restrict__ pcs,
                                                    const Shoup2* __restrict__ tw_all) {
  constexpr int N2 = 1 << S0, LOGN = 14 + S0;
  const u32 rb = blockIdx.x >> 6;
  const u32 j1 = ((blockIdx.x & 63) << 8) | threadIdx.x;
  const int slot = (int)(rb / (u32)count);
  const i64 row = (i64)(rb % (u32)count) * nslots + slot;
  const int prime = prime_of_slot ? prime_of_slot[slot] : slot;
  const u64 q = pcs[prime].q, two_q = pcs[prime].two_q;
  const Shoup2* __restrict__ tw = tw_all + ((i64)prime << LOGN);
  const i64 base = (row << LOGN) + j1;
  u64 v[N2];
#pragma unroll
  for (int h = 0; h < N2; ++h) v[h] = rows[base + ((i64)h << 14)];
#pragma unroll
  for (int s = 0; s < S0; ++s) {
    const int dist = 1 << (S0 - 1 - s);
#pragma unroll
    for (int h = 0; h < N2; ++h) {
      if (h & dist) continue;
      const Shoup2 w = tw[(1 << s) + (h >> (S0 - s))];
      bfly_fwd(v[h], v[h + dist], w.w, w.wp, q, two_q);
    }
  }
#pragma unroll
  for (int h = 0; h < N2; ++h) rows[base + ((i64)h << 14)] = v[h];
}

bool ntt_digits_suborder(const fhesi_ctx* ctx, int digit_bits) { return ctx->pow2 && ctx->logn == 15 && digit_bits < 32; }
bool ntt_orderfree_two_pass(const fhesi_ctx* ctx) { return ctx->pow2 && ntt_tile2_supported(ctx->logn); }
int launch_ntt_fwd_head(fhesi_ctx* ctx, u64* d_rows, i64 count, int nslots, const int* d_pos) {
  const unsigned grid = (unsigned)(count * nslots) << 6;
  if (!grid) return 0;
  switch (ctx->logn - 14) {
    case 1: ntt_fwd_head<1><<<grid, 256, 0, ctx->stream>>>(d_rows, count, nslots, d_pos, ctx->d_pc, ctx->d_tw_fwd); break;
    case 2: ntt_fwd_head<2><<<grid, 256, 0, ctx->stream>>>(d_rows, count, nslots, d_pos, ctx->d_pc, ctx->d_tw_fwd); break;
    default: ntt_fwd_head<3><<<grid, 256, 0, ctx->stream>>>(d_rows, count, nslots, d_pos, ctx->d_pc, ctx->d_tw_fwd); break;
  }
  HIP_TRY(hipGetLastError());
  return 0;
}
int launch_ntt_sub(fhesi_ctx* ctx, bool fwd, u64* d_rows, i64 count, int nslots, const int* d_pos) {
  if (!(count * nslots)) return 0;
  ProfScope prof(ctx, fwd ? PROF_NTT_FWD : PROF_NTT_INV, (double)(count * nslots));
  return launch_tile_sub(ctx, fwd, d_rows, count * nslots, nslots, d_pos);
}

static int launch_fwd_tail(fhesi_ctx* ctx, const u64* src, u64* dst, i64 count, int nslots, int slot0, int nslot_launch, const int* d_pos) {
  const unsigned grid = (unsigned)(count * nslot_launch) << 6;
  switch (ctx->logn - 14) {
    case 1: ntt_fwd_tail<1><<<grid, 256, 0, ctx->stream>>>(src, dst, count, nslots, slot0, d_pos, ctx->d_pc, ctx->d_tail_fwd); break;
    case 2: ntt_fwd_tail<2><<<grid, 256, 0, ctx->stream>>>(src, dst, count, nslots, slot0, d_pos, ctx->d_pc, ctx->d_tail_fwd); break;
    default: ntt_fwd_tail<3><<<grid, 256, 0, ctx->stream>>>(src, dst, count, nslots, slot0, d_pos, ctx->d_pc, ctx->d_tail_fwd); break;
  }
  HIP_TRY(hipGetLastError());
  return 0;
}
static int launch_inv_tail(fhesi_ctx* ctx, const u64* src, u64* dst, i64 count, int nslots, const int* d_pos) {
  const unsigned grid = (unsigned)(count * nslots) << 6;
  switch (ctx->logn - 14) {
    case 1: ntt_inv_tail<1><<<grid, 256, 0, ctx->stream>>>(src, dst, count, nslots, d_pos, ctx->d_pc, ctx->d_tw_inv); break;
    case 2: ntt_inv_tail<2><<<grid, 256, 0, ctx->stream>>>(src, dst, count, nslots, d_pos, ctx->d_pc, ctx->d_tw_inv); break;
    default: ntt_inv_tail<3><<<grid, 256, 0, ctx->stream>>>(src, dst, count, nslots, d_pos, ctx->d_pc, ctx->d_tw_inv); break;
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_ntt_inv_tail_inplace(fhesi_ctx* ctx, u64* d_rows, i64 count, int nslots, const int* d_pos) {
  if (!(count * nslots)) return 0;
  return launch_inv_tail(ctx, d_rows, d_rows, count, nslots, d_pos);
}

// ------------------------------------------------------------------------------------------ launchers
static int lds_threads(int logb) { int t = 1 << (logb > 0 ? logb - 1 : 0); return t > 1024 ? 1024 : (t < 64 ? 64 : t); }

int launch_ntt_fwd(fhesi_ctx* ctx, u64* d_rows, i64 count, int nslots, const int* d_prime_of_slot, bool bitrev) {
  const int logn = ctx->logn;
  const i64 nrows = count * nslots;
  if (nrows == 0) return 0;
  ProfScope prof(ctx, PROF_NTT_FWD, (double)nrows);
  u64 skip_q = 0;
  if (bitrev && ntt_tile_supported(logn)) {
    return launch_ntt_fwd_tile(ctx, d_rows, nrows, nslots, d_prime_of_slot);     // (rows of small primes: its EXACT instantiation)
  }
  if (!bitrev && ntt_tile2_supported(logn)) {      // order-free (convolutions): head stages, then in-place sub-transforms
    FHESI_TRY(launch_ntt_fwd_head(ctx, d_rows, count, nslots, d_prime_of_slot));
    return launch_tile_sub(ctx, true, d_rows, nrows, nslots, d_prime_of_slot);
  }
  if (bitrev && ntt_tile2_supported(logn)) {
    void* tmp;
    FHESI_TRY(ws_reserve(ctx, 6, (size_t)nrows << (logn + 3), &tmp));
    FHESI_TRY(launch_tile_big(ctx, true, d_rows, (u64*)tmp, nrows, nslots, d_prime_of_slot));
    return launch_fwd_tail(ctx, (const u64*)tmp, d_rows, count, nslots, 0, nslots, d_prime_of_slot);
  }
  const int logb = logn > NTT_LDS_MAX_LOG ? NTT_LDS_MAX_LOG : logn;
  for (int st = 0; st < logn - logb; ++st) {
    i64 total = nrows << (logn - 1);
    ntt_fwd_global_stage<<<(unsigned)((total + 255) / 256), 256, 0, ctx->stream>>>(d_rows, logn, st, nslots, d_prime_of_slot, ctx->d_pc, ctx->d_tw_fwd, total);
  }
  const size_t shmem = sizeof(u64) << logb;
  const unsigned grid = (unsigned)(nrows << (logn - logb));
  if (bitrev && logb == logn) {
    HIP_TRY(hipFuncSetAttribute((const void*)ntt_fwd_lds<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    ntt_fwd_lds<true><<<grid, lds_threads(logb), shmem, ctx->stream>>>(d_rows, logn, logb, nslots, d_prime_of_slot, ctx->d_pc, ctx->d_tw_fwd, skip_q);
  } else {
    HIP_TRY(hipFuncSetAttribute((const void*)ntt_fwd_lds<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    ntt_fwd_lds<false><<<grid, lds_threads(logb), shmem, ctx->stream>>>(d_rows, logn, logb, nslots, d_prime_of_slot, ctx->d_pc, ctx->d_tw_fwd, skip_q);
    if (bitrev) {
      void* tmp;
      FHESI_TRY(ws_reserve(ctx, 6, (size_t)nrows << (logn + 3), &tmp));
      i64 total = nrows << logn;
      bitrev_rows<<<(unsigned)((total + 255) / 256), 256, 0, ctx->stream>>>(d_rows, (u64*)tmp, logn, total);
      HIP_TRY(hipMemcpyAsync(d_rows, tmp, (size_t)total * 8, hipMemcpyDeviceToDevice, ctx->stream));
    }
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_ntt_inv(fhesi_ctx* ctx, u64* d_rows, i64 count, int nslots, const int* d_prime_of_slot, bool bitrev) {
  const int logn = ctx->logn;
  const i64 nrows = count * nslots;
  if (nrows == 0) return 0;
  ProfScope prof(ctx, PROF_NTT_INV, (double)nrows);
  u64 skip_q = 0;
  if (bitrev && ntt_tile_supported(logn)) {
    return launch_ntt_inv_tile(ctx, d_rows, nrows, nslots, d_prime_of_slot);
  }
  if (!bitrev && ntt_tile2_supported(logn)) {      // order-free: in-place sub-transforms, then the tail stages
    FHESI_TRY(launch_tile_sub(ctx, false, d_rows, nrows, nslots, d_prime_of_slot));
    return launch_inv_tail(ctx, d_rows, d_rows, count, nslots, d_prime_of_slot);
  }
  if (bitrev && ntt_tile2_supported(logn)) {
    void* tmp;
    FHESI_TRY(ws_reserve(ctx, 6, (size_t)nrows << (logn + 3), &tmp));
    FHESI_TRY(launch_tile_big(ctx, false, d_rows, (u64*)tmp, nrows, nslots, d_prime_of_slot));
    return launch_inv_tail(ctx, (const u64*)tmp, d_rows, count, nslots, d_prime_of_slot);
  }
  const int logb = logn > NTT_LDS_MAX_LOG ? NTT_LDS_MAX_LOG : logn;
  const size_t shmem = sizeof(u64) << logb;
  const unsigned grid = (unsigned)(nrows << (logn - logb));
  if (bitrev && logb == logn) {
    HIP_TRY(hipFuncSetAttribute((const void*)ntt_inv_lds<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    ntt_inv_lds<true><<<grid, lds_threads(logb), shmem, ctx->stream>>>(d_rows, logn, logb, nslots, d_prime_of_slot, ctx->d_pc, ctx->d_tw_inv, skip_q);
  } else {
    if (bitrev) {
      void* tmp;
      FHESI_TRY(ws_reserve(ctx, 6, (size_t)nrows << (logn + 3), &tmp));
      i64 total = nrows << logn;
      bitrev_rows<<<(unsigned)((total + 255) / 256), 256, 0, ctx->stream>>>(d_rows, (u64*)tmp, logn, total);
      HIP_TRY(hipMemcpyAsync(d_rows, tmp, (size_t)total * 8, hipMemcpyDeviceToDevice, ctx->stream));
    }
    HIP_TRY(hipFuncSetAttribute((const void*)ntt_inv_lds<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    ntt_inv_lds<false><<<grid, lds_threads(logb), shmem, ctx->stream>>>(d_rows, logn, logb, nslots, d_prime_of_slot, ctx->d_pc, ctx->d_tw_inv, skip_q);
  }
  for (int st = logn - logb - 1; st >= 0; --st) {
    i64 total = nrows << (logn - 1);
    ntt_inv_global_stage<<<(unsigned)((total + 255) / 256), 256, 0, ctx->stream>>>(d_rows, logn, st, nslots, d_prime_of_slot, ctx->d_pc, ctx->d_tw_inv, total);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// ByteDecomp fused into the forward transform: digit rows [npolys*nd][L][n] straight from the scaled-down parts.
// layout_slots: slots per DoubleCRT in d_out_rows (0 = L); a compact layout (layout_slots = nslot, slot0 = 0) holds only the computed rows.
int launch_ntt_fwd_digits(fhesi_ctx* ctx, const u64* d_parts, int nl, int logQ, int digit_bits, int nd, i64 npolys, u64* d_out_rows, int slot0, int nslot, int layout_slots) {
  if (nslot <= 0) { slot0 = 0; nslot = ctx->L; }
  if (layout_slots <= 0) layout_slots = ctx->L;
  if (slot0 + nslot > layout_slots) FHESI_FAIL("digit transform: slots %d..%d outside a layout of %d", slot0, slot0 + nslot - 1, layout_slots);
  if (!npolys) return 0;
  const bool two_pass = ntt_tile2_supported(ctx->logn);
  if (!(ntt_tile_supported(ctx->logn) || two_pass) || digit_bits >= 32) {      // generic sizes: separate digit kernel, then the row transform
    if (layout_slots != ctx->L) FHESI_FAIL("digit transform: compact layouts need a tile size");
    FHESI_TRY(launch_digits(ctx, d_parts, nl, logQ, digit_bits, nd, npolys, d_out_rows));
    return launch_ntt_fwd(ctx, d_out_rows, npolys * nd, ctx->L, nullptr, true);
  }
  ProfScope prof(ctx, PROF_NTT_FWD, (double)(npolys * nd * nslot));
  const DigitSrc ds{d_parts, nl, digit_bits, nd, slot0};
  if (two_pass) {      // sub-transforms straight from the parts into the rows
    if (layout_slots != ctx->L && !ntt_digits_suborder(ctx, digit_bits)) FHESI_FAIL("digit transform: compact layouts need a single-pass size");
    FHESI_TRY(launch_tile_big_digits(ctx, ds, npolys * nd, d_out_rows, nslot, layout_slots));
    if (ntt_digits_suborder(ctx, digit_bits)) return 0;                     // n = 2^15: head stage fused into the loader, no tail (sub-block order)
    return launch_fwd_tail(ctx, d_out_rows, d_out_rows, npolys * nd, ctx->L, slot0, nslot, nullptr);     // tail in place
  }
  switch (ctx->logn) {
    case 11: FHESI_TRY(launch_tile_digits<11>(ctx, ds, npolys * nd, d_out_rows, nslot, layout_slots)); break;
    case 12: FHESI_TRY(launch_tile_digits<12>(ctx, ds, npolys * nd, d_out_rows, nslot, layout_slots)); break;
    case 13: FHESI_TRY(launch_tile_digits<13>(ctx, ds, npolys * nd, d_out_rows, nslot, layout_slots)); break;
    default: FHESI_TRY(launch_tile_digits<14>(ctx, ds, npolys * nd, d_out_rows, nslot, layout_slots)); break;
  }
  return 0;
}
