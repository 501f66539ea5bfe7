// kernels_ksaux.hip -- the key-switch dot product through two auxiliary primes.
//
// KeySwitchSI::ApplyKeySwitch (FHE-SI.cpp:241-260) computes, for every chain prime q_i and both key rows r,
//     out[r] = sum_k  DoubleCRT(digit_k) * key[r][k]        (DotProduct, Util.h:79-98)
// which costs one forward transform of every digit polynomial PER CHAIN PRIME (3 nd L row transforms per ciphertext, 93 % of
// all transforms of a multiplication).  The digit polynomials have tiny coefficients (below 2^(8 decompSize)), so the negacyclic
// product  V[r][i] = sum_k digit_k (*) keycoef[r][k][i]  with the key taken as its coefficient vector in [0, q_i)  is an INTEGER
// polynomial whose coefficients are bounded by  ncol * n * 2^digit_bits * q_i  <  2^106 -- far below the product of two 60-bit
// primes.  It is therefore computed exactly modulo the two largest chain primes a in {0, 1} only:
//     digits:  2 transforms per digit polynomial instead of L            (launch_ntt_fwd_digits, slots 0..1, compact layout)
//     keys:    K2[a][i][r][k] = NTT_a( iNTT_i(key[r][k][i]) mod q_a )    built once per matrix (ksaux_build), stored tiled by 64-element slices
//     dot:     O[ct][r][i][a] = sum_k D[ct][k][a] * K2[a][i][r][k]  mod q_a                       (dot_aux_kernel)
//     back:    iNTT_a, then V = CRT(O[..][0], O[..][1]) centred modulo q_0 q_1, then V mod q_i    (aux_crt_kernel)
// and V mod q_i is, coefficient by coefficient, exactly what toPoly of the reference's dot product holds modulo q_i -- the rows the
// final intVecCRT (kernels_crt.hip) takes.  Same bits, 2 (nd ncomp) + 2 L 2 row transforms instead of (nd ncomp) L + 2 L.
//
// That is the RESIDUE mode (index i = chain prime).  In LIMB mode (ks_limb_plan: the metric and stress chain shapes) the table is
// built from the key polynomial's integer coefficients in [0, P) instead (toPoly over the chain, once per matrix), cut into limbs of
// B bits: index i = limb, 15 x 74 bits instead of 18 residues at the metric ring, and the closing step is ks_recombine_kernel
// (kernels_crt.hip): S = sum_l V_l 2^(B l), reduced modulo P exactly.  At n = 2^14 the limb products are carried by four 30-bit
// primes instead of q_0, q_1 (kernels_aux32.hip: own 32-bit transforms, one multiply per multiply-add); this file then only plans
// and builds.  Switches for A/B runs: FHESI_KS_DIRECT (per-prime dot product), FHESI_KS_RESIDUES, FHESI_KS_AUX60.
#include "fhesi_internal.h"
#include <cmath>

// 128-bit value -> [0,q)
__device__ __forceinline__ u64 aux_fold128(u128 a, const PrimeConst& pc) {
  const u64 q = pc.q, lo = (u64)a, hi = (u64)(a >> 64);
  const u64 h = d_shoup(hi, 1, pc.one_sh, q);
  const u64 t = d_shoup_lazy(h, pc.r64, pc.r64_sh, q);
  const u64 l = d_shoup_lazy(lo, 1, pc.one_sh, q);
  u64 r = t + l;
  if (r >= pc.two_q) r -= pc.two_q;
  if (r >= q) r -= q;
  return r;
}

// coefficient rows of the key [2*ncol][L][n] (values in [0, q_i)) -> K2[a][i][r][k][n] reduced modulo q_a (q_a > 2^59: one step)
__global__ void __launch_bounds__(256) ksaux_scatter_kernel(const u64* __restrict__ coef, u64* __restrict__ k2, int ncol, int L, i64 n, u64 q0, u64 q1) {
  const i64 row = blockIdx.y;                 // (r * ncol + k) * L + i
  const int i = (int)(row % L);
  const i64 rk = row / L;
  const int k = (int)(rk % ncol), r = (int)(rk / ncol);
  const u64* src = coef + row * n;
  u64* d0 = k2 + ((((i64)0 * L + i) * 2 + r) * ncol + k) * n;
  u64* d1 = k2 + ((((i64)1 * L + i) * 2 + r) * ncol + k) * n;
  for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x) {
    const u64 c = src[j];
    d0[j] = c >= q0 ? c - q0 : c;
    d1[j] = c >= q1 ? c - q1 : c;
  }
}

// Limb mode: kint [2*ncol][n][W] = the key polynomial's integer coefficients in [0, P) (toPoly over the whole chain);
// K2[a][l][r][k][n] = (limb l of B bits) mod q_a.  Build-time only.
__global__ void __launch_bounds__(256) ks_limb_scatter_kernel(const u64* __restrict__ kint, u64* __restrict__ k2, int ncol, int NLB, int B, int W, i64 n,
                                                              u64 q0, u64 q1) {
  const i64 row = blockIdx.y;                 // (r * ncol + k) * NLB + l
  const int l = (int)(row % NLB);
  const i64 rk = row / NLB;
  const int k = (int)(rk % ncol), r = (int)(rk / ncol);
  const int s = B * l, wd = s >> 6, bt = s & 63;
  u64* d0 = k2 + ((((i64)0 * NLB + l) * 2 + r) * ncol + k) * n;
  u64* d1 = k2 + ((((i64)1 * NLB + l) * 2 + r) * ncol + k) * n;
  const u64 r64_0 = (u64)(((u128)1 << 64) % q0), r64_1 = (u64)(((u128)1 << 64) % q1);
  for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x) {
    const u64* x = kint + (rk * n + j) * W;
    auto word = [&](int i) -> u64 { return i < W ? x[i] : 0; };
    const u64 w0 = word(wd), w1 = word(wd + 1), w2 = word(wd + 2);
    const u64 lo = bt ? ((w0 >> bt) | (w1 << (64 - bt))) : w0;
    u64 hi = bt ? ((w1 >> bt) | (w2 << (64 - bt))) : w1;
    hi &= ((u64)1 << (B - 64)) - 1;                                  // B in (64, 128)
    d0[j] = (u64)(((u128)(lo % q0) + (u128)hi * r64_0) % q0);
    d1[j] = (u64)(((u128)(lo % q1) + (u128)hi * r64_1) % q1);
  }
}

// Split representation of a residue v < 2^60 for the dot product: low 30 bits in the low dword, high 30 bits in the high dword.
// The four partial products of two such values are below 2^60 each, so they accumulate in plain 64-bit v_mad_u64_u32 chains --
// no carries, no 128-bit additions -- for 8 columns (the two middle products share one accumulator) before they are gathered.
__device__ __forceinline__ u64 pack30(u64 v) { return (v & 0x3fffffffull) | ((v >> 30) << 32); }
// rows of one auxiliary prime, [i][r][k][n] after the transforms  ->  tiled and split: [i][slice][r][k][64], so that the 2 ncol key
// slices a wave streams for one (chain prime, slice) are one contiguous block and consecutive columns are 512 bytes apart
// (immediate offsets in the loads instead of 64-bit address arithmetic)
__global__ void __launch_bounds__(256) ksaux_retile_kernel(const u64* __restrict__ src, u64* __restrict__ dst, int ncol, i64 n) {
  const i64 row = blockIdx.y;                 // (i * 2 + r) * ncol + k
  const int k = (int)(row % ncol);
  const i64 ir = row / ncol;
  const int r = (int)(ir & 1);
  const i64 i = ir >> 1;
  const i64 nsl = n >> 6;
  for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x)
    dst[((((i * nsl + (j >> 6)) * 2 + r) * ncol + k) << 6) + (j & 63)] = pack30(src[row * n + j]);
}

// O[ct][r][i][a][slice] = sum_k D[ct][k][a][slice] * K2[a][i][r][k][slice]  mod q_a.
// One workgroup = one 64-element slice of CT ciphertexts for one auxiliary prime: their ncol x CT digit slices sit in LDS (reduced
// modulo q_a -- the fused digit transform stores lazy representatives -- and split) and every wave walks its share of the chain
// primes, streaming that prime's 2 ncol key slices (stored split; L2-resident across the ciphertext tiles that follow on the same
// XCD, see the block order in the launcher), 16 key loads in flight per wave.
template <int CT, int NW, int R>
__global__ void __launch_bounds__(NW * 64) dot_aux_kernel(const u64* __restrict__ k2, const u64* __restrict__ dig, int ncol, i64 n, int L, i64 count,
                                                          u64* __restrict__ out, const PrimeConst* __restrict__ pcs, int ntiles, int nsl8) {
  // digit tile [ncol][CT/2][64 lanes][2]: a lane's 16-byte reads are 16 bytes apart (conflict-free ds_read_b128)
  extern __shared__ __attribute__((aligned(16))) u64 dl[];
  constexpr int PW = CT >= 2 ? 2 : 1, CH = CT / PW;
#define DL_IDX(k, c) ((((k) * CH + ((c) / PW)) * 64 + lane) * PW + ((c) % PW))
  const u32 lane = threadIdx.x & 63;       // unsigned: the lane offset of a load is a zero-extended 32-bit index next to a scalar base
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave index as a scalar: the key row pointers stay in SGPRs
  // block order: 8 consecutive slices (one per XCD), then the ciphertext tiles, then the slice groups, then the auxiliary prime
  u32 b = blockIdx.x;
  const u32 s_lo = b & 7; b >>= 3;
  const u32 tile = b % (u32)ntiles; b /= (u32)ntiles;
  const u32 s_hi = b % (u32)nsl8;
  const int a = (int)(b / (u32)nsl8);
  const i64 soff = (i64)(s_hi * 8 + s_lo) * 64;                   // uniform: first element of the slice
  const i64 ct0 = (i64)tile * CT;
  const PrimeConst pc = pcs[a];
  // tile load, 8 row slices per wave in flight (one at a time would pay the HBM latency ncol CT / NW times per workgroup)
  for (int e0 = w; e0 < ncol * CT; e0 += 8 * NW) {
    u64 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * NW, k = e / CT, c = e % CT;
      const i64 ct = ct0 + c;
      v[u] = (e < ncol * CT && ct < count) ? __builtin_nontemporal_load(&(dig + ((ct * ncol + k) * 2 + a) * n + soff)[lane]) : 0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * NW, k = e / CT, c = e % CT;
      if (e < ncol * CT) dl[DL_IDX(k, c)] = pack30(d_shoup(v[u], 1, pc.one_sh, pc.q));
    }
  }
  __syncthreads();
  // R = 2: a wave takes both key rows of a chain prime for its CT ciphertexts; R = 1: one key row, so that a key load feeds CT
  // multiply-adds with the same number of accumulators (half the L2 traffic per multiply-add when CT is doubled)
  for (int pr = w; pr < L * (2 / R); pr += NW) {
    const int i = R == 2 ? pr : pr >> 1, r0 = R == 2 ? 0 : pr & 1;
    // this wave's key streams: contiguous [r][k][64] block of the tiled table
    const u64* k0 = k2 + ((((((i64)a * L + i) * (n >> 6) + (soff >> 6)) * 2 + r0) * ncol) << 6) + lane;
    const u64* k1 = k0 + ((i64)ncol << 6);
    // second level: the three partial-product sums of every 8-column group are added into 96-bit totals (64-bit low word + a
    // carry counter) and combined once per chain prime -- no 128-bit shifts and additions inside the column loop
    u64 tl[CT][R][3];
    u32 th[CT][R][3];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int p = 0; p < 3; ++p) { tl[c][r][p] = 0; th[c][r][p] = 0; }
    // columns in groups of 8, software-pipelined: the key loads of group g + 1 are issued before group g is multiplied
    auto gather = [&](const u64 (&ll)[CT][R], const u64 (&mid)[CT][R], const u64 (&hh)[CT][R]) {
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const u64 part[3] = {ll[c][r], mid[c][r], hh[c][r]};
#pragma unroll
          for (int p = 0; p < 3; ++p) { const u64 t = tl[c][r][p] + part[p]; th[c][r][p] += t < part[p] ? 1u : 0u; tl[c][r][p] = t; }
        }
    };
    auto column = [&](u64 (&ll)[CT][R], u64 (&mid)[CT][R], u64 (&hh)[CT][R], u64 x0, u64 x1, int k) {
      const u32 xa[2][2] = {{(u32)x0, (u32)(x0 >> 32)}, {(u32)x1, (u32)(x1 >> 32)}};
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const u64 d = dl[DL_IDX(k, c)];
        const u32 d0 = (u32)d, d1 = (u32)(d >> 32);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          ll[c][r] += (u64)xa[r][0] * d0;
          mid[c][r] += (u64)xa[r][0] * d1;
          mid[c][r] += (u64)xa[r][1] * d0;
          hh[c][r] += (u64)xa[r][1] * d1;
        }
      }
    };
    const int nfull = ncol & ~7;
    constexpr bool PF = R == 1;          // (with both key rows per wave the second register set spills)
    u64 xc[R][8], xn[R][8];
    if (PF && nfull) {
#pragma unroll
      for (int u = 0; u < 8; ++u) { xc[0][u] = k0[u << 6]; if (R == 2) xc[R - 1][u] = k1[u << 6]; }
    }
    for (int kb = 0; kb < nfull; kb += 8) {
      if (PF) {
        if (kb + 8 < nfull) {
          const u64* p0 = k0 + ((kb + 8) << 6);
#pragma unroll
          for (int u = 0; u < 8; ++u) xn[0][u] = p0[u << 6];
        }
      } else {
        const u64* p0 = k0 + (kb << 6);
        const u64* p1 = k1 + (kb << 6);
#pragma unroll
        for (int u = 0; u < 8; ++u) { xc[0][u] = p0[u << 6]; if (R == 2) xc[R - 1][u] = p1[u << 6]; }
      }
      u64 ll[CT][R], mid[CT][R], hh[CT][R];
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < R; ++r) { ll[c][r] = 0; mid[c][r] = 0; hh[c][r] = 0; }
#pragma unroll
      for (int u = 0; u < 8; ++u) column(ll, mid, hh, xc[0][u], R == 2 ? xc[R - 1][u] : 0, kb + u);
      gather(ll, mid, hh);
      if (PF) {
#pragma unroll
        for (int u = 0; u < 8; ++u) xc[0][u] = xn[0][u];
      }
    }
    if (nfull < ncol) {       // the last ncol mod 8 columns
      u64 ll[CT][R], mid[CT][R], hh[CT][R];
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < R; ++r) { ll[c][r] = 0; mid[c][r] = 0; hh[c][r] = 0; }
      for (int k = nfull; k < ncol; ++k) column(ll, mid, hh, k0[k << 6], R == 2 ? k1[k << 6] : 0, k);
      gather(ll, mid, hh);
    }
    u128 tot[CT][R];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int r = 0; r < R; ++r)
        tot[c][r] = (((u128)th[c][r][0] << 64) | tl[c][r][0]) + ((((u128)th[c][r][1] << 64) | tl[c][r][1]) << 30) + ((((u128)th[c][r][2] << 64) | tl[c][r][2]) << 60);
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      if (ct0 + c < count) {
#pragma unroll
        for (int r = 0; r < R; ++r) (out + ((((ct0 + c) * 2 + r0 + r) * L + i) * 2 + a) * n + soff)[lane] = aux_fold128(tot[c][r], pc);
      }
    }
  }
}

// o: [rows][2][n] coefficient residues modulo (q_0, q_1) of an integer V with |V| < q_0 q_1 / 2;  dst[row][j] = V mod q_i, i = row % L.
__global__ void __launch_bounds__(256) aux_crt_kernel(const u64* __restrict__ o, u64* __restrict__ dst, i64 n, int L, const PrimeConst* __restrict__ pcs,
                                                      u64 q0, u64 q1, u64 q0inv, u64 q0inv_sh, const u64* __restrict__ a_mod /* [L]: q_0 q_1 mod q_i */,
                                                      u64 half_hi, u64 half_lo) {
  const i64 row = blockIdx.y;
  const int i = (int)(row % L);
  const PrimeConst pc = pcs[i];
  const u64 am = a_mod[i];
  const u64* v0p = o + (row * 2 + 0) * n;
  const u64* v1p = o + (row * 2 + 1) * n;
  const u128 half = ((u128)half_hi << 64) | half_lo;            // (q_0 q_1 - 1) / 2
  for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x) {
    const u64 v0 = v0p[j], v1 = v1p[j];
    const u64 v0r = v0 >= q1 ? v0 - q1 : v0;                    // q_1 < q_0 < 2 q_1
    const u64 t = d_shoup(d_submod(v1, v0r, q1), q0inv, q0inv_sh, q1);
    const u128 V = (u128)q0 * t + v0;                           // in [0, q_0 q_1)
    u64 r = aux_fold128(V, pc);
    if (V > half) r = d_submod(r, am, pc.q);                    // centred value V - q_0 q_1
    dst[row * n + j] = r;
  }
}

// Which form of the exact-integer key switch runs for this chain, ring and option set (KS_MODE_*); KS_MODE_DIRECT = none of them: the
// per-chain-prime dot product of key_switch_tail.  The limb plan is evaluated HERE, so a chain the plan rejects (fewer than three
// limbs in the chain product, limbs of at most 64 bits, ...) falls through to the direct path instead of failing in ksaux_build.
int ksaux_mode(fhesi_ctx* ctx, const CrtTables* t, int ncol, int digit_bits, int logQ) {
  if (ctx->L < 2 || digit_bits >= 32 || ctx->opt.ks_direct) return KS_MODE_DIRECT;
  // (1) limb mode over the four 30-bit auxiliary primes (kernels_aux32.hip): rows of 2^14 / 2^15 and the linear-convolution rings, ANY
  //     chain -- the auxiliary modulus does not involve the chain primes, so a chain of 50-bit primes (NTL_SP_NBITS = 50) takes it too
  //     (up to 160 digit columns: what the dot products' LDS tiles hold whatever the limb count of the matrix -- launch_dot32; logQ <= 1264 with
  //     byte digits.  Beyond that the chain forms below run: slower, not refused.)
  if (aux32_applies(ctx) && !ctx->opt.ks_aux60 && !ctx->opt.ks_residues && ncol <= 160) {
    const u32* p32 = aux32_primes(ctx);
    KsLimbPlan plan;
    if (p32 && ks_limb_plan(ctx, t, ncol, digit_bits, logQ, &plan, p32) && plan.a32) return KS_MODE_LIMB32;
  }
  if (ctx->lin_q) return KS_MODE_DIRECT;                       // no other exact form exists on those rings: per-prime Bluestein rows
  // (2) the two largest chain primes as auxiliary modulus: needs the reference's chain descending from 2^60 (FHEContext.cpp:92-108)
  const u64 q0 = ctx->q[0], q1 = ctx->q[1];
  if (q0 <= q1 || q1 < (1ull << 59)) return KS_MODE_DIRECT;
  for (int i = 2; i < ctx->L; ++i) if (ctx->q[i] >= q1) return KS_MODE_DIRECT;
  if (!ctx->pow2) return KS_MODE_DIRECT;
  if (!((ctx->logn >= 11 && ctx->logn <= 14) || ntt_digits_suborder(ctx, digit_bits))) return KS_MODE_DIRECT;      // single-pass digit transforms
  // |V| <= ncol * n * 2^digit_bits * q_0  must stay below q_0 q_1 / 2
  const double lg = std::log2((double)ncol) + (double)ctx->logn + digit_bits + 60.0;
  if (!(lg + 2.0 < std::log2((double)q0) + std::log2((double)q1))) return KS_MODE_DIRECT;
  KsLimbPlan plan;
  return ks_limb_plan(ctx, t, ncol, digit_bits, logQ, &plan, nullptr) ? KS_MODE_LIMB60 : KS_MODE_RESIDUE60;
}

// Limb mode for any chain: the key polynomial's integer coefficient in [0, P) is cut into NLB limbs of B bits, with B the largest
// width for which every limb product sum stays exact in the auxiliary modulus A (the product of the two largest chain primes, or of
// the four 30-bit primes of kernels_aux32.hip at n = 2^14):  |V_l| <= ncol * n * 2^digit_bits * 2^B < A / 2.  The recombination
// (ks_recombine_kernel) shifts V_l + 2^119 by B l bits as a (lo, hi) pair spread over three limbs, which needs 64 < B <= 119 - ... and
// the whole sum inside W + 1 limbs; |S| < 2^(mbits-1) P with the quotient estimate below 2^63.  Anything else stays in residue mode.
bool ks_limb_plan(const fhesi_ctx* ctx, const CrtTables* t, int ncol, int digit_bits, int logQ, KsLimbPlan* plan, const u32* p32) {
  if (ctx->opt.ks_residues) return false;                       // A/B switch: residue mode
  if (t->nidx != ctx->L || t->W < 3) return false;
  KsLimbPlan p;
  u128 A = (u128)ctx->q[0] * ctx->q[1];
  if (p32 && aux32_applies(ctx) && !ctx->opt.ks_aux60) {       // option ks_aux60: two 60-bit primes even where the four 30-bit primes apply
    p.a32 = true;
    A = (u128)((u64)p32[0] * p32[1]) * ((u64)p32[2] * p32[3]);
  }
  if (ctx->lin_q && !p.a32) return false;
  if (A >> 119 > 1) return false;                               // the +2^119 offset needs the auxiliary modulus below 2^120
  // ncol * n * 2^digit_bits bounds a coefficient of one limb product sum; on the linear-convolution rings the fold modulo X^q' + 1 and
  // Phi_m combines four of them (C(j) - C(j+q') -+ (C(phi) - C(phi+q'))), each kept below A / 8 so that the combination stays below A / 2
  const u128 terms = ((u128)ncol * (u128)ctx->phim << digit_bits) * (ctx->lin_q ? 4 : 1);
  if (terms >> 50) return false;
  int B = 0;
  while (B < 100 && (((A / 2) >> (B + 1)) > terms)) ++B;        // the largest B with terms * 2^B < A / 2
  if (B <= 64) return false;                                    // (the limb-scatter kernels cut limbs of 65..100 bits)
  std::vector<u64> P{1};
  for (int i = 0; i < ctx->L; ++i) P = hm::bn_mul_small(P, ctx->q[i]);
  int pbits = (int)(P.size() - 1) * 64;
  for (u64 top = P.back(); top; top >>= 1) ++pbits;
  p.W = t->W; p.LQ = logQ; p.B = B; p.NLB = (pbits + B - 1) / B;
  if ((p.NLB - 1) * p.B + 120 > 64 * (p.W + 1)) return false;   // x = D + sum (V_l + 2^119) 2^(B l) fits W + 1 limbs
  if (logQ < 1 || logQ > 64 * (p.W - 1)) return false;
  int m = 0;
  while (((u128)1 << m) <= terms) ++m;
  p.mbits = m + 1;                                              // |S| < terms * P < 2^(mbits-1) P
  if (p.mbits > 60) return false;
  *plan = p;
  return true;
}

// host big integers (little-endian u64 limbs, fixed width) for the limb mode's constants
static void bn_shl(std::vector<u64>& a, int sh) {
  const int n = (int)a.size(), w = sh >> 6, b = sh & 63;
  std::vector<u64> r(n, 0);
  for (int i = n - 1; i >= w; --i) r[i] = (a[i - w] << b) | ((b && i - w - 1 >= 0) ? (a[i - w - 1] >> (64 - b)) : 0);
  a.swap(r);
}
static void bn_sub(std::vector<u64>& a, const std::vector<u64>& b) {          // a -= b  (mod 2^(64 n))
  u64 borrow = 0;
  for (size_t i = 0; i < a.size(); ++i) {
    const u64 bi = i < b.size() ? b[i] : 0, d = a[i] - bi, b1 = a[i] < bi, d2 = d - borrow, b2 = d < borrow;
    a[i] = d2; borrow = b1 | b2;
  }
}
static bool bn_ge(const std::vector<u64>& a, const std::vector<u64>& b) {
  for (int i = (int)a.size() - 1; i >= 0; --i) { const u64 bi = i < (int)b.size() ? b[i] : 0; if (a[i] != bi) return a[i] > bi; }
  return true;
}

// builds k->d_aux from k->d_rows (device work only; the caller holds the context's stream)
int ksaux_build(fhesi_ctx* ctx, fhesi_ksk* k, int digit_bits, int logQ, int mode) {
  const i64 n = ctx->phim;
  const int L = ctx->L, ncol = k->ncomp * k->ndigits;
  const bool suborder = ntt_digits_suborder(ctx, digit_bits);
  std::vector<int> all(L);
  for (int i = 0; i < L; ++i) all[i] = i;
  CrtTables* t;
  FHESI_TRY(get_crt_tables(ctx, all, &t));
  KsLimbPlan plan;
  if (mode == KS_MODE_DIRECT) FHESI_FAIL("key switch: no auxiliary-prime table for the direct path");
  const bool limb = mode != KS_MODE_RESIDUE60 && ks_limb_plan(ctx, t, ncol, digit_bits, logQ, &plan, mode == KS_MODE_LIMB32 ? aux32_primes(ctx) : nullptr);
  if ((mode != KS_MODE_RESIDUE60) != limb || (limb && plan.a32 != (mode == KS_MODE_LIMB32))) FHESI_FAIL("key switch: the limb plan does not match the selected form");
  int R = limb ? plan.NLB : L;                                // output rows per (ciphertext, key row, auxiliary prime)
  bool centred = false;
  int key_bits = 0;
  auto alloc_table = [&]() -> int {
    // table size: residue / limb rows for two 8-byte (or four 4-byte) auxiliary residues; 2^14-element rows on the linear-convolution rings
    const size_t need = (size_t)2 * R * 2 * ncol * (plan.a32 && limb ? aux32_row_len(ctx) : n) * 8;
    if (k->d_aux && k->aux_bytes < need) { HIP_TRY(hipStreamSynchronize(ctx->stream)); HIP_TRY(hipFree(k->d_aux)); k->d_aux = nullptr; }
    if (!k->d_aux) { HIP_TRY(hipMalloc(&k->d_aux, need)); k->aux_bytes = need; }
    return 0;
  };
  if (!(limb && plan.a32)) FHESI_TRY(alloc_table());           // (the 30-bit limb form sizes its table after it has looked at the key's coefficients)
  if (!k->d_aux_consts) {
    HIP_TRY(hipMalloc(&k->d_aux_consts, (size_t)(L + 2) * 8));
    std::vector<u64> h(L + 2, 0);
    const u128 A = (u128)ctx->q[0] * ctx->q[1];
    for (int i = 0; i < L; ++i) h[i] = (u64)(A % ctx->q[i]);
    ((int*)&h[L])[0] = 0; ((int*)&h[L])[1] = 1;                  // prime_of_slot of the auxiliary rows
    HIP_TRY(hipMemcpyAsync(k->d_aux_consts, h.data(), h.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  void* tmp;
  FHESI_TRY(ws_reserve(ctx, 0, std::max(k->bytes, (size_t)R * 2 * ncol * aux32_row_len(ctx) * 4), &tmp));      // the key rows, later one auxiliary prime's table rows
  HIP_TRY(hipMemcpyAsync(tmp, k->d_rows, k->bytes, hipMemcpyDeviceToDevice, ctx->stream));
  if (ctx->pow2) FHESI_TRY(launch_ntt_inv(ctx, (u64*)tmp, 2 * ncol, L, nullptr, true));
  else FHESI_TRY(launch_bluestein_inv(ctx, (u64*)tmp, 2 * ncol, L, all.data()));          // Cmod::iFFT of every key row on a general-m ring
  if (limb) {
    const int W = t->W;
    u64* kint = nullptr;                                        // the key's integer coefficients: one-off scratch
    HIP_TRY(hipMalloc(&kint, (size_t)2 * ncol * n * W * 8));
    int rc = 0;
    if (plan.a32 && !ctx->opt.ks_long_keys) {
      // Centred limbs: KeySwitchSI::Init draws its polynomial modulo 2^logQ and reduces b modulo 2^logQ (FHE-SI.cpp:176-204), so the integer
      // coefficients of a generated matrix lie in [-2^(logQ-1), 2^(logQ-1)] where the chain product has more than twice as many bits.  The
      // size is MEASURED here, on the matrix at hand (any matrix is legal input): if ceil(nb / B) limbs of the centred integer are fewer
      // than the plan's, and the dot product  sum_k digit_k (*) K_k  of such coefficients is below P / 2 by itself, the table is built from
      // the centred integers -- then S needs no reduction modulo P at all, only modulo 2^logQ (ks_recombine_centred_kernel).
      rc = launch_crt(ctx, t, (const u64*)tmp, L, nullptr, 2 * ncol, 0, 0, 0, kint, W);
      if (!rc) rc = ks32_key_bits(ctx, kint, (i64)2 * ncol * n, W, &key_bits);
      if (!rc) {
        std::vector<u64> P{1};
        for (int i = 0; i < L; ++i) P = hm::bn_mul_small(P, ctx->q[i]);
        int pbits = (int)(P.size() - 1) * 64;
        for (u64 top = P.back(); top; top >>= 1) ++pbits;
        const int nlc = std::max(1, (key_bits + plan.B - 1) / plan.B);
        // |S| <= terms 2^nb  (terms = ncol n 2^digit_bits, x 4 on the folded rings: mbits - 1 bits)  must stay below P / 2
        if (nlc < plan.NLB && plan.mbits + key_bits + 1 < pbits && logQ <= 1024) { centred = true; R = nlc; }
      }
    }
    if (!rc && !centred) rc = launch_crt(ctx, t, (const u64*)tmp, L, nullptr, 2 * ncol, 0, 1, 0, kint, W);
    if (!rc && plan.a32) rc = alloc_table();
    if (!rc && plan.a32) rc = ks32_build(ctx, k, kint, W, plan.B, R, tmp, centred);
    else if (!rc) {
      dim3 grid((unsigned)((n + 255) / 256 > 64 ? 64 : (n + 255) / 256), (unsigned)(2 * ncol * R));
      ks_limb_scatter_kernel<<<grid, 256, 0, ctx->stream>>>(kint, k->d_aux, ncol, R, plan.B, W, n, ctx->q[0], ctx->q[1]);
      if (hipGetLastError() != hipSuccess) rc = 1;
    }
    hipStreamSynchronize(ctx->stream);
    hipFree(kint);
    if (rc) FHESI_FAIL("key switch, limb mode: building the limb table failed");
    // constants of the recombination kernel: D = 2^(mbits-1) P - sum_l 2^(119 + B l)  (W+1 limbs), floor(2^(64(W-2)+128) / P)
    std::vector<u64> P{1};
    for (int i = 0; i < L; ++i) P = hm::bn_mul_small(P, ctx->q[i]);
    std::vector<u64> D(W + 1, 0);
    for (size_t i = 0; i < P.size(); ++i) D[i] = P[i];
    bn_shl(D, plan.mbits - 1);
    for (int l = 0; l < R; ++l) { std::vector<u64> o(W + 1, 0); const int bit = 119 + plan.B * l; o[bit >> 6] = (u64)1 << (bit & 63); bn_sub(D, o); }
    // restoring division of 2^e by P, e = 64 (W-2) + 128
    const int e = 64 * (W - 2) + 128;
    std::vector<u64> rem(W + 1, 0), Pw(W + 1, 0), quo(3, 0);
    for (size_t i = 0; i < P.size(); ++i) Pw[i] = P[i];
    for (int bit = e; bit >= 0; --bit) {
      bn_shl(rem, 1);
      if (bit == e) rem[0] |= 1;
      quo[2] = (quo[2] << 1) | (quo[1] >> 63); quo[1] = (quo[1] << 1) | (quo[0] >> 63); quo[0] <<= 1;
      if (bn_ge(rem, Pw)) { bn_sub(rem, Pw); quo[0] |= 1; }
    }
    if (quo[2]) FHESI_FAIL("key switch, limb mode: reciprocal of the chain product does not fit 128 bits");
    std::vector<u64> h(D);
    h.push_back(quo[0]); h.push_back(quo[1]);
    if (!k->d_limb_consts) HIP_TRY(hipMalloc(&k->d_limb_consts, (size_t)(L + 8) * 8));
    HIP_TRY(hipMemcpy(k->d_limb_consts, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  } else {
    dim3 grid((unsigned)((n + 255) / 256 > 64 ? 64 : (n + 255) / 256), (unsigned)(2 * ncol * L));
    ksaux_scatter_kernel<<<grid, 256, 0, ctx->stream>>>((const u64*)tmp, k->d_aux, ncol, L, n, ctx->q[0], ctx->q[1]);
    HIP_TRY(hipGetLastError());
  }
  k->aux32 = limb && plan.a32;
  k->aux_centred = centred;
  k->aux_long_opt = ctx->opt.ks_long_keys;
  k->aux_key_bits = key_bits;
  k->aux_fold = k->aux32 ? (ctx->lin_prime ? -ctx->lin_q : ctx->lin_q) : 0;      // (negative: the fold of a prime m)
  const int* d_slot = (const int*)(k->d_aux_consts + L);
  const i64 rows_per_a = (i64)R * 2 * ncol;
  if (!k->aux32)
  for (int a = 0; a < 2; ++a) FHESI_TRY(launch_ntt_fwd(ctx, k->d_aux + (i64)a * rows_per_a * n, rows_per_a, 1, d_slot + a, !suborder));
  for (int a = 0; a < 2 && !k->aux32; ++a) {   // tmp (the matrix's own size) takes one auxiliary prime's rows at a time
    u64* half = k->d_aux + (i64)a * rows_per_a * n;
    HIP_TRY(hipMemcpyAsync(tmp, half, (size_t)rows_per_a * n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    dim3 g2((unsigned)((n + 255) / 256 > 64 ? 64 : (n + 255) / 256), (unsigned)rows_per_a);
    ksaux_retile_kernel<<<g2, 256, 0, ctx->stream>>>((const u64*)tmp, half, ncol, n);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipStreamSynchronize(ctx->stream));       // one-off: another lane's stream may use the table right away
  k->aux_suborder = suborder;
  k->aux_rows = R; k->aux_limb_bits = limb ? plan.B : 0; k->aux_logQ = logQ;
  k->aux_mode = mode;
  k->aux_valid = true;
  return 0;
}

#undef DL_IDX
template <int CT, int NW, int R>
static int launch_dot_aux_t(fhesi_ctx* ctx, const fhesi_ksk* k, const u64* d_dig, int ncol, i64 count, u64* d_out) {
  const i64 n = ctx->phim;
  const int ntiles = (int)((count + CT - 1) / CT), nsl8 = (int)(n / 64 / 8);
  const size_t shmem = (size_t)ncol * CT * 64 * 8;
  static std::atomic<unsigned long long> attr_done{0};
  if (!(attr_done.load() >> ctx->device & 1)) {
    HIP_TRY(hipFuncSetAttribute((const void*)dot_aux_kernel<CT, NW, R>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done.fetch_or(1ull << ctx->device);
  }
  const i64 blocks = (i64)8 * ntiles * nsl8 * 2;
  if (blocks > 0x7fffffff) FHESI_FAIL("dot_aux: too many ciphertexts per call");
  PROF_KERNEL(ctx, PROF_DOT, dot_aux_kernel<CT, NW, R>);
  dot_aux_kernel<CT, NW, R><<<(unsigned)blocks, NW * 64, shmem, ctx->stream>>>(k->d_aux, d_dig, ncol, n, k->aux_rows, count, d_out, ctx->d_pc, ntiles, nsl8);
  HIP_TRY(hipGetLastError());
  return 0;
}

// d_dig: [count*ncol][2][n] (compact, auxiliary primes 0 and 1); d_out: [count][2][L][2][n]
int launch_dot_aux(fhesi_ctx* ctx, const fhesi_ksk* k, const u64* d_dig, int ncol, i64 count, u64* d_out) {
  if (!count) return 0;
  if (ctx->phim % 512) FHESI_FAIL("dot_aux: phi(m) must be a multiple of 512");
  // every product of two residues below 2^60 is below 2^120: up to 255 columns fit the 128-bit total
  if (ncol > 255) FHESI_FAIL("dot_aux: %d columns overflow the 128-bit sum", ncol);
  ProfScope prof(ctx, PROF_DOT, (double)count);
  // Measured at the metric ring (66 columns): 4 ciphertexts per tile with one key row per wave 1.76 ms, 2 ciphertexts with both key
  // rows per wave 2.05 ms, 6 waves instead of 16 per workgroup 2.25 ms.  The LDS tile decides what fits (stress ring, 129 columns: 2).
  if ((size_t)ncol * 4 * 512 <= 150 * 1024) return launch_dot_aux_t<4, 16, 1>(ctx, k, d_dig, ncol, count, d_out);
  if ((size_t)ncol * 2 * 512 <= 150 * 1024) return launch_dot_aux_t<2, 16, 2>(ctx, k, d_dig, ncol, count, d_out);
  if ((size_t)ncol * 512 > 160 * 1024) FHESI_FAIL("dot_aux: %d columns do not fit the LDS tile", ncol);
  return launch_dot_aux_t<1, 16, 2>(ctx, k, d_dig, ncol, count, d_out);
}

// d_o: [nrows][2][n] after the inverse transforms; d_dst: [nrows][n], row r belongs to chain prime r % L
int launch_aux_crt(fhesi_ctx* ctx, const fhesi_ksk* k, const u64* d_o, u64* d_dst, i64 nrows) {
  if (!nrows) return 0;
  const i64 n = ctx->phim;
  const u64 q0 = ctx->q[0], q1 = ctx->q[1];
  const u64 inv = hm::invmod(q0 % q1, q1);
  const u128 half = ((u128)q0 * q1 - 1) / 2;
  ProfScope prof(ctx, PROF_EW, (double)nrows);
  dim3 grid((unsigned)((n + 255) / 256 > 64 ? 64 : (n + 255) / 256), (unsigned)nrows);
  aux_crt_kernel<<<grid, 256, 0, ctx->stream>>>(d_o, d_dst, n, ctx->L, ctx->d_pc, q0, q1, inv, hm::shoup(inv, q1), k->d_aux_consts, (u64)(half >> 64), (u64)half);
  HIP_TRY(hipGetLastError());
  return 0;
}
