// kernels_crt.hip -- big-integer <-> residue conversions of the DoubleCRT path.
//   rns_reduce   conv(in,x) of Cmod::FFT (CModulus.cpp:96): signed multi-limb coefficient -> residue mod each q_i,
//                optionally times a word scalar (the `poly * p` lift of Ciphertext.cpp:171)
//   crt          DoubleCRT::toPoly (DoubleCRT.cpp:349-398) = incremental intVecCRT (NumbTh.cpp:307-335), fused with
//                  mode 1: Ciphertext::ScaleDown rounding (Ciphertext.cpp:205-213) + positive Reduce (Util.cpp:3-26, Ciphertext.cpp:94)
//                  mode 2: ReduceCoefficients centered mod 2^logQ (Util.cpp:28-33; FHE-SI.cpp:256)
//                  mode 3: positive residue mod 2^logQ (Reduce(...,true), Ciphertext.cpp:94), limb-major like mode 1
//   digits       Ciphertext::ByteDecompPart (Ciphertext.cpp:82-105) + conv of the digit polys (FHE-SI.cpp:246-249)
//
// The reference's intVecCRT keeps a centred accumulator at every step; the final value is the unique symmetric
// residue of x modulo the full product P, so this kernel runs the mixed-radix recurrence with non-negative digits
// (x stays in [0, P_k) ) and centres once at the end -- same value, no signed big-int arithmetic per step.
#include "fhesi_internal.h"

// 128-bit value -> [0,q): hi * (2^64 mod q) + lo, each reduced by a Shoup step
__device__ __forceinline__ u64 crt_fold128(u128 a, const PrimeConst& pc) {
  const u64 q = pc.q, lo = (u64)a, hi = (u64)(a >> 64);
  const u64 h = d_shoup(hi, 1, pc.one_sh, q);                 // hi mod q
  const u64 t = d_shoup_lazy(h, pc.r64, pc.r64_sh, q);        // hi * 2^64 mod q, in [0,2q)
  const u64 l = d_shoup_lazy(lo, 1, pc.one_sh, q);            // lo mod q, in [0,2q)
  u64 r = t + l;
  if (r >= pc.two_q) r -= pc.two_q;
  if (r >= q) r -= q;
  return r;
}

// ----------------------------------------------------------------------------------------- rns_reduce
// limbs: [npolys_total][n][nlimbs] two's complement.  rows: [npolys_total][nslots][n].  pow64: [L][nlimbs+1].
// One block stages 256 coefficients (all limbs, coalesced) in LDS, limb-major, and then produces their residues for every
// prime slot: the limbs leave HBM/L2 once instead of once per prime.
__global__ void __launch_bounds__(256) rns_reduce_kernel(const u64* __restrict__ limbs, int nlimbs, i64 ncoeffs, i64 n, int npoly_mod,
                                                          const u64* __restrict__ scalar_res /* [npoly_mod][nslots] residues or null */,
                                                          u64* __restrict__ rows, int nslots, const int* __restrict__ prime_of_slot,
                                                          const PrimeConst* __restrict__ pcs, const Shoup2* __restrict__ pow64) {
  extern __shared__ __attribute__((aligned(16))) u64 sl[];       // [nlimbs][256]
  const i64 poly = blockIdx.y;
  const i64 j0 = (i64)blockIdx.x * 256;
  const int tid = threadIdx.x;
  const u64* src = limbs + poly * ncoeffs * nlimbs;
  const i64 navail = ncoeffs - j0 < 256 ? (ncoeffs - j0 < 0 ? 0 : ncoeffs - j0) : 256;     // coefficients of this tile that exist
  const i64 nwords = navail * nlimbs;
  for (i64 e = tid; e < 256 * (i64)nlimbs; e += 256) {
    const int cj = (int)(e / nlimbs), k = (int)(e % nlimbs);
    sl[k * 256 + cj] = e < nwords ? src[j0 * nlimbs + e] : 0;
  }
  __syncthreads();
  const i64 j = j0 + tid;
  if (j >= n) return;
  const bool neg = (sl[(nlimbs - 1) * 256 + tid] >> 63) != 0;
  for (int slot = 0; slot < nslots; ++slot) {
    const int prime = prime_of_slot ? prime_of_slot[slot] : slot;
    const PrimeConst pc = pcs[prime];
    const u64 q = pc.q, two_q = pc.two_q;
    const Shoup2* pw = pow64 + (i64)prime * (nlimbs + 1);
    u64 acc = 0;
    for (int k = 0; k < nlimbs; ++k) {
      acc += d_shoup_lazy(sl[k * 256 + tid], pw[k].w, pw[k].wp, q);     // each term in [0,2q)
      if (acc >= two_q) acc -= two_q;                                      // keep acc in [0,2q)
    }
    if (acc >= q) acc -= q;
    if (neg) acc = d_submod(acc, pw[nlimbs].w, q);                         // two's complement: value = unsigned - 2^(64 nlimbs)
    if (scalar_res) { const u64 sc = scalar_res[(poly % npoly_mod) * nslots + slot]; if (sc) acc = d_mulmod(acc, sc, pc); }
    rows[(poly * nslots + slot) * n + j] = acc;
  }
}

// The same for a compile-time limb count (the usual coefficient widths): the limbs of a coefficient sit in registers across the
// prime loop, and the residue is ONE exact 128-bit sum of x_k * (2^(64k) mod q) (4 word multiplies per limb instead of a Shoup
// step's 10), folded every 8 limbs when the sum could overflow and reduced once.
template <int NL>
__global__ void __launch_bounds__(256) rns_reduce_kernel_t(const u64* __restrict__ limbs, i64 ncoeffs, i64 n, int npoly_mod, i64 pow_poly_stride,
                                                            u64* __restrict__ rows, int nslots, const int* __restrict__ prime_of_slot,
                                                            const PrimeConst* __restrict__ pcs, const u64* __restrict__ pows, int L) {
  __shared__ __attribute__((aligned(16))) u64 sl[NL * 256];       // [NL][256]
  const i64 poly = blockIdx.y;
  const i64 j0 = (i64)blockIdx.x * 256;
  const int tid = threadIdx.x;
  const u64* src = limbs + poly * ncoeffs * NL;
  const i64 navail = ncoeffs - j0 < 256 ? (ncoeffs - j0 < 0 ? 0 : ncoeffs - j0) : 256;
  const int nwords = (int)navail * NL;
#pragma unroll
  for (int it = 0; it < NL; ++it) {
    const int e = it * 256 + tid;
    sl[(e % NL) * 256 + e / NL] = e < nwords ? src[j0 * NL + e] : 0;
  }
  __syncthreads();
  const i64 j = j0 + tid;
  if (j >= n) return;
  u64 x[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) x[k] = sl[k * 256 + tid];
  const bool neg = (x[NL - 1] >> 63) != 0;
  for (int slot = 0; slot < nslots; ++slot) {
    const int prime = prime_of_slot ? prime_of_slot[slot] : slot;
    const PrimeConst pc = pcs[prime];
    // pows[class][prime][k] = s * 2^(64k) mod q: the word scalar s of the polynomial's class (the `poly * p` lift) is folded into the table
    const u64* pw = pows + (poly % npoly_mod) * pow_poly_stride + (i64)prime * (NL + 3);
    u128 acc = 0;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      acc += (u128)x[k] * pw[k];
      if (NL > 15 && (k & 7) == 7 && k + 1 < NL) acc = crt_fold128(acc, pc);      // 8 more terms below 2^124 on top of a value below q
    }
    // acc < 2^127 -> [0, q): quotient by mu = floor(2^128 / q) = pw[NL+2]:pw[NL+1], low 64 bits only (all the remainder needs),
    // taken from the three high partial products (at most 3 below the true one), then up to three subtractions of q < 2^60
    const u64 a0 = (u64)acc, a1 = (u64)(acc >> 64), m0 = pw[NL + 1], m1 = pw[NL + 2];
    const u64 qh = a1 * m1 + __umul64hi(a1, m0) + __umul64hi(a0, m1);
    u64 r = a0 - qh * pc.q;
    if (r >= pc.two_q) r -= pc.two_q;
    if (r >= pc.two_q) r -= pc.two_q;
    if (r >= pc.q) r -= pc.q;
    if (neg) r = d_submod(r, pw[NL], pc.q);
    rows[(poly * nslots + slot) * n + j] = r;
  }
}
template <int NL>
static void launch_rns_t(fhesi_ctx* ctx, dim3 grid, const u64* d_limbs, i64 ncoeffs, i64 n, int npoly, i64 pow_poly_stride, u64* d_rows, int nslots,
                         const int* d_prime_of_slot, const u64* d_pows) {
  PROF_KERNEL(ctx, PROF_RNS, rns_reduce_kernel_t<NL>);
  rns_reduce_kernel_t<NL><<<grid, 256, 0, ctx->stream>>>(d_limbs, ncoeffs, n, npoly, pow_poly_stride, d_rows, nslots, d_prime_of_slot, ctx->d_pc, d_pows, ctx->L);
}

int launch_rns_reduce(fhesi_ctx* ctx, const u64* d_limbs, int nlimbs, i64 ncoeffs, i64 count, int npoly, const u64* scalar_mul,
                      u64* d_rows, int nslots, const int* d_prime_of_slot) {
  if (!count || !npoly) return 0;
  // 2^(64k) mod q table for this limb count
  Shoup2* d_pow = nullptr;
  auto it = ctx->pow64_cache.find(nlimbs);
  if (it == ctx->pow64_cache.end()) {
    std::vector<Shoup2> h((size_t)ctx->L * (nlimbs + 1));
    for (int l = 0; l < ctx->L; ++l) {
      const u64 q = ctx->q[l];
      const u64 b = (u64)(((u128)1 << 64) % q);
      u64 cur = 1 % q;
      for (int k = 0; k <= nlimbs; ++k) { h[(size_t)l * (nlimbs + 1) + k] = {cur, hm::shoup(cur, q)}; cur = hm::mulmod(cur, b, q); }
    }
    HIP_TRY(hipMalloc(&d_pow, h.size() * sizeof(Shoup2)));
    HIP_TRY(hipMemcpyAsync(d_pow, h.data(), h.size() * sizeof(Shoup2), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->pow64_cache[nlimbs] = d_pow;
  } else d_pow = it->second;
  // per-(poly, slot) scalar residues, cached on the device (no host synchronisation in the steady state)
  u64* d_sc = nullptr;
  if (scalar_mul) {
    if (d_prime_of_slot) FHESI_FAIL("rns_reduce: scalar lift only supported on the full prime set");
    std::vector<u64> key(scalar_mul, scalar_mul + npoly);
    key.push_back((u64)nslots);
    auto sit = ctx->scalar_cache.find(key);
    if (sit == ctx->scalar_cache.end()) {
      std::vector<u64> h((size_t)npoly * nslots, 0);
      for (int p = 0; p < npoly; ++p)
        for (int s = 0; s < nslots; ++s) h[(size_t)p * nslots + s] = scalar_mul[p] ? scalar_mul[p] % ctx->q[s] : 0;
      HIP_TRY(hipMalloc(&d_sc, h.size() * 8));
      HIP_TRY(hipMemcpy(d_sc, h.data(), h.size() * 8, hipMemcpyHostToDevice));
      ctx->scalar_cache[key] = d_sc;
    } else d_sc = sit->second;
  }
  const i64 n = ctx->phim;
  ProfScope prof(ctx, PROF_RNS, (double)(count * npoly));
  dim3 grid((unsigned)((n + 255) / 256), (unsigned)(count * npoly));
  const i64 nc = ncoeffs < n ? ncoeffs : n;
  // table of the compile-time-width kernel: s * 2^(64k) mod q per (class, prime, k), s = the class's word scalar (1 without a lift)
  u64* d_pows = nullptr;
  const i64 pow_stride = (i64)ctx->L * (nlimbs + 3);
  if (nlimbs <= 20) {
    std::vector<u64> key;
    if (scalar_mul) key.assign(scalar_mul, scalar_mul + npoly); else key.assign(1, 0);
    key.push_back(0x7461626c65320000ull | (u64)nlimbs);          // tag: scaled power table (+ Barrett constant) of this limb count
    auto pit = ctx->scalar_cache.find(key);
    if (pit == ctx->scalar_cache.end()) {
      const int ncls = scalar_mul ? npoly : 1;
      std::vector<u64> h((size_t)ncls * pow_stride);
      for (int p = 0; p < ncls; ++p)
        for (int l = 0; l < ctx->L; ++l) {
          const u64 q = ctx->q[l];
          const u64 b = (u64)(((u128)1 << 64) % q);
          u64 cur = (scalar_mul && scalar_mul[p]) ? scalar_mul[p] % q : 1 % q;
          u64* e = &h[(size_t)p * pow_stride + (size_t)l * (nlimbs + 3)];
          for (int k = 0; k <= nlimbs; ++k) { e[k] = cur; cur = hm::mulmod(cur, b, q); }
          // floor(2^128 / q) = floor((2^128 - 1) / q) for odd q > 1
          const u128 mu = ~(u128)0 / q;
          e[nlimbs + 1] = (u64)mu; e[nlimbs + 2] = (u64)(mu >> 64);
        }
      HIP_TRY(hipMalloc(&d_pows, h.size() * 8));
      HIP_TRY(hipMemcpy(d_pows, h.data(), h.size() * 8, hipMemcpyHostToDevice));
      ctx->scalar_cache[key] = d_pows;
    } else d_pows = pit->second;
  }
  const int npoly_cls = scalar_mul ? npoly : 1;
#define RNS_CASE(NL) case NL: launch_rns_t<NL>(ctx, grid, d_limbs, nc, n, npoly_cls, pow_stride, d_rows, nslots, d_prime_of_slot, d_pows); HIP_TRY(hipGetLastError()); return 0;
  switch (nlimbs) {
    RNS_CASE(1) RNS_CASE(2) RNS_CASE(3) RNS_CASE(4) RNS_CASE(5) RNS_CASE(6) RNS_CASE(7) RNS_CASE(8) RNS_CASE(9) RNS_CASE(10)
    RNS_CASE(11) RNS_CASE(12) RNS_CASE(13) RNS_CASE(14) RNS_CASE(15) RNS_CASE(16) RNS_CASE(17) RNS_CASE(18) RNS_CASE(19) RNS_CASE(20)
    default: break;
  }
#undef RNS_CASE
  const size_t shmem = (size_t)nlimbs * 256 * sizeof(u64);
  if (shmem > 160 * 1024) FHESI_FAIL("rns_reduce: coefficients of %d limbs are too wide", nlimbs);
  HIP_TRY(hipFuncSetAttribute((const void*)rns_reduce_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  rns_reduce_kernel<<<grid, 256, shmem, ctx->stream>>>(d_limbs, nlimbs, ncoeffs < n ? ncoeffs : n, n, npoly, d_sc, d_rows, nslots, d_prime_of_slot, ctx->d_pc, d_pow);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ----------------------------------------------------------------------------------------- CRT tables
int get_crt_tables(fhesi_ctx* ctx, const std::vector<int>& idx, CrtTables** out) {
  auto it = ctx->crt_cache.find(idx);
  if (it != ctx->crt_cache.end()) { *out = it->second; return 0; }
  const int K = (int)idx.size();
  if (K == 0) FHESI_FAIL("CRT over an empty prime set");
  // product limbs
  std::vector<std::vector<u64>> P(K + 1);
  P[0] = {1};
  for (int k = 0; k < K; ++k) P[k + 1] = hm::bn_mul_small(P[k], ctx->q[idx[k]]);
  const int W = (int)P[K].size() + 1;   // one spare limb for the sign / rounding carry
  CrtTables* t = new CrtTables();
  t->nidx = K; t->W = W; t->idx = idx;
  // blob layout (u64 words): idx[K] (as u64) | pow64[K][W] (2 words each) | pinv[K] (2 words) | P[K+1][W] | halfP[W] | M[K][W] | cinv[K][3]
  const size_t n_idx = K, n_pow = (size_t)K * W * 2, n_pinv = (size_t)K * 2, n_P = (size_t)(K + 1) * W, n_half = W, n_M = (size_t)K * W, n_cinv = (size_t)K * 3;
  std::vector<u64> blob(n_idx + n_pow + n_pinv + n_P + n_half + n_M + n_cinv, 0);
  u64* b_idx = blob.data();
  u64* b_pow = b_idx + n_idx;
  u64* b_pinv = b_pow + n_pow;
  u64* b_P = b_pinv + n_pinv;
  u64* b_half = b_P + n_P;
  for (int k = 0; k < K; ++k) {
    const u64 q = ctx->q[idx[k]];
    ((int*)b_idx)[k] = idx[k];
    const u64 base = (u64)(((u128)1 << 64) % q);
    u64 cur = 1 % q;
    for (int j = 0; j < W; ++j) { b_pow[((size_t)k * W + j) * 2] = cur; b_pow[((size_t)k * W + j) * 2 + 1] = hm::shoup(cur, q); cur = hm::mulmod(cur, base, q); }
    // P_k mod q_k
    u64 pk = 0;
    for (int j = (int)P[k].size() - 1; j >= 0; --j) pk = (u64)((((u128)pk << 64) | P[k][j]) % q);
    const u64 inv = hm::invmod(pk, q);
    b_pinv[k * 2] = inv; b_pinv[k * 2 + 1] = hm::shoup(inv, q);
  }
  for (int k = 0; k <= K; ++k)
    for (size_t j = 0; j < P[k].size(); ++j) b_P[(size_t)k * W + j] = P[k][j];
  // halfP = (P-1)/2  (P odd)
  {
    std::vector<u64> h(W, 0);
    for (size_t j = 0; j < P[K].size(); ++j) h[j] = P[K][j];
    h[0] -= 1;   // P odd -> no borrow
    for (int j = 0; j < W; ++j) h[j] = (h[j] >> 1) | (j + 1 < W ? h[j + 1] << 63 : 0);
    for (int j = 0; j < W; ++j) b_half[j] = h[j];
  }
  // sum-form tables
  {
    u64* b_M = b_half + n_half;
    u64* b_c = b_M + n_M;
    for (int i = 0; i < K; ++i) {
      const u64 q = ctx->q[idx[i]];
      std::vector<u64> M{1};
      for (int k = 0; k < K; ++k) if (k != i) M = hm::bn_mul_small(M, ctx->q[idx[k]]);
      u64 mi = 0;
      for (int j = (int)M.size() - 1; j >= 0; --j) mi = (u64)((((u128)mi << 64) | M[j]) % q);
      const u64 c = hm::invmod(mi, q);
      for (size_t j = 0; j < M.size(); ++j) b_M[(size_t)i * W + j] = M[j];
      const u128 num = (u128)c << 64;
      const u64 hi = (u64)(num / q), rem = (u64)(num % q);
      b_c[i * 3] = c; b_c[i * 3 + 1] = hi; b_c[i * 3 + 2] = (u64)(((u128)rem << 64) / q);
    }
  }
  HIP_TRY(hipMalloc(&t->d_blob, blob.size() * 8));
  HIP_TRY(hipMemcpyAsync(t->d_blob, blob.data(), blob.size() * 8, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  t->d_idx = (int*)t->d_blob;
  t->d_pow64 = (Shoup2*)(t->d_blob + n_idx);
  t->d_pinv = (Shoup2*)(t->d_blob + n_idx + n_pow);
  t->d_P = t->d_blob + n_idx + n_pow + n_pinv;
  t->d_halfP = t->d_P + n_P;
  t->d_M = t->d_halfP + n_half;
  t->d_cinv = t->d_M + n_M;
  ctx->crt_cache[idx] = t;
  *out = t;
  return 0;
}

// ----------------------------------------------------------------------------------------- CRT kernel
// Mode-dependent store of a finished W-limb two's-complement value whose limb count and logQ are compile-time constants
// (static register indices).  Mode 1 reads the limbs from bit logQ - 1 upwards only, modes 2 and 3 the limbs below bit logQ only.
template <int MAXW, int WFIX, int LQFIX>
__device__ __forceinline__ void crt_store_fixed(const u64 (&x)[MAXW], int mode, u64* __restrict__ out, i64 poly, i64 n, i64 j, int nl_out) {
  static_assert(WFIX > 0 && WFIX <= MAXW, "static epilogue needs a compile-time limb count");
    constexpr int LQ = LQFIX, w = LQ >> 6, b = LQ & 63, sw = (LQ - 1) >> 6, sb = (LQ - 1) & 63;
    const u64 sf = (x[WFIX - 1] >> 63) ? ~0ull : 0ull;
#define XS(i) ((i) < WFIX ? x[(i) < WFIX ? (i) : 0] : sf)
    const u64 hbit = (XS(sw) >> sb) & 1;           // bit logQ-1: the rounding carry (mode 1) / the sign of the centred residue (mode 2)
    if (mode == 1) {
      u64 carry = hbit;
      u64* o = out + poly * nl_out * n + j;
#pragma unroll
      for (int i = 0; i < MAXW; ++i) {
        if (i < nl_out) {
          const u64 lo = XS(i + w), hi = XS(i + w + 1);
          u64 val = b ? ((lo >> b) | (hi << ((64 - b) & 63))) : lo;
          val += carry;
          carry = (carry && val == 0);
          const int bits_left = LQ - 64 * i;
          if (bits_left < 64) val &= (bits_left <= 0) ? 0ull : ((1ull << (bits_left & 63)) - 1);
          o[(i64)i * n] = val;
        }
      }
      for (int i = MAXW; i < nl_out; ++i) o[(i64)i * n] = 0;
    } else if (mode == 3) {
      u64* o = out + poly * nl_out * n + j;
#pragma unroll
      for (int i = 0; i < MAXW; ++i) {
        if (i < nl_out) {
          u64 val = XS(i);
          const int bits_left = LQ - 64 * i;
          if (bits_left < 64) val &= (bits_left <= 0) ? 0ull : ((1ull << (bits_left & 63)) - 1);
          o[(i64)i * n] = val;
        }
      }
      for (int i = MAXW; i < nl_out; ++i) o[(i64)i * n] = 0;
    } else {
      // coefficient-major output.  Whole waves (n a multiple of 64; every caller runs one thread per coefficient in workgroups of whole waves and
      // returns past the end before this point) whose caller holds exactly ceil(logQ / 64) limbs hand their 64 x NLQ limbs -- one contiguous
      // block -- through LDS rows of their own and store 512 contiguous bytes per instruction; a lane's direct stores touch 32-64 cache lines
      // per instruction for 8-16 bytes each (round 5, as ks_recombine_centred_kernel: -0.5 ms per 1024 there)
      constexpr int NLQ = (LQ + 63) >> 6;
      if (nl_out == NLQ && NLQ <= MAXW && (NLQ & (NLQ - 1)) == 0 && (n & 63) == 0 && (blockDim.x & 63) == 0) {
        __shared__ u64 stg[4][64 * (NLQ + 1)];
        const u32 lane = threadIdx.x & 63, wv = (threadIdx.x >> 6) & 3;
        u64* __restrict__ st = stg[wv];
        if (blockDim.x <= 256) {
#pragma unroll
          for (int i = 0; i < NLQ; ++i) {
            u64 val = XS(i);
            const int bits_left = LQ - 64 * i;
            if (bits_left < 64) { const u64 mask = (1ull << (bits_left & 63)) - 1; val = hbit ? (val | ~mask) : (val & mask); }
            st[lane * (NLQ + 1) + i] = val;
          }
          __builtin_amdgcn_wave_barrier();
          u64* __restrict__ ow = out + (poly * n + (j - lane)) * NLQ;
#pragma unroll
          for (int k = 0; k < NLQ; ++k) { const u32 e = k * 64 + lane; ow[e] = st[(e / NLQ) * (NLQ + 1) + (e % NLQ)]; }
          return;
        }
      }
      u64* o = out + (poly * n + j) * nl_out;
#pragma unroll
      for (int i = 0; i < MAXW; ++i) {
        if (i < nl_out) {
          u64 val = XS(i);
          const int bits_left = LQ - 64 * i;
          if (bits_left <= 0) val = hbit ? ~0ull : 0ull;
          else if (bits_left < 64) { const u64 mask = (1ull << (bits_left & 63)) - 1; val = hbit ? (val | ~mask) : (val & mask); }
          o[i] = val;
        }
      }
      for (int i = MAXW; i < nl_out; ++i) o[i] = hbit ? ~0ull : 0ull;
    }
#undef XS
}

// One thread per coefficient.  MAXW = compile-time bound on the limb count (register array, static indices only);
// the finished W-limb integer is staged in LDS (limb-major, conflict-free) for the mode-dependent epilogue that
// needs run-time limb indices (shift by logQ).
// KFIX / WFIX > 0: the prime count and limb count are compile-time constants, so the triangular recurrence unrolls
// completely (no uniform branches, constants hoisted); 0 = run-time values.
// LQFIX > 0 (with KFIX, WFIX): logQ is a compile-time constant too, so the epilogue indexes the limb registers statically and
// the LDS staging (and its cap on resident workgroups) disappears; modes 1-3 only.
template <int MAXW, int KFIX, int WFIX, int LQFIX = 0>
__global__ void __launch_bounds__(128) crt_kernel(const u64* __restrict__ rows, i64 n, int nslots_layout, const int* __restrict__ slot_of /* [K] or null */,
                                                  int K_rt, int W_rt, const int* __restrict__ idx, const Shoup2* __restrict__ pow64,
                                                  const Shoup2* __restrict__ pinv, const u64* __restrict__ Ptab, const u64* __restrict__ halfP,
                                                  const PrimeConst* __restrict__ pcs, int mode, int positive, int logQ,
                                                  u64* __restrict__ out, int nl_out, const unsigned char* __restrict__ block_flags) {
  extern __shared__ __attribute__((aligned(16))) u64 sx[];   // [W][blockDim.x]
  // clean-up pass behind crt_sum_kernel: only the workgroups it flagged recompute (and overwrite) their coefficients
  if (block_flags && !block_flags[(size_t)blockIdx.y * gridDim.x + blockIdx.x]) return;
  const i64 poly = blockIdx.y;
  const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = j < n;
  const int K = KFIX ? KFIX : K_rt, W = WFIX ? WFIX : W_rt;
  const u64* base = rows + poly * nslots_layout * n;
  u64 x[MAXW];
#pragma unroll
  for (int i = 0; i < MAXW; ++i) x[i] = 0;
  if (active) {
    x[0] = base[(i64)(slot_of ? slot_of[0] : idx[0]) * n + j];
#pragma unroll
    for (int k = 1; k < (KFIX ? KFIX : K); ++k) {
      const PrimeConst pc = pcs[idx[k]];
      const u64 q = pc.q, two_q = pc.two_q;
      const u64 rk = base[(i64)(slot_of ? slot_of[k] : idx[k]) * n + j];
      // t = x mod q_k   (x < P_k < 2^(64k): k limbs)
      // as one exact 128-bit sum of products x_i * (2^(64 i) mod q_k), folded every 8 limbs (q_k < 2^60: 8 products of
      // < 2^124 plus a residue fit 128 bits) -- 4 multiplies per limb instead of the 10 of a Shoup product per limb
      const Shoup2* pw = pow64 + (i64)k * W;      // wave-uniform scalar loads (an LDS copy of the tables measured slower)
      u128 acc = 0;
      constexpr int fold_mask = 7;
#pragma unroll
      for (int i = 0; i < MAXW; ++i) {
        if (i < k && i < W) {
          acc += (u128)x[i] * pw[i].w;
          if ((i & fold_mask) == fold_mask) acc = crt_fold128(acc, pc);
        }
      }
      u64 t = crt_fold128(acc, pc);
      (void)two_q;
      // v = (r_k - t) * P_k^-1 mod q_k   (NumbTh.cpp:314-317 without the centring, see file header)
      const u64 v = d_shoup(d_submod(rk, t, q), pinv[k].w, pinv[k].wp, q);
      // x += v * P_k
      const u64* Pk = Ptab + (i64)k * W;
      u64 carry = 0;
#pragma unroll
      for (int i = 0; i < MAXW; ++i) {
        if (i <= k && i < W) {
          const u128 s = (u128)v * Pk[i] + x[i] + carry;      // < 2^128: (2^64-1)^2 + 2 (2^64-1)
          x[i] = (u64)s;
          carry = (u64)(s >> 64);
        }
      }
    }
    // centre: if x > (P-1)/2 then x -= P   (DoubleCRT.cpp:375-376 / NumbTh.cpp:318; `positive` keeps [0,P), :392-395)
    if (!positive) {
      bool gt = false, decided = false;
#pragma unroll
      for (int i = MAXW - 1; i >= 0; --i) {
        if (i < W) {
          const u64 h = halfP[i];
          if (!decided && x[i] != h) { gt = x[i] > h; decided = true; }
        }
      }
      if (gt) {
        const u64* Pf = Ptab + (i64)K * W;
        u64 borrow = 0;
#pragma unroll
        for (int i = 0; i < MAXW; ++i) {
          if (i < W) {
            const u64 p = Pf[i];
            const u64 d = x[i] - p;
            const u64 b1 = x[i] < p;
            const u64 d2 = d - borrow;
            const u64 b2 = d < borrow;
            x[i] = d2;
            borrow = b1 | b2;
          }
        }
      }
    }
  }
  if constexpr (LQFIX > 0) {
    if (!active) return;
    crt_store_fixed<MAXW, WFIX, LQFIX>(x, mode, out, poly, n, j, nl_out);
    return;
  }
#pragma unroll
  for (int i = 0; i < MAXW; ++i)
    if (i < W) sx[i * blockDim.x + threadIdx.x] = x[i];
  // (each thread only reads back its own column: no barrier needed)
  if (!active) return;
  const u64 sign_fill = (sx[(W - 1) * blockDim.x + threadIdx.x] >> 63) ? ~0ull : 0ull;
  auto X = [&](int i) -> u64 { return i < W ? sx[i * blockDim.x + threadIdx.x] : sign_fill; };
  if (mode == 0) {
    u64* o = out + (poly * n + j) * nl_out;
    for (int i = 0; i < nl_out; ++i) o[i] = X(i);
  } else if (mode == 1) {
    // y = floor((2x + q) / 2q) = (x >> logQ) + bit_{logQ-1}(x), q = 2^logQ (arithmetic shift = floor);
    // keep the positive residue mod 2^logQ, limb-major [nl_out][n]
    const int w = logQ >> 6, b = logQ & 63;
    u64 carry = (X((logQ - 1) >> 6) >> ((logQ - 1) & 63)) & 1;
    u64* o = out + poly * nl_out * n + j;
    for (int i = 0; i < nl_out; ++i) {
      const u64 lo = X(i + w), hi = X(i + w + 1);
      u64 val = b ? ((lo >> b) | (hi << (64 - b))) : lo;
      val += carry;
      carry = (carry && val == 0);
      const int bits_left = logQ - 64 * i;
      if (bits_left < 64) val &= (bits_left <= 0) ? 0ull : ((1ull << bits_left) - 1);
      o[(i64)i * n] = val;
    }
  } else if (mode == 3) {
    // positive residue mod 2^logQ of the value itself, limb-major [nl_out][n]: what ByteDecompPart (Ciphertext.cpp:94) takes of
    // an unscaled part (the key switch after Ciphertext::operator>>=, Regression.h:171-172)
    u64* o = out + poly * nl_out * n + j;
    for (int i = 0; i < nl_out; ++i) {
      u64 val = X(i);
      const int bits_left = logQ - 64 * i;
      if (bits_left < 64) val &= (bits_left <= 0) ? 0ull : ((1ull << bits_left) - 1);
      o[(i64)i * n] = val;
    }
  } else {
    // centred residue mod 2^logQ, two's complement, coefficient-major
    const u64 sbit = (X((logQ - 1) >> 6) >> ((logQ - 1) & 63)) & 1;
    u64* o = out + (poly * n + j) * nl_out;
    for (int i = 0; i < nl_out; ++i) {
      u64 val = X(i);
      const int bits_left = logQ - 64 * i;
      if (bits_left <= 0) val = sbit ? ~0ull : 0ull;
      else if (bits_left < 64) { const u64 mask = (1ull << bits_left) - 1; val = sbit ? (val | ~mask) : (val & mask); }
      o[i] = val;
    }
  }
}

template <int MAXW, int KFIX = 0, int WFIX = 0, int LQFIX = 0>
static int launch_crt_t(fhesi_ctx* ctx, const CrtTables* t, const u64* d_rows, int nslots_layout, const int* d_slot_of, i64 npolys, int mode,
                        int positive, int logQ, u64* d_out, int nl_out, const unsigned char* d_block_flags = nullptr) {
  const int TB = 128;
  const size_t shmem = LQFIX ? 0 : (size_t)t->W * TB * sizeof(u64);
  static std::atomic<unsigned long long> attr_done{0};     // one bit per device: the attribute is per device
  if (!LQFIX && !(attr_done.load() >> ctx->device & 1)) { HIP_TRY(hipFuncSetAttribute((const void*)crt_kernel<MAXW, KFIX, WFIX, LQFIX>, hipFuncAttributeMaxDynamicSharedMemorySize, MAXW * TB * 8)); attr_done.fetch_or(1ull << ctx->device); }
  dim3 grid((unsigned)((ctx->phim + TB - 1) / TB), (unsigned)npolys);
  crt_kernel<MAXW, KFIX, WFIX, LQFIX><<<grid, TB, shmem, ctx->stream>>>(d_rows, ctx->phim, nslots_layout, d_slot_of, t->nidx, t->W, t->d_idx, t->d_pow64, t->d_pinv, t->d_P,
                                                    t->d_halfP, ctx->d_pc, mode, positive, logQ, d_out, nl_out, d_block_flags);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ----------------------------------------------------------------------------------------- CRT, sum form
// The pipeline's conversions (modes 1-3, centred, compile-time prime count / limb count / logQ) do not need the whole integer:
//   x = sum_i y_i M_i - kappa P,   M_i = P/q_i,   y_i = r_i (M_i^-1) mod q_i,   kappa = floor(sum_i y_i/q_i + 1/2)   (centred x)
// and mode 1 (ScaleDown) only reads the bits from logQ-1 upwards, modes 2/3 only the bits below logQ.  So
//   * kappa comes from a 128-bit fixed-point sum of frac(r_i c_i / q_i), each term taken from r_i * floor(c_i 2^128 / q_i) and
//     under-estimated by less than 2^-68: the sum is low by less than 2^-63, and only a fraction within 2^-62 below 1/2 leaves
//     kappa undecided;
//   * modes 2/3 accumulate only the limbs below logQ of y_i M_i and kappa P -- exact, low limbs do not depend on high ones;
//   * mode 1 accumulates the limbs from two below the one holding bit logQ-1: dropping the lower limbs of every product costs less
//     than 2 units of that limb, which can change the rounded quotient only if (x + 2^(logQ-1)) mod 2^logQ is within a few units
//     of wrapping around.
// A workgroup that meets an undecided coefficient (probability ~2^-60 per coefficient on real data; certain for crafted values such
// as +-P/2) sets its flag, and the exact mixed-radix kernel, launched right behind with the flag array, redoes just those
// workgroups.  Result: identical bits, 18 x 8 (or 18 x 12) limb products instead of the recurrence's 2 x 153 + folds.
template <int K, int W, int LQ, bool HIGH>
__global__ void __launch_bounds__(128) crt_sum_kernel(const u64* __restrict__ rows, i64 n, int nslots_layout, const int* __restrict__ slot_of,
                                                      const int* __restrict__ idx, const u64* __restrict__ Mtab, const u64* __restrict__ cinv,
                                                      const u64* __restrict__ Pfull, const PrimeConst* __restrict__ pcs, int mode,
                                                      u64* __restrict__ out, int nl_out, unsigned char* __restrict__ block_flags) {
  constexpr int w = LQ >> 6, sw = (LQ - 1) >> 6;
  static_assert((LQ & 63) == 0 && w >= 2 && w + 1 < W, "sum-form CRT: logQ a multiple of 64 inside the product");
  constexpr int J0 = HIGH ? sw - 1 : 0, J1 = HIGH ? W : w;          // limbs [J0, J1) are accumulated
  const i64 poly = blockIdx.y;
  const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = j < n;
  const u64* base = rows + poly * nslots_layout * n;
  u64 x[W];
#pragma unroll
  for (int i = 0; i < W; ++i) x[i] = 0;
  int undecided = 0;
  if (active) {
    u64 f_lo = 0, f_hi = 0;        // fractional part of sum y_i/q_i, 128-bit fixed point
    u32 kint = 0;                  // its integer part
#pragma unroll 2
    for (int i = 0; i < K; ++i) {
      const PrimeConst pc = pcs[idx[i]];
      const u64 q = pc.q;
      const u64 r = base[(i64)(slot_of ? slot_of[i] : idx[i]) * n + j];
      const u64 c = cinv[i * 3], ch = cinv[i * 3 + 1], cl = cinv[i * 3 + 2];
      const u64 y = d_shoup(r, c, ch, q);
      // low 128 bits of r * (ch:cl)
      const u128 pl = (u128)r * cl;
      const u64 t_lo = (u64)pl, t_hi = (u64)(pl >> 64) + r * ch;
      const u128 fs = ((u128)f_hi << 64 | f_lo) + ((u128)t_hi << 64 | t_lo);
      kint += (fs < ((u128)t_hi << 64 | t_lo)) ? 1u : 0u;
      f_lo = (u64)fs; f_hi = (u64)(fs >> 64);
      // x[J0..J1) += y * M_i[J0..J1)
      const u64* Mi = Mtab + (i64)i * W;
      u64 carry = 0;
#pragma unroll
      for (int l = J0; l < J1; ++l) {
        const u128 s = (u128)y * Mi[l] + x[l] + carry;
        x[l] = (u64)s;
        carry = (u64)(s >> 64);
      }
    }
    // centred: kappa = floor(phi + 1/2); the computed fraction is low by less than 2^-63
    const u32 kappa = kint + (u32)(f_hi >> 63);
    undecided |= (f_hi >= 0x7ffffffffffffffcull && f_hi < 0x8000000000000000ull) ? 1 : 0;
    // x[J0..J1) -= kappa * P[J0..J1)
    u64 carry = 0, borrow = 0;
#pragma unroll
    for (int l = J0; l < J1; ++l) {
      const u128 t = (u128)kappa * Pfull[l] + carry;
      carry = (u64)(t >> 64);
      const u64 tl = (u64)t, d = x[l] - tl, b1 = x[l] < tl, d2 = d - borrow, b2 = d < borrow;
      x[l] = d2;
      borrow = b1 | b2;
    }
    if (HIGH) {
      // limbs below J0 were dropped: the limb holding bit logQ-1 may be off by a few units
      const u64 u = x[sw] + ((u64)1 << 63);
      undecided |= (u + 8 < 16) ? 1 : 0;
    }
    crt_store_fixed<W, W, LQ>(x, mode, out, poly, n, j, nl_out);
  }
  const int any = __syncthreads_or(undecided);
  if (threadIdx.x == 0) block_flags[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = any ? 1 : 0;
}

template <int K, int W, int LQ>
static int launch_crt_sum(fhesi_ctx* ctx, const CrtTables* t, const u64* d_rows, int nslots_layout, const int* d_slot_of, i64 npolys, int mode,
                          u64* d_out, int nl_out) {
  const int TB = 128;
  dim3 grid((unsigned)((ctx->phim + TB - 1) / TB), (unsigned)npolys);
  const size_t need = (size_t)grid.x * grid.y;
  if (need > t->flags_cap) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (t->d_flags) HIP_TRY(hipFree(t->d_flags));
    HIP_TRY(hipMalloc(&t->d_flags, need));
    t->flags_cap = need;
  }
  if (mode == 1) PROF_KERNEL(ctx, PROF_CRT, crt_sum_kernel<K, W, LQ, true>); else PROF_KERNEL(ctx, PROF_CRT, crt_sum_kernel<K, W, LQ, false>);
  if (mode == 1) crt_sum_kernel<K, W, LQ, true><<<grid, TB, 0, ctx->stream>>>(d_rows, ctx->phim, nslots_layout, d_slot_of, t->d_idx, t->d_M, t->d_cinv, t->d_P + (size_t)K * W,
                                                                               ctx->d_pc, mode, d_out, nl_out, t->d_flags);
  else crt_sum_kernel<K, W, LQ, false><<<grid, TB, 0, ctx->stream>>>(d_rows, ctx->phim, nslots_layout, d_slot_of, t->d_idx, t->d_M, t->d_cinv, t->d_P + (size_t)K * W,
                                                                      ctx->d_pc, mode, d_out, nl_out, t->d_flags);
  HIP_TRY(hipGetLastError());
  // exact clean-up of the flagged workgroups (normally none: every workgroup of this launch returns at once)
  if (ctx->opt.crt_skip_cleanup) return 0;      // test hook: shows that a crafted input really needs the clean-up
  return launch_crt_t<K, K, W, LQ>(ctx, t, d_rows, nslots_layout, d_slot_of, npolys, mode, 0, LQ, d_out, nl_out, t->d_flags);
}

// ----------------------------------------------------------------------------------------- key switch, limb mode: recombination
// The auxiliary-prime dot product in limb mode (kernels_ksaux.hip) returns, per output coefficient, NLB pairs of residues of the
// integers V_l = sum_k digit_k (*) K_{k,l}, where K_{k,l} is limb l (B bits) of the key polynomial's integer coefficient in [0, P).
// So S = sum_l V_l 2^(B l) = sum_k digit_k (*) K_k is the dot product as an integer, and ApplyKeySwitch's result
// (toPoly of the dot product, then ReduceCoefficients: FHE-SI.cpp:255-256) is the centred residue of S modulo P, reduced modulo 2^logQ.
//   V_l: Garner from the two residues, centred modulo q_0 q_1 (|V_l| < q_0 q_1 / 2 by the plan's bound), made non-negative by +2^119;
//   x = D + sum_l (V_l + 2^119) 2^(B l)  with  D = 2^m P - sum_l 2^(119 + B l)  (host constant), so x = S + 2^m P in (0, 2^(m+1) P);
//   quotient estimate from the top two limbs and floor(2^(64(W-2)+128) / P) (never above, at most 2 below), remainder, at most two
//   corrections, centring -- all exact integer arithmetic on W + 1 limbs -- then the mode-2 store of the sum-form kernel.
// A32: the residues are those of the four 30-bit auxiliary primes (kernels_aux32.hip), [poly][NLB][4][n] u32, recombined by Garner's
// mixed radix with the constants of Garner32; otherwise the two 60-bit chain primes, [poly][NLB][2][n] u64.
struct Garner32 { u32 p[4]; u32 c[6], cp[6]; u32 tw[4][2], twp[4][2]; };      // tw: the tail of a 2^15-point inverse per prime (1/2 | psi^-brv(1) / 2) and its quotient (kernels with S = 1)      // c = {p0^-1 mod p1, p0^-1 mod p2, p1^-1 mod p2, p0^-1 mod p3, p1^-1 mod p3, p2^-1 mod p3} and floor(c 2^32 / p_j)
__device__ __forceinline__ u32 g32_mul(u32 y, u32 c, u32 cp, u32 p) { const u32 r = y * c - __umulhi(y, cp) * p; return r >= p ? r - p : r; }     // y any u32 -> [0, p)
__device__ __forceinline__ u32 g32_mul_lazy(u32 y, u32 c, u32 cp, u32 p) { return y * c - __umulhi(y, cp) * p; }                               // y any u32 -> [0, 2p)
__device__ __forceinline__ u32 g32_sub(u32 a, u32 b, u32 p) { return a + 2 * p - b; }      // a, b below 2p -> a - b + 2p in (0, 4p): only ever the argument of a g32_mul
// S = 1 (A32 only): the rows are the two sub-inverses A, B of 2^15-point rows as ntt32_inv_kernel3 leaves them -- coefficient e < 2^14 is
// (A_e + B_e) / 2, coefficient e + 2^14 is (A_e - B_e) psi^-brv(1) / 2 (ntt32_tail_kernel's arithmetic, taken here in the loader: the rows are
// read once instead of being rewritten by a pass of their own).
template <int W, int LQ, int B, int NLB, bool A32, int S = 0>
__global__ void __launch_bounds__(128) ks_recombine_kernel(const u64* __restrict__ o, i64 n, u64 q0, u64 q1, u64 q0inv, u64 q0inv_sh, u64 half_hi, u64 half_lo,
                                                           u64 a_hi, u64 a_lo, const u64* __restrict__ consts /* D[W+1], pinv lo, hi */,
                                                           const u64* __restrict__ Pfull, const u64* __restrict__ halfP, u64* __restrict__ out, int nl_out, Garner32 gc) {
  const i64 poly = blockIdx.y;
  const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  u64 x[W + 1];
  u32 cnt[W + 4];
#pragma unroll
  for (int i = 0; i <= W; ++i) x[i] = consts[i];
#pragma unroll
  for (int i = 0; i < W + 4; ++i) cnt[i] = 0;
  const u128 half = ((u128)half_hi << 64) | half_lo, A = ((u128)a_hi << 64) | a_lo;
  const u64* base = o + poly * NLB * 2 * n + j;
  const u32* base32 = reinterpret_cast<const u32*>(o) + poly * NLB * 4 * n + (S ? (j & ((i64)(1 << 14) - 1)) : j);
  const int up = S ? (int)(j >> 14) : 0;           // (uniform per workgroup)
  auto tail = [&](u32 Av, u32 Bv, int a) -> u32 {  // both below p
    const u32 p = gc.p[a];
    return g32_mul(up ? Av + p - Bv : Av + Bv, gc.tw[a][up], gc.twp[a][up], p);
  };
  // all the residues of the coefficient at once (the kernel waits on these loads, not on its arithmetic: one round of 4 NLB loads in
  // flight instead of a round per limb)
  constexpr bool PRE = A32 && NLB <= 16;           // (the 30 limbs of the stress chain would cost a wave per SIMD)
  u32 vin[PRE ? NLB : 1][4];
  if (PRE) {
#pragma unroll
    for (int l = 0; l < NLB; ++l)
#pragma unroll
      for (int a = 0; a < 4; ++a) vin[l][a] = __builtin_nontemporal_load(&base32[(i64)(l * 4 + a) * n]);
  }
#pragma unroll
  for (int l = 0; l < NLB; ++l) {
    u128 V;
    if (A32) {
      u32 v0, v1, v2, v3;
      if (PRE) { v0 = vin[l][0]; v1 = vin[l][1]; v2 = vin[l][2]; v3 = vin[l][3]; }
      else if (S) {
        const u32* r0 = base32 + (i64)(l * 4) * n;
        const u32 a0 = r0[0], b0 = r0[1 << 14], a1 = r0[n], b1 = r0[n + (1 << 14)], a2 = r0[2 * n], b2 = r0[2 * n + (1 << 14)], a3 = r0[3 * n], b3 = r0[3 * n + (1 << 14)];
        v0 = tail(a0, b0, 0); v1 = tail(a1, b1, 1); v2 = tail(a2, b2, 2); v3 = tail(a3, b3, 3);
      }
      else { v0 = base32[(i64)(l * 4 + 0) * n]; v1 = base32[(i64)(l * 4 + 1) * n]; v2 = base32[(i64)(l * 4 + 2) * n]; v3 = base32[(i64)(l * 4 + 3) * n]; }
      const u32 p0 = gc.p[0], p1 = gc.p[1], p2 = gc.p[2], p3 = gc.p[3];
      const u32 x1 = v0;                                                        // all four primes lie in (2^29, 2^30): a residue of one is below twice any other
      // (inner products stay lazy, below 2p; the mixed-radix digits x2, x3, x4 themselves are reduced)
      const u32 x2 = g32_mul(g32_sub(v1, x1, p1), gc.c[0], gc.cp[0], p1);
      const u32 x3 = g32_mul(g32_sub(g32_mul_lazy(g32_sub(v2, x1, p2), gc.c[1], gc.cp[1], p2), x2, p2), gc.c[2], gc.cp[2], p2);
      const u32 x4 = g32_mul(g32_sub(g32_mul_lazy(g32_sub(g32_mul_lazy(g32_sub(v3, x1, p3), gc.c[3], gc.cp[3], p3), x2, p3), gc.c[4], gc.cp[4], p3), x3, p3),
                             gc.c[5], gc.cp[5], p3);
      V = (u128)((u64)x3 + (u64)p2 * x4) * ((u64)p0 * p1) + ((u64)x1 + (u64)p0 * x2);       // x1 + p0 (x2 + p1 (x3 + p2 x4)), below p0 p1 p2 p3
    } else {
    const u64 v0 = base[(i64)(l * 2 + 0) * n], v1 = base[(i64)(l * 2 + 1) * n];
    const u64 v0r = v0 >= q1 ? v0 - q1 : v0;
    const u64 t = d_shoup(d_submod(v1, v0r, q1), q0inv, q0inv_sh, q1);
    V = (u128)q0 * t + v0;                                       // in [0, q_0 q_1)
    }
    if (V > half) V -= A;                                        // centred (two's complement in 128 bits)
    V += (u128)1 << 119;                                         // non-negative, below 2^120
    const int s = B * l, wd = s >> 6, bt = s & 63;               // compile-time after unrolling
    const u64 lo = (u64)V, hi = (u64)(V >> 64);
    const u64 p0 = lo << bt, p1 = bt ? ((lo >> ((64 - bt) & 63)) | (hi << bt)) : hi, p2 = bt ? (hi >> ((64 - bt) & 63)) : 0;
    // the three words go into their limbs; the carries out of a limb are only COUNTED here and added in one pass below (a carry chain
    // through all the higher limbs per term was 40 % of the kernel)
    { const u64 s0 = x[wd] + p0; cnt[wd + 1] += s0 < p0 ? 1u : 0u; x[wd] = s0; }
    if (wd + 1 <= W) { const u64 s1 = x[wd + 1] + p1; cnt[wd + 2] += s1 < p1 ? 1u : 0u; x[wd + 1] = s1; }
    if (wd + 2 <= W && bt) { const u64 s2 = x[wd + 2] + p2; cnt[wd + 3] += s2 < p2 ? 1u : 0u; x[wd + 2] = s2; }
  }
  {
    u64 carry = 0;
#pragma unroll
    for (int i = 1; i <= W; ++i) {
      const u128 sum = (u128)x[i] + cnt[i] + carry;
      x[i] = (u64)sum;
      carry = (u64)(sum >> 64);
    }
  }
  // x = S + 2^m P, below 2^(64 (W-1) + 63): x[W] = 0
  const u64 t0 = x[W - 2], t1 = x[W - 1], p0 = consts[W + 1], p1 = consts[W + 2];
  u128 mid = (u128)t1 * p0 + (u64)(((u128)t0 * p0) >> 64);
  const u128 add = (u128)t0 * p1;
  const u128 mid2 = mid + add;
  u128 qh = (u128)t1 * p1 + (mid2 >> 64) + ((mid2 < add) ? ((u128)1 << 64) : 0);
  const u64 qhat = (u64)qh;                                      // below 2^(m+1) < 2^63
  {
    u64 carry = 0, borrow = 0;
#pragma unroll
    for (int i = 0; i < W; ++i) {
      const u128 t = (u128)qhat * Pfull[i] + carry;
      carry = (u64)(t >> 64);
      const u64 tl = (u64)t, d = x[i] - tl, b1 = x[i] < tl, d2 = d - borrow, b2 = d < borrow;
      x[i] = d2;
      borrow = b1 | b2;
    }
  }
  auto ge = [&](const u64* __restrict__ c) -> bool {             // x[0..W) >= c[0..W)
    bool gt = false, decided = false;
#pragma unroll
    for (int i = W - 1; i >= 0; --i) { const u64 h = c[i]; if (!decided && x[i] != h) { gt = x[i] > h; decided = true; } }
    return gt || !decided;
  };
  auto subP = [&]() {
    u64 borrow = 0;
#pragma unroll
    for (int i = 0; i < W; ++i) { const u64 p = Pfull[i], d = x[i] - p, b1 = x[i] < p, d2 = d - borrow, b2 = d < borrow; x[i] = d2; borrow = b1 | b2; }
  };
  if (ge(Pfull)) subP();
  if (ge(Pfull)) subP();
  // centre: x > (P-1)/2  ->  x - P   (DoubleCRT.cpp:375-376)
  {
    bool gt = false, decided = false;
#pragma unroll
    for (int i = W - 1; i >= 0; --i) { const u64 h = halfP[i]; if (!decided && x[i] != h) { gt = x[i] > h; decided = true; } }
    if (gt) subP();
  }
  u64 y[W];
#pragma unroll
  for (int i = 0; i < W; ++i) y[i] = x[i];
  crt_store_fixed<W, W, LQ>(y, 2, out, poly, n, j, nl_out);
}

// The same recombination for ANY chain shape: limb count W, logQ, limb width B and limb count NLB are run-time values, the W + 1 limbs of
// x live in LDS (limb-major, one column per thread: run-time limb indices cost nothing there).  The compile-time instantiations above
// serve the shapes the benchmarks run; this one makes limb mode (and the 30-bit auxiliary primes) available to every chain.
template <int MAXW, bool A32>
__global__ void __launch_bounds__(128) ks_recombine_generic_kernel(const u64* __restrict__ o, i64 n, i64 nrow, i64 fold_q, int W, int LQ, int B, int NLB, u64 q0, u64 q1, u64 q0inv, u64 q0inv_sh,
                                                                   u64 half_hi, u64 half_lo, u64 a_hi, u64 a_lo, const u64* __restrict__ consts /* D[W+1], pinv lo, hi */,
                                                                   const u64* __restrict__ Pfull, const u64* __restrict__ halfP, u64* __restrict__ out, int nl_out, Garner32 gc) {
  __shared__ u64 xs[(MAXW + 1) * 128];
#define X(i) xs[(i) * 128 + threadIdx.x]
  const i64 poly = blockIdx.y;
  const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  for (int i = 0; i <= W; ++i) X(i) = consts[i];
  const u128 half = ((u128)half_hi << 64) | half_lo, A = ((u128)a_hi << 64) | a_lo;
  // rows of nrow elements (nrow = n except on the linear-convolution rings, where the rows are the 2^14-point products)
  const u64* base = o + poly * NLB * 2 * nrow;
  const u32* base32 = reinterpret_cast<const u32*>(o) + poly * NLB * 4 * nrow;
  // the centred integer at position `pos` of limb row l (two's complement in 128 bits)
  auto centred = [&](int l, i64 pos) -> u128 {
    u128 V;
    if (A32) {
      const u32 v0 = base32[(i64)(l * 4 + 0) * nrow + pos], v1 = base32[(i64)(l * 4 + 1) * nrow + pos], v2 = base32[(i64)(l * 4 + 2) * nrow + pos],
                v3 = base32[(i64)(l * 4 + 3) * nrow + pos];
      const u32 p0 = gc.p[0], p1 = gc.p[1], p2 = gc.p[2], p3 = gc.p[3];
      const u32 x1 = v0;
      // (inner products stay lazy, below 2p; the mixed-radix digits x2, x3, x4 themselves are reduced)
      const u32 x2 = g32_mul(g32_sub(v1, x1, p1), gc.c[0], gc.cp[0], p1);
      const u32 x3 = g32_mul(g32_sub(g32_mul_lazy(g32_sub(v2, x1, p2), gc.c[1], gc.cp[1], p2), x2, p2), gc.c[2], gc.cp[2], p2);
      const u32 x4 = g32_mul(g32_sub(g32_mul_lazy(g32_sub(g32_mul_lazy(g32_sub(v3, x1, p3), gc.c[3], gc.cp[3], p3), x2, p3), gc.c[4], gc.cp[4], p3), x3, p3),
                             gc.c[5], gc.cp[5], p3);
      V = (u128)((u64)x3 + (u64)p2 * x4) * ((u64)p0 * p1) + ((u64)x1 + (u64)p0 * x2);
    } else {
      const u64 v0 = base[(i64)(l * 2 + 0) * nrow + pos], v1 = base[(i64)(l * 2 + 1) * nrow + pos];
      const u64 v0r = v0 >= q1 ? v0 - q1 : v0;
      const u64 t = d_shoup(d_submod(v1, v0r, q1), q0inv, q0inv_sh, q1);
      V = (u128)q0 * t + v0;
    }
    if (V > half) V -= A;
    return V;
  };
  // A32: the residues of the next limb are fetched while this one is recombined (the kernel waits on its loads), and on the
  // linear-convolution rings the fold  S_j - S_(j+q') -+ (S_n - S_(n+q'))  is taken on the RESIDUES (it is linear and the plan keeps the
  // combination below A / 2), so one Garner recombination per limb serves instead of four
  u32 cur[4] = {0, 0, 0, 0}, nxt[4] = {0, 0, 0, 0};
  auto fetch = [&](int l, u32 (&v)[4]) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const u32* __restrict__ row = base32 + (i64)(l * 4 + a) * nrow;
      if (!fold_q) { v[a] = row[j]; continue; }
      const u32 p = gc.p[a];
      u32 r;
      if (fold_q > 0) {        // m = 2q':  S_j - S_(j+q') -+ (S_n - S_(n+q'))
        const u32 s0 = row[j], s1 = row[j + fold_q], t0 = row[n], t1 = n + fold_q < nrow ? row[n + fold_q] : 0u;      // all below p
        r = s0 + (p - s1) + ((j & 1) ? t0 + (p - t1) : t1 + (p - t0));                     // below 4p
      } else {                 // m prime (offset -fold_q = m, n = m - 1):  S_j + S_(j+m) - S_(m-1)
        const u32 s0 = row[j], s1 = j - fold_q < nrow ? row[j - fold_q] : 0u, t0 = row[n];
        r = s0 + s1 + (p - t0);                                                               // below 3p
      }
      r = r >= 2 * p ? r - 2 * p : r;
      v[a] = r >= p ? r - p : r;
    }
  };
  auto garner = [&](const u32 (&v)[4]) -> u128 {
    const u32 p0 = gc.p[0], p1 = gc.p[1], p2 = gc.p[2], p3 = gc.p[3];
    const u32 x1 = v[0];
    const u32 x2 = g32_mul(g32_sub(v[1], x1, p1), gc.c[0], gc.cp[0], p1);
    const u32 x3 = g32_mul(g32_sub(g32_mul_lazy(g32_sub(v[2], x1, p2), gc.c[1], gc.cp[1], p2), x2, p2), gc.c[2], gc.cp[2], p2);
    const u32 x4 = g32_mul(g32_sub(g32_mul_lazy(g32_sub(g32_mul_lazy(g32_sub(v[3], x1, p3), gc.c[3], gc.cp[3], p3), x2, p3), gc.c[4], gc.cp[4], p3), x3, p3),
                           gc.c[5], gc.cp[5], p3);
    u128 V = (u128)((u64)x3 + (u64)p2 * x4) * ((u64)p0 * p1) + ((u64)x1 + (u64)p0 * x2);
    if (V > half) V -= A;
    return V;
  };
  if (A32) fetch(0, cur);
  for (int l = 0; l < NLB; ++l) {
    u128 V;
    if (A32) {
      if (l + 1 < NLB) fetch(l + 1, nxt);
      V = garner(cur);
#pragma unroll
      for (int a = 0; a < 4; ++a) cur[a] = nxt[a];
    } else {
      V = centred(l, j);
      if (fold_q > 0) {      // (the two-prime auxiliary form never runs on a linear-convolution ring: ks_limb_plan)
        // S (degree < 2n - 1) modulo X^q' + 1: R_j = S_j - S_(j+q');  modulo Phi_m = 1 - X + X^2 - ... + X^(q'-1) (degree n = q' - 1):
        // out_j = R_j - (-1)^j R_n,  j < n   (Phi_m is monic, so this is the exact integer remainder)
        const u128 top = centred(l, n) - centred(l, n + fold_q);
        V -= centred(l, j + fold_q);
        if (j & 1) V += top; else V -= top;
      }
    }
    V += (u128)1 << 119;
    const int s = B * l, wd = s >> 6, bt = s & 63;
    const u64 lo = (u64)V, hi = (u64)(V >> 64);
    const u64 p0 = lo << bt, p1 = bt ? ((lo >> ((64 - bt) & 63)) | (hi << bt)) : hi, p2 = bt ? (hi >> ((64 - bt) & 63)) : 0;
    u64 carry = 0;
    for (int i = wd; i <= W; ++i) {
      const u64 add = i == wd ? p0 : (i == wd + 1 ? p1 : (i == wd + 2 ? p2 : 0));
      if (i > wd + 2 && !carry) break;
      const u128 sum = (u128)X(i) + add + carry;
      X(i) = (u64)sum;
      carry = (u64)(sum >> 64);
    }
  }
  const u64 t0 = X(W - 2), t1 = X(W - 1), pr0 = consts[W + 1], pr1 = consts[W + 2];
  u128 mid = (u128)t1 * pr0 + (u64)(((u128)t0 * pr0) >> 64);
  const u128 add = (u128)t0 * pr1;
  const u128 mid2 = mid + add;
  u128 qh = (u128)t1 * pr1 + (mid2 >> 64) + ((mid2 < add) ? ((u128)1 << 64) : 0);
  const u64 qhat = (u64)qh;
  {
    u64 carry = 0, borrow = 0;
    for (int i = 0; i < W; ++i) {
      const u128 t = (u128)qhat * Pfull[i] + carry;
      carry = (u64)(t >> 64);
      const u64 xi = X(i), tl = (u64)t, d = xi - tl, b1 = xi < tl, d2 = d - borrow, b2 = d < borrow;
      X(i) = d2;
      borrow = b1 | b2;
    }
  }
  auto ge = [&](const u64* __restrict__ c, bool strict) -> bool {      // x[0..W) >= c (or > c)
    for (int i = W - 1; i >= 0; --i) { const u64 h = c[i], xi = X(i); if (xi != h) return xi > h; }
    return !strict;
  };
  auto subP = [&]() {
    u64 borrow = 0;
    for (int i = 0; i < W; ++i) { const u64 pp = Pfull[i], xi = X(i), d = xi - pp, b1 = xi < pp, d2 = d - borrow, b2 = d < borrow; X(i) = d2; borrow = b1 | b2; }
  };
  if (ge(Pfull, false)) subP();
  if (ge(Pfull, false)) subP();
  if (ge(halfP, true)) subP();                                  // centre: x > (P-1)/2  ->  x - P   (DoubleCRT.cpp:375-376)
  // mode-2 store: the centred residue modulo 2^logQ, two's complement, coefficient-major
  const u64 sf = (X(W - 1) >> 63) ? ~0ull : 0ull;
  const int sw = (LQ - 1) >> 6, sb = (LQ - 1) & 63;
  const u64 hbit = ((sw < W ? X(sw) : sf) >> sb) & 1;
  u64* op = out + (poly * n + j) * nl_out;
  for (int i = 0; i < nl_out; ++i) {
    u64 val = i < W ? X(i) : sf;
    const int bits_left = LQ - 64 * i;
    if (bits_left <= 0) val = hbit ? ~0ull : 0ull;
    else if (bits_left < 64) { const u64 mask = (1ull << (bits_left & 63)) - 1; val = hbit ? (val | ~mask) : (val & mask); }
    op[i] = val;
  }
#undef X
}

static int garner32_consts(fhesi_ctx* ctx, Garner32* gc) {
  const u32* p = aux32_primes(ctx);
  if (!p) return 1;
  for (int i = 0; i < 4; ++i) gc->p[i] = p[i];
  const int pairs[6][2] = {{0, 1}, {0, 2}, {1, 2}, {0, 3}, {1, 3}, {2, 3}};
  for (int e = 0; e < 6; ++e) {
    const u64 pj = p[pairs[e][1]], c = hm::invmod(p[pairs[e][0]] % pj, pj);
    gc->c[e] = (u32)c; gc->cp[e] = (u32)((c << 32) / pj);
  }
  return 0;
}
template <int W, int LQ, int B, int NLB, bool A32, int S = 0>
static int launch_ks_recombine_t(fhesi_ctx* ctx, const CrtTables* t, const fhesi_ksk* k, const u64* d_o, i64 npolys, u64* d_out, int nl_out) {
  const u64 q0 = ctx->q[0], q1 = ctx->q[1];
  const u64 inv = hm::invmod(q0 % q1, q1);
  u128 A = (u128)q0 * q1;
  Garner32 gc{};
  if (A32) {
    if (garner32_consts(ctx, &gc)) return 1;
    A = (u128)((u64)gc.p[0] * gc.p[1]) * ((u64)gc.p[2] * gc.p[3]);
  }
  const u128 half = (A - 1) / 2;
  dim3 grid((unsigned)((ctx->phim + 127) / 128), (unsigned)npolys);
  if (S && aux32_tail_consts(ctx, gc.tw, gc.twp)) return 1;
  ks_recombine_kernel<W, LQ, B, NLB, A32, S><<<grid, 128, 0, ctx->stream>>>(d_o, ctx->phim, q0, q1, inv, hm::shoup(inv, q1), (u64)(half >> 64), (u64)half, (u64)(A >> 64), (u64)A,
                                                                          k->d_limb_consts, t->d_P + (size_t)t->nidx * t->W, t->d_halfP, d_out, nl_out, gc);
  HIP_TRY(hipGetLastError());
  return 0;
}
// ---- centred limbs (fhesi_ksk::aux_centred): the limb product sums V_l = sum_k digit_k (*) K_{k,l} come from the CENTRED integer coefficients
// of the matrix (top limb signed), and S = sum_l V_l 2^(B l) is below P / 2 in magnitude by the build-time check -- S is its own centred
// residue modulo the chain product, so toPoly + ReduceCoefficients (FHE-SI.cpp:255-256) leave exactly the centred residue of S modulo 2^logQ:
// only the words below bit logQ are formed, no quotient by P, no comparison with P.  V_l: Garner over the four 30-bit residues, centred,
// + 2^119 to make it non-negative (the constant sum_l 2^(119 + B l) is taken off the start value).  fold: the linear-convolution rings,
// on the residues, as in ks_recombine_generic_kernel.  NWORDS = 64-bit words below bit logQ (8: logQ <= 512, 16: logQ <= 1024).
struct CentredConsts { u64 d[16]; };
// NLBF, BF, LQF > 0: the limb count, limb width and logQ as compile-time constants (the metric ring's 7 x 74 bits, logQ = 512; plain rows only):
// the limb loop unrolls, the bit offsets are static and all 4 NLBF residues of the coefficient are requested before the first is used.
// FS = 1 (m = 2q'), 2 (m prime): rows of 2^15 on a linear-convolution ring, left as their two sub-inverses -- the fold and the tail stage are both
// taken in the loader: the (up to) four positions a coefficient is folded from are worked out once (they do not depend on the limb or the prime),
// their sub-inverse pairs are loaded unconditionally, summed per half of the row with their signs and multiplied by the two tail constants.
template <int NWORDS, int NLBF = 0, int BF = 0, int LQF = 0, int FS = 0>
__global__ void __launch_bounds__(128) ks_recombine_centred_kernel(const u32* __restrict__ o32, i64 n, i64 nrow, i64 fold_q, int S /* 1: rows of 2^15 left as their two sub-inverses (the tail stage is taken here) */, int LQ_, int B_, int NLB_, u64 half_hi, u64 half_lo,
                                                                   u64 a_hi, u64 a_lo, CentredConsts cc, u64* __restrict__ out, int nl_out, Garner32 gc) {
  constexpr bool FIX = NLBF > 0;
  const int LQ = FIX ? LQF : LQ_, B = FIX ? BF : B_, NLB = FIX ? NLBF : NLB_;
  const i64 poly = blockIdx.y;
  const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  u64 x[NWORDS];
#pragma unroll
  for (int i = 0; i < NWORDS; ++i) x[i] = cc.d[i];
  // FIX: the sum is kept as 2 NWORDS chunks of 32 bits in 64-bit accumulators (every limb adds five chunks, one multiply-add by 1 each; the
  // carries are resolved once at the end).  The rippled form below costs ~6 instructions per word and limb: 40 % of the kernel.
  u64 acc[FIX ? 2 * NWORDS : 1];
  if constexpr (FIX) {
#pragma unroll
    for (int c = 0; c < 2 * NWORDS; ++c) acc[c] = (u32)(cc.d[c >> 1] >> (32 * (c & 1)));
  }
  const u128 half = ((u128)half_hi << 64) | half_lo, A = ((u128)a_hi << 64) | a_lo;
  const u32* base32 = o32 + poly * NLB * 4 * nrow;
  u32 cur[4] = {0, 0, 0, 0}, nxt[4] = {0, 0, 0, 0};
  u32 vin[FIX ? NLBF : 1][4];
  if constexpr (FIX) {
    if (!S) {
#pragma unroll
      for (int l = 0; l < NLBF; ++l)
#pragma unroll
        for (int a = 0; a < 4; ++a) vin[l][a] = __builtin_nontemporal_load(&base32[(i64)(l * 4 + a) * nrow + j]);
    } else {
      // power-of-two rows of 2^15 left as their two sub-inverses (the stress ring): the tail stage in the loader -- coefficient e of the lower
      // half is (A_e + B_e) / 2, of the upper half (A_e - B_e) psi^-brv(1) / 2
      const i64 e = j & ((i64)(1 << 14) - 1);
      const int up = (int)(j >> 14);                   // (uniform per workgroup)
#pragma unroll
      for (int l = 0; l < NLBF; ++l)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const u32* __restrict__ row = base32 + (i64)(l * 4 + a) * nrow;
          const u32 p = gc.p[a], Av = __builtin_nontemporal_load(&row[e]), Bv = __builtin_nontemporal_load(&row[e + (1 << 14)]);
          vin[l][a] = g32_mul(up ? Av + p - Bv : Av + Bv, gc.tw[a][up], gc.twp[a][up], p);
        }
    }
  }
  constexpr int NT = FS == 1 ? 4 : 3;
  u32 eb[NT];
  bool up[NT], ok[NT], ng[NT];
  if constexpr (FS != 0) {
    const i64 off = fold_q > 0 ? fold_q : -fold_q;
    const i64 e[4] = {j, j + off, n, n + off};
    // m = 2q':  S_j - S_(j+q') -+ (S_n - S_(n+q'))  (upper signs for even j);   m prime:  S_j + S_(j+m) - S_(m-1)
    const bool sg[4] = {false, FS == 1, FS == 1 ? !(j & 1) : true, (j & 1) != 0};
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      ok[k] = e[k] < nrow;
      const u32 idx = ok[k] ? (u32)e[k] : 0u;
      up[k] = idx >= (1u << 14);
      eb[k] = idx & ((1u << 14) - 1);
      ng[k] = sg[k];
    }
  }
  auto fetch = [&](int l, u32 (&v)[4]) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const u32* __restrict__ row = base32 + (i64)(l * 4 + a) * nrow;
      if constexpr (FS != 0) {
        const u32 p = gc.p[a], twop = 2 * p;
        u32 Av[NT], Bv[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k) { Av[k] = row[eb[k]]; Bv[k] = row[eb[k] + (1 << 14)]; }
        u32 lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < NT; ++k) {
          const u32 t = up[k] ? Av[k] + p - Bv[k] : Av[k] + Bv[k];                  // at most 2p
          const u32 val = ok[k] ? (ng[k] ? twop - t : t) : 0u;
          const u32 l2 = lo + (up[k] ? 0u : val), h2 = hi + (up[k] ? val : 0u);     // below 4p
          lo = min(l2, l2 - twop); hi = min(h2, h2 - twop);
        }
        u32 r = g32_mul_lazy(lo, gc.tw[a][0], gc.twp[a][0], p) + g32_mul_lazy(hi, gc.tw[a][1], gc.twp[a][1], p);      // below 4p
        r = min(r, r - twop);
        v[a] = min(r, r - p);
        continue;
      }
      if (S) {                 // (power-of-two rows of 2^15: no fold)
        const i64 e = j & ((i64)(1 << 14) - 1);
        const int up = (int)(j >> 14);
        const u32 p = gc.p[a], Av = row[e], Bv = row[e + (1 << 14)];
        v[a] = g32_mul(up ? Av + p - Bv : Av + Bv, gc.tw[a][up], gc.twp[a][up], p);
        continue;
      }
      if (!fold_q) { v[a] = row[j]; continue; }
      const u32 p = gc.p[a];
      u32 r;
      if (fold_q > 0) {        // m = 2q':  S_j - S_(j+q') -+ (S_n - S_(n+q'))
        const u32 s0 = row[j], s1 = row[j + fold_q], t0 = row[n], t1 = n + fold_q < nrow ? row[n + fold_q] : 0u;
        r = s0 + (p - s1) + ((j & 1) ? t0 + (p - t1) : t1 + (p - t0));
      } else {                 // m prime:  S_j + S_(j+m) - S_(m-1)
        const u32 s0 = row[j], s1 = j - fold_q < nrow ? row[j - fold_q] : 0u, t0 = row[n];
        r = s0 + s1 + (p - t0);
      }
      r = r >= 2 * p ? r - 2 * p : r;
      v[a] = r >= p ? r - p : r;
    }
  };
  if constexpr (!FIX) fetch(0, cur);
#pragma unroll
  for (int l = 0; l < (FIX ? NLBF : NLB); ++l) {
    if constexpr (FIX) {
#pragma unroll
      for (int a = 0; a < 4; ++a) cur[a] = vin[l][a];
    } else if (l + 1 < NLB) fetch(l + 1, nxt);
    const u32 p0 = gc.p[0], p1 = gc.p[1], p2 = gc.p[2], p3 = gc.p[3];
    const u32 x1 = cur[0];
    const u32 x2 = g32_mul(g32_sub(cur[1], x1, p1), gc.c[0], gc.cp[0], p1);
    const u32 x3 = g32_mul(g32_sub(g32_mul_lazy(g32_sub(cur[2], x1, p2), gc.c[1], gc.cp[1], p2), x2, p2), gc.c[2], gc.cp[2], p2);
    const u32 x4 = g32_mul(g32_sub(g32_mul_lazy(g32_sub(g32_mul_lazy(g32_sub(cur[3], x1, p3), gc.c[3], gc.cp[3], p3), x2, p3), gc.c[4], gc.cp[4], p3), x3, p3),
                           gc.c[5], gc.cp[5], p3);
    u128 V = (u128)((u64)x3 + (u64)p2 * x4) * ((u64)p0 * p1) + ((u64)x1 + (u64)p0 * x2);
    if (V > half) V -= A;
    V += (u128)1 << 119;
    if constexpr (!FIX) {
#pragma unroll
      for (int a = 0; a < 4; ++a) cur[a] = nxt[a];
    }
    const int s = B * l, wd = s >> 6, bt = s & 63;
    const u64 lo = (u64)V, hi = (u64)(V >> 64);
    if constexpr (FIX) {
      const int c0 = s >> 5, sh = s & 31;              // (compile-time after unrolling)
      const u64 t0 = lo << sh, t1 = sh ? ((hi << sh) | (lo >> ((64 - sh) & 63))) : hi;
      const u32 t2 = sh ? (u32)(hi >> ((64 - sh) & 63)) : 0u;
      const u32 w[5] = {(u32)t0, (u32)(t0 >> 32), (u32)t1, (u32)(t1 >> 32), t2};
#pragma unroll
      for (int k = 0; k < 5; ++k)
        if (c0 + k < 2 * NWORDS) {                     // (chunks above bit logQ are not formed)
          u64 r, cy;
          asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(r), "=s"(cy) : "v"(w[k]), "v"(acc[c0 + k]));
          acc[c0 + k] = r;
        }
    } else {
    const u64 q0 = lo << bt, q1 = bt ? ((lo >> ((64 - bt) & 63)) | (hi << bt)) : hi, q2 = bt ? (hi >> ((64 - bt) & 63)) : 0;
    u64 carry = 0;
#pragma unroll
    for (int i = 0; i < NWORDS; ++i) {              // (words above bit logQ are not formed: the sum is taken modulo 2^(64 NWORDS))
      if (i >= wd) {
        const u64 add = i == wd ? q0 : (i == wd + 1 ? q1 : (i == wd + 2 ? q2 : 0));
        const u64 s1 = x[i] + add, c1 = s1 < add, s2 = s1 + carry, c2 = s2 < carry;
        x[i] = s2;
        carry = c1 | c2;
      }
    }
    }
  }
  if constexpr (FIX) {                                 // the carries, once: every accumulator is below 2^35
    u64 carry = 0;
#pragma unroll
    for (int i = 0; i < NWORDS; ++i) {
      const u64 a0 = acc[2 * i] + carry, a1 = acc[2 * i + 1] + (a0 >> 32);
      x[i] = (u64)(u32)a0 | (a1 << 32);
      carry = a1 >> 32;
    }
  }
  // centred residue modulo 2^logQ, two's complement, coefficient-major (mode 2 of crt_store_fixed)
  const int sw = (LQ - 1) >> 6, sb = (LQ - 1) & 63;
  u64 hbit = 0;
#pragma unroll
  for (int i = 0; i < NWORDS; ++i) if (i == sw) hbit = (x[i] >> sb) & 1;
  u64* o = out + (poly * n + j) * nl_out;
#pragma unroll
  for (int i = 0; i < NWORDS; ++i) {
    u64 val = x[i];
    const int bits_left = LQ - 64 * i;
    if (bits_left <= 0) val = hbit ? ~0ull : 0ull;
    else if (bits_left < 64) { const u64 mask = (1ull << (bits_left & 63)) - 1; val = hbit ? (val | ~mask) : (val & mask); }
    x[i] = val;
  }
  // Coefficient-major output: a lane owns NWORDS consecutive 8-byte limbs, so a direct store instruction of the wave touches 32-64 cache lines
  // for 8-16 bytes each.  When the wave is whole (n a multiple of 64) and the caller's limb count is the kernel's, the wave's 64 x NWORDS limbs --
  // one contiguous block of memory -- pass through its own LDS rows (stride NWORDS + 1: conflict-free) and leave 512 contiguous bytes per
  // store instruction (round 5: the kernel was 43 % VALU busy at 3.7 TB/s).
  if (nl_out == NWORDS && (n & 63) == 0) {
    __shared__ u64 stg[2][64 * (NWORDS + 1)];
    const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    u64* __restrict__ st = stg[wv];
#pragma unroll
    for (int i = 0; i < NWORDS; ++i) st[lane * (NWORDS + 1) + i] = x[i];
    __builtin_amdgcn_wave_barrier();
    u64* __restrict__ ow = out + (poly * n + (j - lane)) * NWORDS;      // the wave's block
#pragma unroll
    for (int k = 0; k < NWORDS; ++k) {
      const u32 e = k * 64 + lane, tl = e / NWORDS, ii = e % NWORDS;   // (NWORDS a power of two: shifts)
      ow[e] = st[tl * (NWORDS + 1) + ii];
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < NWORDS; ++i) if (i < nl_out) o[i] = x[i];
  for (int i = NWORDS; i < nl_out; ++i) o[i] = hbit ? ~0ull : 0ull;
}
static int garner32_consts(fhesi_ctx* ctx, Garner32* gc);
static int launch_ks_recombine_centred(fhesi_ctx* ctx, const fhesi_ksk* k, const u64* d_o, i64 npolys, u64* d_out, int nl_out, bool tail_pending) {
  Garner32 gc{};
  if (garner32_consts(ctx, &gc)) return 1;
  const int S = tail_pending ? 1 : 0;
  if (S && aux32_tail_consts(ctx, gc.tw, gc.twp)) return 1;
  const u128 A = (u128)((u64)gc.p[0] * gc.p[1]) * ((u64)gc.p[2] * gc.p[3]), half = (A - 1) / 2;
  const int LQ = k->aux_logQ, B = k->aux_limb_bits, NLB = k->aux_rows, NW = LQ <= 512 ? 8 : 16;
  if (LQ > 1024) FHESI_FAIL("key switch, centred limbs: logQ=%d above 1024", LQ);
  // start value: - sum_l 2^(119 + B l)  modulo 2^(64 NW)
  CentredConsts cc{};
  {
    std::vector<u64> D((size_t)NW, 0);
    for (int l = 0; l < NLB; ++l) {
      const int bit = 119 + B * l;
      if (bit >= 64 * NW) continue;
      u64 borrow = (u64)1 << (bit & 63);
      for (int i = bit >> 6; i < NW && borrow; ++i) { const u64 v = D[i]; D[i] = v - borrow; borrow = v < borrow ? 1 : 0; }
    }
    for (int i = 0; i < NW; ++i) cc.d[i] = D[i];
  }
  const i64 nrow = aux32_row_len(ctx);
  const i64 fold = k->aux_fold;
  dim3 grid((unsigned)((ctx->phim + 127) / 128), (unsigned)npolys);
  if (NW == 16 && LQ == 1024 && B == 72 && NLB == 15 && !fold)         // the stress ring with a generated matrix (rows of 2^15: the tail stage in the loader when S)
    ks_recombine_centred_kernel<16, 15, 72, 1024><<<grid, 128, 0, ctx->stream>>>((const u32*)d_o, ctx->phim, nrow, fold, S, LQ, B, NLB, (u64)(half >> 64), (u64)half, (u64)(A >> 64), (u64)A, cc, d_out, nl_out, gc);
  else if (NW == 8 && LQ == 512 && B == 74 && NLB == 7 && !fold && !S)      // the metric ring with a generated matrix
    ks_recombine_centred_kernel<8, 7, 74, 512><<<grid, 128, 0, ctx->stream>>>((const u32*)d_o, ctx->phim, nrow, fold, S, LQ, B, NLB, (u64)(half >> 64), (u64)half, (u64)(A >> 64), (u64)A, cc, d_out, nl_out, gc);
  // (a compile-time instantiation <8, 8, 72, 512, FS = 1> for the reference drivers' ring at the metric's size, its residues fetched up front through the
  // run-time loader, was measured: 106 registers, 256 loads before the first Garner step -- crt class 8.7 -> 10.9 ms per step there, and the same
  // loader in front of the stress instantiation cost it 22.7 -> 26.8: the folds keep the run-time form, the compile-time forms their own loaders)
  else if (S && fold && NW == 8 && fold > 0) ks_recombine_centred_kernel<8, 0, 0, 0, 1><<<grid, 128, 0, ctx->stream>>>((const u32*)d_o, ctx->phim, nrow, fold, S, LQ, B, NLB, (u64)(half >> 64), (u64)half, (u64)(A >> 64), (u64)A, cc, d_out, nl_out, gc);
  else if (S && fold && NW == 8) ks_recombine_centred_kernel<8, 0, 0, 0, 2><<<grid, 128, 0, ctx->stream>>>((const u32*)d_o, ctx->phim, nrow, fold, S, LQ, B, NLB, (u64)(half >> 64), (u64)half, (u64)(A >> 64), (u64)A, cc, d_out, nl_out, gc);
  else if (S && fold && fold > 0) ks_recombine_centred_kernel<16, 0, 0, 0, 1><<<grid, 128, 0, ctx->stream>>>((const u32*)d_o, ctx->phim, nrow, fold, S, LQ, B, NLB, (u64)(half >> 64), (u64)half, (u64)(A >> 64), (u64)A, cc, d_out, nl_out, gc);
  else if (S && fold) ks_recombine_centred_kernel<16, 0, 0, 0, 2><<<grid, 128, 0, ctx->stream>>>((const u32*)d_o, ctx->phim, nrow, fold, S, LQ, B, NLB, (u64)(half >> 64), (u64)half, (u64)(A >> 64), (u64)A, cc, d_out, nl_out, gc);
  else if (NW == 8) ks_recombine_centred_kernel<8><<<grid, 128, 0, ctx->stream>>>((const u32*)d_o, ctx->phim, nrow, fold, S, LQ, B, NLB, (u64)(half >> 64), (u64)half, (u64)(A >> 64), (u64)A, cc, d_out, nl_out, gc);
  else ks_recombine_centred_kernel<16><<<grid, 128, 0, ctx->stream>>>((const u32*)d_o, ctx->phim, nrow, fold, S, LQ, B, NLB, (u64)(half >> 64), (u64)half, (u64)(A >> 64), (u64)A, cc, d_out, nl_out, gc);
  HIP_TRY(hipGetLastError());
  return 0;
}

// does the recombination of this matrix take the tail of the 2^15-point inverse in its loader (so that launch_ntt32_inv leaves it out)?
bool ks_recombine_takes_tail(const fhesi_ctx* ctx, const CrtTables* t, const fhesi_ksk* k) {
  if (!k->aux32 || aux32_row_len(ctx) != 2 * kAux32N) return false;
  if (k->aux_centred) return true;                       // ks_recombine_centred_kernel, any chain; on a linear-convolution ring together with the fold
  if (k->aux_fold || !ctx->pow2) return false;
  return t->W == 34 && k->aux_logQ == 1024 && k->aux_limb_bits == 72 && k->aux_rows == 30;
}
int launch_ks_recombine(fhesi_ctx* ctx, const CrtTables* t, const fhesi_ksk* k, const u64* d_o, i64 npolys, u64* d_out, int nl_out, bool tail_pending) {
  if (!npolys) return 0;
  if (tail_pending != ks_recombine_takes_tail(ctx, t, k) && tail_pending) FHESI_FAIL("key switch: rows without their tail stage reached a recombination that does not take it");
  if (k->aux_centred) { ProfScope prof(ctx, PROF_CRT, (double)npolys); return launch_ks_recombine_centred(ctx, k, d_o, npolys, d_out, nl_out, tail_pending); }
  if (tail_pending) { ProfScope prof(ctx, PROF_CRT, (double)npolys); return launch_ks_recombine_t<34, 1024, 72, 30, true, 1>(ctx, t, k, d_o, npolys, d_out, nl_out); }
  ProfScope prof(ctx, PROF_CRT, (double)npolys);
  // compile-time instantiations for the shapes the benchmarks run (the plan of ks_limb_plan at the metric and stress chains) ...
  if (!k->aux_fold && k->aux32 && t->W == 18 && k->aux_logQ == 512 && k->aux_limb_bits == 74 && k->aux_rows == 15) return launch_ks_recombine_t<18, 512, 74, 15, true>(ctx, t, k, d_o, npolys, d_out, nl_out);
  if (!k->aux32 && t->W == 18 && k->aux_logQ == 512 && k->aux_limb_bits == 74 && k->aux_rows == 15) return launch_ks_recombine_t<18, 512, 74, 15, false>(ctx, t, k, d_o, npolys, d_out, nl_out);
  if (!k->aux32 && t->W == 34 && k->aux_logQ == 1024 && k->aux_limb_bits == 72 && k->aux_rows == 30) return launch_ks_recombine_t<34, 1024, 72, 30, false>(ctx, t, k, d_o, npolys, d_out, nl_out);
  if (!k->aux_fold && k->aux32 && t->W == 34 && k->aux_logQ == 1024 && k->aux_limb_bits == 72 && k->aux_rows == 30) return launch_ks_recombine_t<34, 1024, 72, 30, true>(ctx, t, k, d_o, npolys, d_out, nl_out);
  // ... and the run-time form for every other chain
  const u64 q0 = ctx->q[0], q1 = ctx->q[1];
  const u64 inv = hm::invmod(q0 % q1, q1);
  u128 A = (u128)q0 * q1;
  Garner32 gc{};
  if (k->aux32) {
    if (garner32_consts(ctx, &gc)) return 1;
    A = (u128)((u64)gc.p[0] * gc.p[1]) * ((u64)gc.p[2] * gc.p[3]);
  }
  const u128 half = (A - 1) / 2;
  const int W = t->W;
  dim3 grid((unsigned)((ctx->phim + 127) / 128), (unsigned)npolys);
  const i64 nrow = k->aux32 ? aux32_row_len(ctx) : ctx->phim;
#define KS_GEN(MAXW, A32) ks_recombine_generic_kernel<MAXW, A32><<<grid, 128, 0, ctx->stream>>>(d_o, ctx->phim, nrow, k->aux_fold, W, k->aux_logQ, k->aux_limb_bits, k->aux_rows, q0, q1, inv, hm::shoup(inv, q1), \
      (u64)(half >> 64), (u64)half, (u64)(A >> 64), (u64)A, k->d_limb_consts, t->d_P + (size_t)t->nidx * t->W, t->d_halfP, d_out, nl_out, gc)
  if (W <= 20) { if (k->aux32) KS_GEN(20, true); else KS_GEN(20, false); }
  else if (W <= 44) { if (k->aux32) KS_GEN(44, true); else KS_GEN(44, false); }
  else FHESI_FAIL("key switch, limb mode: chain product of %d limbs exceeds the supported 44", W);
#undef KS_GEN
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_crt(fhesi_ctx* ctx, const CrtTables* t, const u64* d_rows, int nslots_layout, const int* d_slot_of, i64 npolys, int mode, int positive,
               int logQ, u64* d_out, int nl_out) {
  if (!npolys) return 0;
  if (mode != 0 && (logQ < 1 || logQ > 64 * (t->W - 1))) FHESI_FAIL("CRT: logQ=%d outside the range covered by the prime set", logQ);
  const int W = t->W;
  ProfScope prof(ctx, PROF_CRT, (double)npolys);
  // fully unrolled instantiation for the metric chain shape (fhe-si logQ = 512: 18 primes, 17-limb product)
  // sum form for the chain shapes of the metric ring (logQ = 512: 18 primes, 17-limb product) and of the stress ring (logQ = 1024: 35 primes,
  // 33-limb product); FHESI_CRT_EXACT=1 keeps the mixed-radix kernel (A/B measurements)
  if (mode != 0 && !positive && !ctx->opt.crt_exact) {
    if (t->nidx == 18 && W == 18 && logQ == 512) return launch_crt_sum<18, 18, 512>(ctx, t, d_rows, nslots_layout, d_slot_of, npolys, mode, d_out, nl_out);
    if (t->nidx == 35 && W == 34 && logQ == 1024) return launch_crt_sum<35, 34, 1024>(ctx, t, d_rows, nslots_layout, d_slot_of, npolys, mode, d_out, nl_out);
  }
  if (t->nidx == 18 && W == 18 && logQ == 512 && mode != 0) return launch_crt_t<18, 18, 18, 512>(ctx, t, d_rows, nslots_layout, d_slot_of, npolys, mode, positive, logQ, d_out, nl_out);
  if (t->nidx == 18 && W == 18) return launch_crt_t<18, 18, 18>(ctx, t, d_rows, nslots_layout, d_slot_of, npolys, mode, positive, logQ, d_out, nl_out);
  if (W <= 4) return launch_crt_t<4>(ctx, t, d_rows, nslots_layout, d_slot_of, npolys, mode, positive, logQ, d_out, nl_out);
  if (W <= 8) return launch_crt_t<8>(ctx, t, d_rows, nslots_layout, d_slot_of, npolys, mode, positive, logQ, d_out, nl_out);
  if (W <= 12) return launch_crt_t<12>(ctx, t, d_rows, nslots_layout, d_slot_of, npolys, mode, positive, logQ, d_out, nl_out);
  if (W <= 20) return launch_crt_t<20>(ctx, t, d_rows, nslots_layout, d_slot_of, npolys, mode, positive, logQ, d_out, nl_out);
  if (W <= 28) return launch_crt_t<28>(ctx, t, d_rows, nslots_layout, d_slot_of, npolys, mode, positive, logQ, d_out, nl_out);
  if (W <= 44) return launch_crt_t<44>(ctx, t, d_rows, nslots_layout, d_slot_of, npolys, mode, positive, logQ, d_out, nl_out);
  FHESI_FAIL("CRT: prime-set product of %d limbs exceeds the supported 44", W);
}

// ----------------------------------------------------------------------------------------- modulus switching
// DoubleCRT::scaleDownToSet (DoubleCRT.cpp:531-545): delta = toPoly over the dropped primes (centred modulo D = their product), then
//   delta[i] = delta[i] * factor - delta[i],  factor = D * (D^-1 mod p),  reduced to the centred residue modulo D p (Util.cpp:35-43).
// factor - 1 is -1 modulo D and 0 modulo p, so the result e is the one centred value with e = -delta (mod D), e = 0 (mod p):
//   t0 = (delta mod p) * u mod p  (u = D^-1 mod p),  e0 = D t0 - delta in (-D/2, D p - D/2],  + D p when negative,  - D p when above
//   floor(D p / 2) -- exact integer arithmetic on W limbs per coefficient, the same bits as the reference's big-integer loop.
// consts: D[W], M = D p [W], floor(M / 2) [W].  delta / e: [n][W] two's complement, coefficient-major.
template <int MAXW>
__global__ void __launch_bounds__(128) modswitch_delta_kernel(const u64* __restrict__ delta, i64 n, int W, const u64* __restrict__ consts, u64 p, u64 u, u64* __restrict__ e) {
  const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  u64 d[MAXW], x[MAXW];
  for (int i = 0; i < W; ++i) d[i] = delta[j * W + i];
  const bool neg = (d[W - 1] >> 63) != 0;
  // |delta| mod p by Horner, then the sign
  u64 r = 0;
  {
    u64 c = 1;
    for (int i = 0; i < W; ++i) { const u64 v = neg ? ~d[i] + c : d[i]; c = (neg && c && v == 0) ? 1 : 0; x[i] = v; }
    for (int i = W - 1; i >= 0; --i) r = (u64)((((u128)r << 64) | x[i]) % p);
    if (neg && r) r = p - r;
  }
  const u64 t0 = (u64)(((u128)r * u) % p);
  // x = D t0 - delta
  const u64* D = consts; const u64* M = consts + W; const u64* H = consts + 2 * W;
  {
    u64 c = 0, br = 0;
    for (int i = 0; i < W; ++i) {
      const u128 t = (u128)D[i] * t0 + c;
      c = (u64)(t >> 64);
      const u64 lo = (u64)t;
      const u128 s = (u128)lo - d[i] - br;
      x[i] = (u64)s; br = (u64)(s >> 64) & 1;
    }
  }
  if (x[W - 1] >> 63) { u64 c = 0; for (int i = 0; i < W; ++i) { const u128 s = (u128)x[i] + M[i] + c; x[i] = (u64)s; c = (u64)(s >> 64); } }
  // x in [0, M): subtract M when x > floor(M / 2)
  int cmp = 0;
  for (int i = W - 1; i >= 0 && !cmp; --i) if (x[i] != H[i]) cmp = x[i] < H[i] ? -1 : 1;
  if (cmp > 0) { u64 br = 0; for (int i = 0; i < W; ++i) { const u128 s = (u128)x[i] - M[i] - br; x[i] = (u64)s; br = (u64)(s >> 64) & 1; } }
  for (int i = 0; i < W; ++i) e[j * W + i] = x[i];
}

// d_delta, d_e: [n][W]; consts_host: 3 W words (D, D p, floor(D p / 2)), all below 2^(64 W - 1)
int launch_modswitch_delta(fhesi_ctx* ctx, const u64* d_delta, int W, const u64* consts_host, u64 p, u64 u, u64* d_e) {
  void* d_c;
  FHESI_TRY(ws_reserve(ctx, 9, (size_t)3 * W * 8 + 64, &d_c));
  HIP_TRY(hipMemcpyAsync(d_c, consts_host, (size_t)3 * W * 8, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));            // the caller's constant array may go away
  const unsigned grid = (unsigned)((ctx->phim + 127) / 128);
  if (W <= 8) modswitch_delta_kernel<8><<<grid, 128, 0, ctx->stream>>>(d_delta, ctx->phim, W, (const u64*)d_c, p, u, d_e);
  else if (W <= 24) modswitch_delta_kernel<24><<<grid, 128, 0, ctx->stream>>>(d_delta, ctx->phim, W, (const u64*)d_c, p, u, d_e);
  else if (W <= 66) modswitch_delta_kernel<66><<<grid, 128, 0, ctx->stream>>>(d_delta, ctx->phim, W, (const u64*)d_c, p, u, d_e);
  else FHESI_FAIL("scaleDownToSet: dropped-prime product of %d limbs is too wide", W);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ----------------------------------------------------------------------------------------- byte decomposition -> digit rows
// parts: positive residues mod 2^logQ, limb-major [npolys][nl][n] (mode-1 output of crt_kernel).
// rows out: [npolys][nd][L][n], digit d of coefficient j reduced mod q_l (digits < 2^digit_bits; Ciphertext.cpp:95-102,
// zero digits simply give zero coefficients).
__global__ void __launch_bounds__(256) digits_kernel(const u64* __restrict__ parts, int nl, i64 n, int logQ, int digit_bits, int nd, int L,
                                                      u64* __restrict__ rows, const PrimeConst* __restrict__ pcs, u64 only_below_q) {
  const i64 poly = blockIdx.z;
  const int d = blockIdx.y;
  const int lo = d * digit_bits, w = lo >> 6, b = lo & 63;
  const u64 mask = (digit_bits >= 64) ? ~0ull : ((1ull << digit_bits) - 1);
  const u64* src = parts + poly * nl * n;
  for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x) {
    u64 v = (w < nl) ? (src[(i64)w * n + j] >> b) : 0;
    if (b + digit_bits > 64 && w + 1 < nl) v |= src[(i64)(w + 1) * n + j] << (64 - b);
    v &= mask;
    for (int l = 0; l < L; ++l) {
      const u64 q = pcs[l].q;
      if (only_below_q && q >= only_below_q) continue;        // rows of the other primes are produced by the fused tile kernel
      rows[((poly * nd + d) * L + l) * n + j] = v < q ? v : v % q;
    }
  }
}

int launch_digits(fhesi_ctx* ctx, const u64* d_parts, int nl, int logQ, int digit_bits, int nd, i64 npolys, u64* d_rows, u64 only_below_q) {
  if (!npolys) return 0;
  ProfScope prof(ctx, PROF_DIGITS, (double)npolys);
  unsigned gx = (unsigned)((ctx->phim + 255) / 256);
  if (gx > 64) gx = 64;
  dim3 grid(gx, (unsigned)nd, (unsigned)npolys);
  digits_kernel<<<grid, 256, 0, ctx->stream>>>(d_parts, nl, ctx->phim, logQ, digit_bits, nd, ctx->L, d_rows, ctx->d_pc, only_below_q);
  HIP_TRY(hipGetLastError());
  return 0;
}
