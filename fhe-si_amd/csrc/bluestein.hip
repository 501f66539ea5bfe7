// bluestein.hip -- Cmod::FFT / iFFT for general (non power-of-two) m on the GPU.
//
// Restates tBluesteinFFT (bluestein.cpp:93-144) and its callers (CModulus.cpp:90-132):
//   x[k] = powers[k] * sum_i (a[i] powers[i]) * b[m-1+k-i],   powers[i] = root^(i^2 mod 2m),  b[m-1+-i] = root^-(i^2)
// The reference evaluates the inner product as an N-point cyclic convolution (N = 2^ceil(log2(2m-1))) inside NTL's
// fftRep, i.e. over NTL's own auxiliary FFT primes followed by a CRT (bluestein.cpp:116-119,138-139).  The chain primes
// only satisfy q = 1 mod 2m, so they have no N-th roots of unity either; this file does the same thing explicitly:
// the convolution is computed exactly over the integers modulo three fixed NTT-friendly auxiliary primes
// P0 P1 P2 > m q^2 (reusing the power-of-two NTT kernels of kernels_ntt.hip through an internal context), recombined
// by Garner's CRT and reduced mod q.  A negacyclic wrap of size N leaves the output window m-1..2m-2 untouched
// (wrapped indices are < m-1), exactly as the cyclic wrap of the reference does.
// iFFT additionally scatters to Z_m^* (CModulus.cpp:117-121), multiplies by m^-1 (:125) and reduces modulo Phi_m (:128-129).
#include "../../include/fhesi_hip.h"
#include "fhesi_internal.h"

#include <algorithm>

extern thread_local bool g_fhesi_internal_ctx;

struct BluesteinTables {
  fhesi_ctx* aux = nullptr;       // internal power-of-two context of size N over the 3 auxiliary primes
  i64 N = 0;
  u64 P[3] = {0, 0, 0};
  Shoup2* d_pow = nullptr;        // [L][2][m]   powers (dir 0) / ipowers (dir 1), Shoup pairs mod q_i
  u64* d_bhat = nullptr;          // [L][2][3][N] transform of the chirp b modulo each auxiliary prime (natural order)
  u64* d_crt = nullptr;           // [L][4] : P0 mod q, P0*P1 mod q, m^-1 mod q, unused
  u64* d_garner = nullptr;        // [2] : P0^-1 mod P1, (P0 P1)^-1 mod P2
  i64* d_phi = nullptr;           // [phim+1] Phi_m coefficients
  int phi_kind = 0;               // 0 generic (long division in LDS, or two convolutions: phi_conv), 1 m prime, 2 m = 2 * odd prime
  bool phi_conv = false;          // generic m above 16384 (or FHESI_PHI_CONV=1): rem(f, Phi_m) = f - Phi_m * top(f * Psi_m), see blue_conv_*
  u64* d_khat = nullptr;          // [L][2][3][N] transforms of Psi_m (dir 0) and Phi_m (dir 1), coefficients taken modulo q_i first
  // N = 2^15 .. 2^17: the auxiliary transforms run order-free (head stages + in-place sub-transforms; sub-transforms + tail
  // stages), the chirp transform is stored in that order.  N = 2^15 additionally fuses the single head stage into blue_pre
  // (its partner words are zero: m <= N/2) and the single tail stage into blue_post.
  bool orderfree = false, fused = false;
};

// ------------------------------------------------------------------------------------------------ kernels
// a -> X_j[k] = (a[k] * powers[k] mod q) mod P_j  for k < m, 0 above.  grid: (ceil(N/256), rows)
// DUP: the first Cooley-Tukey stage of the size-N transform pairs word k with word k + N/2, which is zero (m <= N/2), so both
// outputs equal word k: write it twice and let the transform start at stage 1.  grid.x then covers N/2 words.
template <bool INV, bool DUP = false>
__global__ void __launch_bounds__(256) blue_pre(const u64* __restrict__ rows, i64 phim, i64 m, i64 N, int nslots, const int* __restrict__ prime_of_slot,
                                                const PrimeConst* __restrict__ pcs, const Shoup2* __restrict__ pow_all, const int* __restrict__ zms_idx,
                                                u64 P0, u64 P1, u64 P2, u64* __restrict__ X) {
  const i64 r = blockIdx.y;
  const int slot = (int)(r % nslots);
  const int prime = prime_of_slot ? prime_of_slot[slot] : slot;
  const u64 q = pcs[prime].q;
  const Shoup2* pw = pow_all + ((i64)prime * 2 + (INV ? 1 : 0)) * m;
  const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= (DUP ? N / 2 : N)) return;
  u64 t = 0;
  if (k < m) {
    u64 a = 0;
    if (!INV) { if (k < phim) a = rows[r * phim + k]; }
    else { const int z = zms_idx[k]; if (z >= 0) a = rows[r * phim + z]; }
    t = d_shoup(a, pw[k].w, pw[k].wp, q);
  }
  u64* x = X + r * 3 * N;
  const u64 t0 = t >= P0 ? t % P0 : t, t1 = t >= P1 ? t % P1 : t, t2 = t >= P2 ? t % P2 : t;
  x[k] = t0;
  x[N + k] = t1;
  x[2 * N + k] = t2;
  if (DUP) {
    x[k + N / 2] = t0;
    x[N + k + N / 2] = t1;
    x[2 * N + k + N / 2] = t2;
  }
}

// X[r][j][:] *= bhat[prime][dir][j][:]  mod P_j   (aux PrimeConst table: index j)
__global__ void __launch_bounds__(256) blue_mul(u64* __restrict__ X, i64 N, int nslots, const int* __restrict__ prime_of_slot, int dir,
                                                const u64* __restrict__ bhat, const PrimeConst* __restrict__ aux_pcs) {
  const i64 r = blockIdx.y / 3;
  const int j = blockIdx.y % 3;
  const int slot = (int)(r % nslots);
  const int prime = prime_of_slot ? prime_of_slot[slot] : slot;
  const PrimeConst pc = aux_pcs[j];
  u64* x = X + (r * 3 + j) * N;
  const u64* b = bhat + (((i64)prime * 2 + dir) * 3 + j) * N;
  for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < N; k += (i64)gridDim.x * blockDim.x) x[k] = d_mulmod(x[k], b[k], pc);
}

// window m-1..2m-2 of the convolution: Garner CRT -> mod q -> * powers[k]; FFT keeps Z_m^*, iFFT multiplies by m^-1.
// FUSED (N = 2^15): X holds the output of the inverse sub-transforms; the last Gentleman-Sande stage (partner distance N/2,
// twiddle aux_tw_inv[1]) is applied here to the one word of each pair that the window needs.
template <bool INV, bool FUSED = false>
__global__ void __launch_bounds__(256) blue_post(const u64* __restrict__ X, i64 phim, i64 m, i64 N, int nslots, const int* __restrict__ prime_of_slot,
                                                 const PrimeConst* __restrict__ pcs, const PrimeConst* __restrict__ aux_pcs, const Shoup2* __restrict__ pow_all,
                                                 const int* __restrict__ zms_idx, const u64* __restrict__ crt, const u64* __restrict__ garner,
                                                 u64* __restrict__ out /* FFT: rows [R][phim]; iFFT: full [R][m] */, const Shoup2* __restrict__ aux_tw_inv) {
  const i64 r = blockIdx.y;
  const int slot = (int)(r % nslots);
  const int prime = prime_of_slot ? prime_of_slot[slot] : slot;
  const PrimeConst pc = pcs[prime];
  const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= m) return;
  const int z = zms_idx[k];
  if (!INV && z < 0) return;
  const PrimeConst a1 = aux_pcs[1], a2 = aux_pcs[2];
  u64 c0, c1, c2;
  if (!FUSED) {
    const u64* x = X + r * 3 * N + (m - 1 + k);
    c0 = x[0]; c1 = x[N]; c2 = x[2 * N];
  } else {
    const i64 w = m - 1 + k, h = N / 2;
    const u64* x = X + r * 3 * N + (w < h ? w : w - h);
    const PrimeConst a0 = aux_pcs[0];
    if (w < h) {        // X' = X + Y
      c0 = d_addmod(x[0], x[h], a0.q); c1 = d_addmod(x[N], x[N + h], a1.q); c2 = d_addmod(x[2 * N], x[2 * N + h], a2.q);
    } else {            // Y' = (X - Y) * psi^-brv(1)
      c0 = d_mulmod(d_submod(x[0], x[h], a0.q), aux_tw_inv[1].w, a0);
      c1 = d_mulmod(d_submod(x[N], x[N + h], a1.q), aux_tw_inv[N + 1].w, a1);
      c2 = d_mulmod(d_submod(x[2 * N], x[2 * N + h], a2.q), aux_tw_inv[2 * N + 1].w, a2);
    }
  }
  // mixed radix digits: value = v0 + v1 P0 + v2 P0 P1
  const u64 v0 = c0;
  const u64 v1 = d_mulmod(d_submod(c1, v0 >= a1.q ? v0 % a1.q : v0, a1.q), garner[0], a1);
  const u64 v0m2 = v0 >= a2.q ? v0 % a2.q : v0, v1m2 = v1 >= a2.q ? v1 % a2.q : v1;
  // (c2 - v0 - v1 P0) * (P0 P1)^-1 mod P2
  const u64 p0m2 = aux_pcs[0].q % a2.q;
  const u64 t2 = d_submod(d_submod(c2, v0m2, a2.q), d_mulmod(v1m2, p0m2, a2), a2.q);
  const u64 v2 = d_mulmod(t2, garner[1], a2);
  // reduce modulo the chain prime
  const u64 q = pc.q;
  const u64 r0 = v0 % q, r1 = v1 % q, r2 = v2 % q;
  u64 val = d_addmod(r0, d_addmod(d_mulmod(r1, crt[prime * 4 + 0], pc), d_mulmod(r2, crt[prime * 4 + 1], pc), q), q);
  const Shoup2 pw = pow_all[((i64)prime * 2 + (INV ? 1 : 0)) * m + k];
  val = d_shoup(val, pw.w, pw.wp, q);
  if (!INV) out[r * phim + z] = val;
  else out[r * m + k] = d_mulmod(val, crt[prime * 4 + 2], pc);
}

// rem(out, Phi_m) (CModulus.cpp:128-129) for the three shapes of m.  f: [R][m] -> rows [R][phim]
// kind 1: m prime, Phi = 1 + X + ... + X^(m-1);  kind 2: m = 2p', Phi(X) = sum (-X)^j, X^p' = -1
__global__ void __launch_bounds__(256) blue_phi_fast(const u64* __restrict__ f, i64 phim, i64 m, int kind, int nslots, const int* __restrict__ prime_of_slot,
                                                     const PrimeConst* __restrict__ pcs, u64* __restrict__ rows) {
  const i64 r = blockIdx.y;
  const int slot = (int)(r % nslots);
  const u64 q = pcs[prime_of_slot ? prime_of_slot[slot] : slot].q;
  const u64* fr = f + r * m;
  const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= phim) return;
  if (kind == 1) rows[r * phim + j] = d_submod(fr[j], fr[m - 1], q);
  else {
    const i64 pp = m / 2;                                   // phim = pp - 1
    const u64 top = d_submod(fr[pp - 1], fr[2 * pp - 1], q);       // coefficient of X^(pp-1) after folding X^pp = -1
    const u64 g = d_submod(fr[j], fr[j + pp], q);
    rows[r * phim + j] = (j & 1) ? d_addmod(g, top, q) : d_submod(g, top, q);   // Phi_m has coefficient (-1)^j at X^j
  }
}
// generic monic long division, one block per row, polynomial held in LDS (m <= 16384)
__global__ void __launch_bounds__(256) blue_phi_generic(const u64* __restrict__ f, i64 phim, i64 m, const i64* __restrict__ phi, int nslots,
                                                        const int* __restrict__ prime_of_slot, const PrimeConst* __restrict__ pcs, u64* __restrict__ rows) {
  extern __shared__ __attribute__((aligned(16))) u64 s[];
  const i64 r = blockIdx.x;
  const int slot = (int)(r % nslots);
  const PrimeConst pc = pcs[prime_of_slot ? prime_of_slot[slot] : slot];
  const u64 q = pc.q;
  for (i64 i = threadIdx.x; i < m; i += blockDim.x) s[i] = f[r * m + i];
  __syncthreads();
  for (i64 k = m - 1; k >= phim; --k) {
    const u64 c = s[k];
    __syncthreads();
    if (c) {
      for (i64 j = threadIdx.x; j < phim; j += blockDim.x) {
        const i64 co = phi[j];
        if (co) {
          const u64 cm = co > 0 ? (u64)co % q : (q - ((u64)(-co) % q)) % q;
          s[k - phim + j] = d_submod(s[k - phim + j], d_mulmod(c, cm, pc), q);
        }
      }
    }
    __syncthreads();
  }
  for (i64 i = threadIdx.x; i < phim; i += blockDim.x) rows[r * phim + i] = s[i];
}

// ---- rem(f, Phi_m) for general m by two exact convolutions (CModulus.cpp:128-129 calls NTL's rem, itself FFT-based)
// f = Q Phi + r, deg r < phi(m), deg Q < t = m - phi(m).  With Psi = (X^m - 1) / Phi (integer coefficients, hm::cyclotomic_cofactor):
//   f Psi = Q (X^m - 1) + r Psi,  deg(r Psi) < m   =>   Q = the coefficients m .. m + t - 1 of f Psi      (deg(f Psi) <= 2m - 2 < N: no wrap)
//   r = (f - Q Phi) mod X^phi(m)                                                                          (deg(Q Phi) < m)
// Both products run like the chirp convolution: modulo the three auxiliary primes, Garner, mod q; the kernels' coefficients are reduced
// modulo q first, so every integer is non-negative and below m q^2.  The long division in LDS needs the whole polynomial in 128 KB and
// m - phi(m) dependent steps; this form has neither limit (any m below 2^20, FHEContext.cpp:89).
// X_j[i] = src[r][i] mod P_j for i < len, 0 above.  grid: (ceil(N/256), rows)
__global__ void __launch_bounds__(256) blue_conv_pre(const u64* __restrict__ src, i64 stride, i64 len, i64 N, u64 P0, u64 P1, u64 P2, u64* __restrict__ X) {
  const i64 r = blockIdx.y;
  const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= N) return;
  const u64 t = k < len ? src[r * stride + k] : 0;
  u64* x = X + r * 3 * N;
  x[k] = t >= P0 ? t % P0 : t;
  x[N + k] = t >= P1 ? t % P1 : t;
  x[2 * N + k] = t >= P2 ? t % P2 : t;
}
// the integer behind (c0, c1, c2) modulo the chain prime q (Garner's mixed radix digits, as blue_post)
static __device__ __forceinline__ u64 blue_garner_mod_q(u64 c0, u64 c1, u64 c2, const PrimeConst* __restrict__ aux_pcs, const u64* __restrict__ garner,
                                                        const u64* __restrict__ crt, int prime, const PrimeConst& pc) {
  const PrimeConst a1 = aux_pcs[1], a2 = aux_pcs[2];
  const u64 v0 = c0;
  const u64 v1 = d_mulmod(d_submod(c1, v0 >= a1.q ? v0 % a1.q : v0, a1.q), garner[0], a1);
  const u64 v0m2 = v0 >= a2.q ? v0 % a2.q : v0, v1m2 = v1 >= a2.q ? v1 % a2.q : v1;
  const u64 p0m2 = aux_pcs[0].q % a2.q;
  const u64 t2 = d_submod(d_submod(c2, v0m2, a2.q), d_mulmod(v1m2, p0m2, a2), a2.q);
  const u64 v2 = d_mulmod(t2, garner[1], a2);
  const u64 q = pc.q;
  const u64 r0 = v0 % q, r1 = v1 % q, r2 = v2 % q;
  return d_addmod(r0, d_addmod(d_mulmod(r1, crt[prime * 4 + 0], pc), d_mulmod(r2, crt[prime * 4 + 1], pc), q), q);
}
// out[r][j] = (sub ? sub[r][j] - v : v) mod q with v = coefficient first + j of the product in X, j < len.  grid: (ceil(len/256), rows)
__global__ void __launch_bounds__(256) blue_conv_post(const u64* __restrict__ X, i64 N, i64 first, i64 len, int nslots, const int* __restrict__ prime_of_slot,
                                                      const PrimeConst* __restrict__ pcs, const PrimeConst* __restrict__ aux_pcs, const u64* __restrict__ crt,
                                                      const u64* __restrict__ garner, const u64* __restrict__ sub, i64 sub_stride, u64* __restrict__ out, i64 out_stride) {
  const i64 r = blockIdx.y;
  const int slot = (int)(r % nslots);
  const int prime = prime_of_slot ? prime_of_slot[slot] : slot;
  const PrimeConst pc = pcs[prime];
  const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= len) return;
  const u64* x = X + r * 3 * N + first + j;
  const u64 v = blue_garner_mod_q(x[0], x[N], x[2 * N], aux_pcs, garner, crt, prime, pc);
  out[r * out_stride + j] = sub ? d_submod(sub[r * sub_stride + j], v, pc.q) : v;
}

// ------------------------------------------------------------------------------------------------ setup
static u64 root_of_two_power_order(u64 p, int k) {       // element of exact order 2^k modulo prime p (2^k | p-1)
  for (u64 h = 2;; ++h) {
    const u64 x = hm::powmod(h, (p - 1) >> k, p);
    if (hm::powmod(x, 1ull << (k - 1), p) != 1) return x;
  }
}

int bluestein_init(fhesi_ctx* c) {
  const i64 m = c->m;
  const int L = c->L;
  BluesteinTables* B = new BluesteinTables();
  c->blue = B;
  int k = 0;
  while ((1ll << k) < 2 * m - 1) ++k;                    // NextPowerOfTwo(2m-1), bluestein.cpp:116
  if (k < 2) k = 2;
  B->N = 1ll << k;
  const i64 N = B->N;
  // three auxiliary primes = 1 mod 4N, descending from 2^60 (product ~2^180 > m q^2 for every q < 2^60, m <= 2^20)
  {
    const u64 step = 4 * (u64)N;
    u64 p = (1ull << 60) - ((1ull << 60) % step) + 1;
    int found = 0;
    while (found < 3) {
      p -= step;
      if (hm::is_prime(p) && std::find(c->q.begin(), c->q.end(), p) == c->q.end()) B->P[found++] = p;
    }
  }
  u64 aroot[3];
  for (int j = 0; j < 3; ++j) aroot[j] = root_of_two_power_order(B->P[j], k + 2);     // primitive 2*(2N)-th root
  g_fhesi_internal_ctx = true;
  const int rc = fhesi_ctx_create(&B->aux, 2 * N, 3, (const uint64_t*)B->P, (const uint64_t*)aroot, c->device);
  g_fhesi_internal_ctx = false;
  if (rc) return rc;

  // per chain prime: powers / ipowers (bluestein.cpp:103-109) and the chirp b (:121-133) modulo each auxiliary prime
  std::vector<Shoup2> pw((size_t)L * 2 * m);
  std::vector<u64> bh((size_t)L * 2 * 3 * N, 0), crt((size_t)L * 4, 0);
  for (int i = 0; i < L; ++i) {
    const u64 q = c->q[i], root = c->root[i], rinv = hm::invmod(root, q);
    std::vector<u64> p0(m), p1(m);
    // root^(j^2) by the recurrence (j + 1)^2 = j^2 + (2j + 1): two products per entry where a power per entry took ~30 (12 s of the
    // context's creation at m = 2^19, 50 s at 2^20); valid when root^(2m) = 1, which is what reducing the exponent modulo 2m assumed
    const bool rec = hm::powmod(root, 2 * (u64)m, q) == 1;
    u64 g0 = root, g1 = rinv;                                // root^(2j + 1), rinv^(2j + 1)
    const u64 r2 = hm::mulmod(root, root, q), ri2 = hm::mulmod(rinv, rinv, q);
    for (i64 j = 0; j < m; ++j) {
      if (!rec) {
        const u64 e = (u64)(((u128)j * j) % (2 * (u64)m));
        p0[j] = hm::powmod(root, e, q);
        p1[j] = hm::powmod(rinv, e, q);
      } else if (j == 0) p0[0] = p1[0] = 1 % q;
      else {
        p0[j] = hm::mulmod(p0[j - 1], g0, q); g0 = hm::mulmod(g0, r2, q);
        p1[j] = hm::mulmod(p1[j - 1], g1, q); g1 = hm::mulmod(g1, ri2, q);
      }
      pw[((size_t)i * 2 + 0) * m + j] = {p0[j], hm::shoup(p0[j], q)};
      pw[((size_t)i * 2 + 1) * m + j] = {p1[j], hm::shoup(p1[j], q)};
    }
    for (int dir = 0; dir < 2; ++dir) {
      const std::vector<u64>& neg = dir == 0 ? p1 : p0;      // forward chirp uses root^-(i^2); the inverse direction swaps roles
      for (int a = 0; a < 3; ++a) {
        u64* b = bh.data() + (((size_t)i * 2 + dir) * 3 + a) * N;
        for (i64 j = 0; j < m; ++j) b[m - 1 + j] = b[m - 1 - j] = neg[j] % B->P[a];
      }
    }
    crt[i * 4 + 0] = B->P[0] % q;
    crt[i * 4 + 1] = hm::mulmod(B->P[0] % q, B->P[1] % q, q);
    crt[i * 4 + 2] = hm::invmod((u64)m % q, q);
  }
  u64 garner[2];
  garner[0] = hm::invmod(B->P[0] % B->P[1], B->P[1]);
  garner[1] = hm::invmod(hm::mulmod(B->P[0] % B->P[2], B->P[1] % B->P[2], B->P[2]), B->P[2]);
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMalloc(&B->d_pow, pw.size() * sizeof(Shoup2)));
  HIP_TRY(hipMemcpy(B->d_pow, pw.data(), pw.size() * sizeof(Shoup2), hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&B->d_bhat, bh.size() * 8));
  HIP_TRY(hipMemcpy(B->d_bhat, bh.data(), bh.size() * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&B->d_crt, crt.size() * 8));
  HIP_TRY(hipMemcpy(B->d_crt, crt.data(), crt.size() * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&B->d_garner, 16));
  HIP_TRY(hipMemcpy(B->d_garner, garner, 16, hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&B->d_phi, c->phi.size() * 8));
  HIP_TRY(hipMemcpy(B->d_phi, c->phi.data(), c->phi.size() * 8, hipMemcpyHostToDevice));
  // transform the chirps once (role of the cached Rb, bluestein.cpp:121-136): layout [L*2][3][N] = 2L "DoubleCRTs" of the aux context
  B->orderfree = ntt_orderfree_two_pass(B->aux);
  B->fused = B->orderfree && k == 15;
  FHESI_TRY(launch_ntt_fwd(B->aux, B->d_bhat, (i64)L * 2, 3, nullptr, !B->orderfree));
  HIP_TRY(hipStreamSynchronize(B->aux->stream));
  // shape of m for the reduction modulo Phi_m
  if (hm::is_prime((u64)m)) B->phi_kind = 1;
  else if (m % 2 == 0 && (m / 2) % 2 == 1 && hm::is_prime((u64)(m / 2))) B->phi_kind = 2;
  else B->phi_kind = 0;
  // generic m: the long division in LDS up to m = 16384, two convolutions above (FHESI_PHI_CONV=1, a test hook read here, asks for them at
  // any size: the rings small enough for the oracle and for the long division to stand beside it)
  if (B->phi_kind == 0) { const char* e = getenv("FHESI_PHI_CONV"); B->phi_conv = m > 16384 || (e && atoi(e) != 0); }
  if (B->phi_conv) {
    const std::vector<i64> psi = hm::cyclotomic_cofactor(m);
    const i64 t = m - c->phim;
    if ((i64)psi.size() != t + 1 || (i64)c->phi.size() != c->phim + 1) FHESI_FAIL("Phi_m / Psi_m: unexpected degrees");
    {   // Phi_m(x) Psi_m(x) = x^m - 1 at two points modulo the first chain prime (a wrong table would only show as wrong coefficients)
      const u64 q = c->q[0];
      for (u64 x : {(u64)3, (u64)0x9e3779b97f4a7c15ull % q}) {
        auto eval = [&](const std::vector<i64>& f) { u64 v = 0; for (size_t i = f.size(); i-- > 0;) { const i64 co = f[i]; const u64 cm = co >= 0 ? (u64)co % q : (q - (u64)(-co) % q) % q; v = (hm::mulmod(v, x, q) + cm) % q; } return v; };
        if (hm::mulmod(eval(c->phi), eval(psi), q) != (hm::powmod(x, (u64)m, q) + q - 1) % q) FHESI_FAIL("Phi_m * Psi_m != X^m - 1 for m = %lld", (long long)m);
      }
    }
    std::vector<u64> kh((size_t)L * 2 * 3 * N, 0);
    for (int i = 0; i < L; ++i) {
      const u64 q = c->q[i];
      for (int dir = 0; dir < 2; ++dir) {
        const std::vector<i64>& f = dir == 0 ? psi : c->phi;
        for (size_t j = 0; j < f.size(); ++j) {
          const i64 co = f[j];
          if (!co) continue;
          const u64 cm = co > 0 ? (u64)co % q : (q - ((u64)(-co) % q)) % q;
          for (int a = 0; a < 3; ++a) kh[(((size_t)i * 2 + dir) * 3 + a) * N + j] = cm % B->P[a];
        }
      }
    }
    HIP_TRY(hipMalloc(&B->d_khat, kh.size() * 8));
    HIP_TRY(hipMemcpy(B->d_khat, kh.data(), kh.size() * 8, hipMemcpyHostToDevice));
    FHESI_TRY(launch_ntt_fwd(B->aux, B->d_khat, (i64)L * 2, 3, nullptr, !B->orderfree));
    HIP_TRY(hipStreamSynchronize(B->aux->stream));
  }
  return 0;
}

void bluestein_destroy(fhesi_ctx* c) {
  BluesteinTables* B = c->blue;
  if (!B) return;
  if (B->aux) fhesi_ctx_destroy(B->aux);
  hipFree(B->d_pow); hipFree(B->d_bhat); hipFree(B->d_khat); hipFree(B->d_crt); hipFree(B->d_garner); hipFree(B->d_phi);
  delete B;
  c->blue = nullptr;
}

// ------------------------------------------------------------------------------------------------ launch
// One chunk of `count` DoubleCRTs (R = count * nslots rows).  Everything is enqueued on the caller's stream: the auxiliary
// convolution context borrows it for the duration of the call, so no host synchronisation is needed between the stages.
static int blue_chunk(fhesi_ctx* c, u64* d_rows, i64 count, int nslots, const int* d_pos, bool inv, u64* X, u64* dF) {
  BluesteinTables* B = c->blue;
  const i64 R = count * nslots, N = B->N, m = c->m, phim = c->phim;
  dim3 gpre((unsigned)(((B->fused ? N / 2 : N) + 255) / 256), (unsigned)R);
  if (B->fused) {
    if (!inv) blue_pre<false, true><<<gpre, 256, 0, c->stream>>>(d_rows, phim, m, N, nslots, d_pos, c->d_pc, B->d_pow, c->d_zms_idx, B->P[0], B->P[1], B->P[2], X);
    else blue_pre<true, true><<<gpre, 256, 0, c->stream>>>(d_rows, phim, m, N, nslots, d_pos, c->d_pc, B->d_pow, c->d_zms_idx, B->P[0], B->P[1], B->P[2], X);
  } else {
    if (!inv) blue_pre<false><<<gpre, 256, 0, c->stream>>>(d_rows, phim, m, N, nslots, d_pos, c->d_pc, B->d_pow, c->d_zms_idx, B->P[0], B->P[1], B->P[2], X);
    else blue_pre<true><<<gpre, 256, 0, c->stream>>>(d_rows, phim, m, N, nslots, d_pos, c->d_pc, B->d_pow, c->d_zms_idx, B->P[0], B->P[1], B->P[2], X);
  }
  HIP_TRY(hipGetLastError());
  if (B->fused) FHESI_TRY(launch_ntt_sub(B->aux, true, X, R, 3, nullptr));
  else FHESI_TRY(launch_ntt_fwd(B->aux, X, R, 3, nullptr, !B->orderfree));
  unsigned gx = (unsigned)((N + 255) / 256);
  if (gx > 64) gx = 64;
  blue_mul<<<dim3(gx, (unsigned)(R * 3)), 256, 0, c->stream>>>(X, N, nslots, d_pos, inv ? 1 : 0, B->d_bhat, B->aux->d_pc);
  HIP_TRY(hipGetLastError());
  if (B->fused) FHESI_TRY(launch_ntt_sub(B->aux, false, X, R, 3, nullptr));
  else FHESI_TRY(launch_ntt_inv(B->aux, X, R, 3, nullptr, !B->orderfree));
  dim3 gpost((unsigned)((m + 255) / 256), (unsigned)R);
  const Shoup2* atw = B->aux->d_tw_inv;
  if (!inv) {
    if (B->fused) blue_post<false, true><<<gpost, 256, 0, c->stream>>>(X, phim, m, N, nslots, d_pos, c->d_pc, B->aux->d_pc, B->d_pow, c->d_zms_idx, B->d_crt, B->d_garner, d_rows, atw);
    else blue_post<false><<<gpost, 256, 0, c->stream>>>(X, phim, m, N, nslots, d_pos, c->d_pc, B->aux->d_pc, B->d_pow, c->d_zms_idx, B->d_crt, B->d_garner, d_rows, atw);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  if (B->fused) blue_post<true, true><<<gpost, 256, 0, c->stream>>>(X, phim, m, N, nslots, d_pos, c->d_pc, B->aux->d_pc, B->d_pow, c->d_zms_idx, B->d_crt, B->d_garner, dF, atw);
  else blue_post<true><<<gpost, 256, 0, c->stream>>>(X, phim, m, N, nslots, d_pos, c->d_pc, B->aux->d_pc, B->d_pow, c->d_zms_idx, B->d_crt, B->d_garner, dF, atw);
  HIP_TRY(hipGetLastError());
  if (B->phi_kind) {
    blue_phi_fast<<<dim3((unsigned)((phim + 255) / 256), (unsigned)R), 256, 0, c->stream>>>((const u64*)dF, phim, m, B->phi_kind, nslots, d_pos, c->d_pc, d_rows);
  } else if (B->phi_conv) {
    const i64 t = m - phim;
    u64* Q = dF + R * m;                       // [R][t], behind the polynomials (blue_run reserves both)
    const unsigned gN = (unsigned)((N + 255) / 256);
    // Q = coefficients m .. m + t - 1 of f Psi
    blue_conv_pre<<<dim3(gN, (unsigned)R), 256, 0, c->stream>>>((const u64*)dF, m, m, N, B->P[0], B->P[1], B->P[2], X);
    FHESI_TRY(launch_ntt_fwd(B->aux, X, R, 3, nullptr, !B->orderfree));
    blue_mul<<<dim3(gx, (unsigned)(R * 3)), 256, 0, c->stream>>>(X, N, nslots, d_pos, 0, B->d_khat, B->aux->d_pc);
    FHESI_TRY(launch_ntt_inv(B->aux, X, R, 3, nullptr, !B->orderfree));
    blue_conv_post<<<dim3((unsigned)((t + 255) / 256), (unsigned)R), 256, 0, c->stream>>>(X, N, m, t, nslots, d_pos, c->d_pc, B->aux->d_pc, B->d_crt, B->d_garner,
                                                                                          nullptr, 0, Q, t);
    // rows = (f - Q Phi) mod X^phi(m)
    blue_conv_pre<<<dim3(gN, (unsigned)R), 256, 0, c->stream>>>((const u64*)Q, t, t, N, B->P[0], B->P[1], B->P[2], X);
    FHESI_TRY(launch_ntt_fwd(B->aux, X, R, 3, nullptr, !B->orderfree));
    blue_mul<<<dim3(gx, (unsigned)(R * 3)), 256, 0, c->stream>>>(X, N, nslots, d_pos, 1, B->d_khat, B->aux->d_pc);
    FHESI_TRY(launch_ntt_inv(B->aux, X, R, 3, nullptr, !B->orderfree));
    blue_conv_post<<<dim3((unsigned)((phim + 255) / 256), (unsigned)R), 256, 0, c->stream>>>(X, N, 0, phim, nslots, d_pos, c->d_pc, B->aux->d_pc, B->d_crt, B->d_garner,
                                                                                             (const u64*)dF, m, d_rows, phim);
  } else {
    const size_t shmem = (size_t)m * 8;
    HIP_TRY(hipFuncSetAttribute((const void*)blue_phi_generic, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    blue_phi_generic<<<(unsigned)R, 256, shmem, c->stream>>>((const u64*)dF, phim, m, B->d_phi, nslots, d_pos, c->d_pc, d_rows);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

static int blue_run(fhesi_ctx* c, u64* d_rows, i64 count, int nslots, const int* prime_of_slot_host, bool inv) {
  BluesteinTables* B = c->blue;
  if (!B) FHESI_FAIL("Bluestein tables missing");
  const i64 N = B->N, m = c->m, phim = c->phim;
  if (!count || !nslots) return 0;
  ProfScope prof(c, inv ? PROF_NTT_INV : PROF_NTT_FWD, (double)(count * nslots));
  // device copy of the slot -> prime map unless it is the identity over all primes
  int* d_pos = nullptr;
  bool identity = nslots == c->L;
  for (int s = 0; identity && s < nslots; ++s) identity = prime_of_slot_host[s] == s;
  if (!identity) {
    void* p;
    FHESI_TRY(ws_reserve(c, 7, nslots * sizeof(int) + 64, &p));
    HIP_TRY(hipMemcpyAsync(p, prime_of_slot_host, nslots * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    d_pos = (int*)p;
  }
  // the convolution buffer holds 3 auxiliary-prime rows of N words per row: bound it (and the auxiliary transform's scratch of
  // the same size) to about 2 GiB per chunk -- thousands of rows, far more than the 512 workgroups the chip holds
  i64 chunk = (i64)((2ull << 30) / ((u64)nslots * 3 * N * 8));
  if (chunk > 21845 / nslots) chunk = 21845 / nslots;      // grid.y of blue_mul is 3 * rows <= 65535
  if (chunk < 1) chunk = 1;
  if (chunk > count) chunk = count;
  void *dX, *dF = nullptr;
  FHESI_TRY(ws_reserve(c, 8, (size_t)chunk * nslots * 3 * N * 8, &dX));
  if (inv) FHESI_TRY(ws_reserve(c, 9, (size_t)chunk * nslots * (B->phi_conv ? 2 * m - phim : m) * 8, &dF));      // (phi_conv: the quotients behind the polynomials)
  hipStream_t aux_stream = B->aux->stream;
  B->aux->stream = c->stream;
  int rc = 0;
  for (i64 done = 0; done < count && !rc; done += chunk)
    rc = blue_chunk(c, d_rows + done * nslots * phim, std::min(chunk, count - done), nslots, d_pos, inv, (u64*)dX, (u64*)dF);
  B->aux->stream = aux_stream;
  return rc;
}

int launch_bluestein_fwd(fhesi_ctx* c, u64* d_rows, i64 count, int nslots, const int* prime_of_slot_host) { return blue_run(c, d_rows, count, nslots, prime_of_slot_host, false); }
int launch_bluestein_inv(fhesi_ctx* c, u64* d_rows, i64 count, int nslots, const int* prime_of_slot_host) { return blue_run(c, d_rows, count, nslots, prime_of_slot_host, true); }
