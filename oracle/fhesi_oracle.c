/* fhesi_oracle.c -- CPU restatement (plain C) of fhe-si's DoubleCRT hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library, and only as the checker / reported CPU baseline.  The product path
 * (fhe-si_amd/csrc, the HIP library behind include/fhesi_hip.h) never links, loads or calls it.
 *
 * PARITY UNPINNED: the reference needs NTL (absent, un-vendored, no version pinned) so it cannot
 * be built here, and its tests hold no golden vectors for this path.
 * This restatement is pinned by (i) the independent Python big-int restatement
 * (oracle/fhesi_pyref.py) through tests/golden fixtures, (ii) the reference's own slow definition
 * tDFT (bluestein.cpp:149-172) and (iii) the end-to-end predicate of Test_AddMul.cpp:84-86.
 *
 * Conventions (same as include/fhesi_hip.h):
 *   rows      uint64_t[phim], canonical residues in [0,q_i), Z_m^* ascending order
 *   big ints  little-endian 64-bit limbs, two's complement, fixed nlimbs, coefficient-major
 * Citations are file:line relative to /root/reference.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;
typedef int64_t i64;

/* ------------------------------------------------------------------ word-size modular arithmetic
 * role of NTL AddMod/SubMod/MulMod/PowerMod/InvMod on `long` (used throughout DoubleCRT.cpp) */
static inline u64 addmod(u64 a, u64 b, u64 q) { u64 s = a + b; return s >= q ? s - q : s; }
static inline u64 submod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }
static inline u64 mulmod(u64 a, u64 b, u64 q) { return (u64)(((u128)a * b) % q); }
static u64 powmod(u64 a, u64 e, u64 q) { u64 r = 1 % q; a %= q; while (e) { if (e & 1) r = mulmod(r, a, q); a = mulmod(a, a, q); e >>= 1; } return r; }
static u64 invmod(u64 a, u64 q) { /* q prime */ return powmod(a, q - 2, q); }
/* precomputed-quotient multiply (role of NTL MulModPrecon, DoubleCRT.cpp:195-197) */
static inline u64 shoup_pre(u64 w, u64 q) { return (u64)(((u128)w << 64) / q); }
static inline u64 mulmod_shoup(u64 y, u64 w, u64 wp, u64 q) { u64 Q = (u64)(((u128)y * wp) >> 64); u64 r = y * w - Q * q; return r >= q ? r - q : r; }

static int is_prime_u64(u64 n) {
  static const u64 sp[] = {2,3,5,7,11,13,17,19,23,29,31,37};
  if (n < 2) return 0;
  for (int i = 0; i < 12; i++) { if (n % sp[i] == 0) return n == sp[i]; }
  u64 d = n - 1; int s = 0; while ((d & 1) == 0) { d >>= 1; s++; }
  for (int i = 0; i < 12; i++) {
    u64 x = powmod(sp[i], d, n); if (x == 1 || x == n - 1) continue;
    int ok = 0; for (int r = 1; r < s; r++) { x = mulmod(x, x, n); if (x == n - 1) { ok = 1; break; } }
    if (!ok) return 0;
  }
  return 1;
}
static u64 gcd_u64(u64 a, u64 b) { while (b) { u64 t = a % b; a = b; b = t; } return a; }

/* ------------------------------------------------------------------ fixed-width signed big ints */
static int bn_sign(const u64* a, int n) { return (int)(a[n - 1] >> 63); }
static void bn_neg(u64* a, int n) { u64 c = 1; for (int i = 0; i < n; i++) { u64 v = ~a[i] + c; c = (c && v == 0); a[i] = v; } }
static void bn_add(u64* a, const u64* b, int n) { u64 c = 0; for (int i = 0; i < n; i++) { u128 s = (u128)a[i] + b[i] + c; a[i] = (u64)s; c = (u64)(s >> 64); } }
static void bn_sub(u64* a, const u64* b, int n) { u64 br = 0; for (int i = 0; i < n; i++) { u128 s = (u128)a[i] - b[i] - br; a[i] = (u64)s; br = (u64)(s >> 64) & 1; } }
static void bn_set_i64(u64* a, i64 v, int n) { a[0] = (u64)v; for (int i = 1; i < n; i++) a[i] = v < 0 ? ~0ull : 0; }
static void bn_mul_u64(u64* a, u64 m, int n) { /* magnitude multiply, a >= 0 */ u64 c = 0; for (int i = 0; i < n; i++) { u128 s = (u128)a[i] * m + c; a[i] = (u64)s; c = (u64)(s >> 64); } }
static u64 bn_mod_u64_mag(const u64* a, int n, u64 q) { u64 r = 0; for (int i = n - 1; i >= 0; i--) r = (u64)((((u128)r << 64) | a[i]) % q); return r; }
/* non-negative residue of a signed big int (role of NTL rem(ZZ,long): result in [0,q)) */
static u64 bn_mod_u64(const u64* a, int n, u64 q) {
  if (!bn_sign(a, n)) return bn_mod_u64_mag(a, n, q);
  u64 tmp[n]; memcpy(tmp, a, 8 * n); bn_neg(tmp, n);
  u64 r = bn_mod_u64_mag(tmp, n, q); return r ? q - r : 0;
}
static int bn_cmp_signed(const u64* a, const u64* b, int n) {
  int sa = bn_sign(a, n), sb = bn_sign(b, n); if (sa != sb) return sa ? -1 : 1;
  for (int i = n - 1; i >= 0; i--) if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1; return 0;
}
/* a += s * p  where s is a signed word, p a non-negative big int */
static void bn_addmul_i64(u64* a, const u64* p, i64 s, int n) {
  u64 t[n]; memcpy(t, p, 8 * n); u64 mag = s < 0 ? (u64)(-(s + 1)) + 1 : (u64)s; bn_mul_u64(t, mag, n);
  if (s < 0) bn_sub(a, t, n); else bn_add(a, t, n);
}
/* arithmetic shift right by k bits */
static void bn_sar(u64* a, int n, int k) {
  u64 fill = bn_sign(a, n) ? ~0ull : 0; int w = k / 64, b = k % 64;
  for (int i = 0; i < n; i++) { u64 lo = (i + w < n) ? a[i + w] : fill; u64 hi = (i + w + 1 < n) ? a[i + w + 1] : fill; a[i] = b ? (lo >> b) | (hi << (64 - b)) : lo; }
}
static void bn_shl(u64* a, int n, int k) {
  int w = k / 64, b = k % 64;
  for (int i = n - 1; i >= 0; i--) { u64 hi = (i - w >= 0) ? a[i - w] : 0; u64 lo = (i - w - 1 >= 0) ? a[i - w - 1] : 0; a[i] = b ? (hi << b) | (lo >> (64 - b)) : hi; }
}
/* sign-extending copy between widths */
static void bn_copy_ext(u64* dst, int nd, const u64* src, int ns) {
  u64 fill = bn_sign(src, ns) ? ~0ull : 0;
  for (int i = 0; i < nd; i++) dst[i] = i < ns ? src[i] : fill;
}

/* ------------------------------------------------------------------ context (FHEcontext / PAlgebra / Cmodulus) */
typedef struct {
  u64 q, root, rinv;          /* CModulus.h:44,53-54 */
  int pow2;
  /* power-of-two m: negacyclic tables, psi = root^2 */
  u64 *psi, *psi_sh, *ipsi, *ipsi_sh; u64 ninv, ninv_sh;
  /* general m: Bluestein tables (bluestein.cpp:103-109,121-133) */
  u64 *powers, *b, *ipowers, *ib; u64 minv;
  /* FFT form of Bluestein (orc_set_bluestein_fft): N-point cyclic transforms of b / ib modulo the three auxiliary primes
   * (the reference's fftRep Rb, bluestein.cpp:121-136) */
  u64 *Rb[3], *iRb[3];
} orc_prime;

/* one auxiliary FFT prime of the N-point cyclic convolution (role of NTL's FFT primes behind fftRep) */
typedef struct { u64 p; u64 *w, *wsh, *iw, *iwsh; u64 ninv, ninv_sh; } orc_fftprime;

typedef struct {
  i64 m, phim; int L;
  int* zms_idx;               /* PAlgebra.cpp:50-52 */
  i64* phi;                   /* Phi_m(X) coefficients, length phim+1 (PAlgebra.cpp:55) */
  orc_prime* pr;
  int use_slow_dft;           /* evaluate through the tDFT definition instead (bluestein.cpp:149-172) */
  int use_bluestein_fft;      /* evaluate EVERY m the way the reference does: Bluestein + N-point cyclic convolution by multi-prime FFT + CRT */
  i64 N; int logN; orc_fftprime fp[3];
} orc_ctx;

static int mobius_i(i64 n) { int mu = 1; for (i64 p = 2; p * p <= n; p++) if (n % p == 0) { n /= p; if (n % p == 0) return 0; mu = -mu; } if (n > 1) mu = -mu; return mu; }

/* Cyclotomic (NumbTh.cpp:142-158): prod over d|m of (X^{m/d}-1)^{mu(d)}; small integer coefficients for the
 * m handled here (i64 suffices; checked against the Python restatement in tests). */
static i64* cyclotomic(i64 m, i64 phim) {
  /* numerator / denominator degrees are sums of m/d over the divisors with mu(d) = +-1: bounded by sigma(m) < m (1 + ln m) */
  i64* num = calloc(32 * m + 2, 8); i64* den = calloc(32 * m + 2, 8); i64 dn = 0, dd = 0; num[0] = 1; den[0] = 1;
  for (i64 d = 1; d <= m; d++) if (m % d == 0) {
    int mu = mobius_i(d); if (!mu) continue; i64 e = m / d;
    i64* t = mu == 1 ? num : den; i64* dg = mu == 1 ? &dn : &dd;
    /* t *= (X^e - 1) */
    for (i64 i = *dg + e; i >= 0; i--) { i64 hi = (i - e >= 0 && i - e <= *dg) ? t[i - e] : 0; i64 lo = (i <= *dg) ? t[i] : 0; t[i] = hi - lo; }
    *dg += e;
  }
  /* exact division num/den, den monic up to sign */
  i64* quo = calloc(phim + 1, 8);
  for (i64 i = dn - dd; i >= 0; i--) { i64 c = num[i + dd] / den[dd]; quo[i] = c; if (!c) continue; for (i64 j = 0; j <= dd; j++) num[i + j] -= c * den[j]; }
  free(num); free(den); return quo;
}

static u64 brv(u64 x, int bits) { u64 r = 0; for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; } return r; }
static int ilog2(i64 n) { int k = 0; while ((1ll << k) < n) k++; return k; }

static char orc_err[256];
const char* orc_last_error(void) { return orc_err; }

void orc_ctx_destroy(orc_ctx* c) {
  if (!c) return;
  for (int i = 0; i < c->L; i++) { orc_prime* p = &c->pr[i]; free(p->psi); free(p->psi_sh); free(p->ipsi); free(p->ipsi_sh); free(p->powers); free(p->b); free(p->ipowers); free(p->ib);
    for (int a = 0; a < 3; a++) { free(p->Rb[a]); free(p->iRb[a]); } }
  for (int a = 0; a < 3; a++) { free(c->fp[a].w); free(c->fp[a].wsh); free(c->fp[a].iw); free(c->fp[a].iwsh); }
  free(c->pr); free(c->zms_idx); free(c->phi); free(c);
}

/* FHEcontext::AddPrime (FHEContext.cpp:30-43) + Cmod::privateInit (CModulus.cpp:60-86); tables are built
 * eagerly instead of lazily (bluestein.cpp:103,121). root[i]==0 is rejected: the caller supplies roots. */
orc_ctx* orc_ctx_create(i64 m, int L, const u64* q, const u64* root) {
  orc_err[0] = 0;
  if (m < 2 || m > (1 << 20)) { snprintf(orc_err, 256, "m undefined or larger than 2^20"); return NULL; }
  orc_ctx* c = calloc(1, sizeof(*c)); c->m = m; c->L = L;
  c->zms_idx = malloc(sizeof(int) * m); i64 k = 0;
  for (i64 i = 0; i < m; i++) c->zms_idx[i] = (gcd_u64(i, m) == 1) ? (int)k++ : -1;
  c->phim = k; c->phi = cyclotomic(m, k);
  c->pr = calloc(L, sizeof(orc_prime));
  int pow2 = (m & (m - 1)) == 0 && m >= 4;
  for (int i = 0; i < L; i++) {
    orc_prime* p = &c->pr[i]; p->q = q[i]; p->root = root[i]; p->pow2 = pow2;
    int dup = 0; for (int j = 0; j < i; j++) dup |= (q[j] == q[i]);
    if (!is_prime_u64(q[i]) || q[i] % (2 * m) != 1 || dup) { snprintf(orc_err, 256, "AddPrime: prime %d rejected (not prime, not 1 mod 2m, or already in chain)", i); orc_ctx_destroy(c); return NULL; }
    if (powmod(root[i], m, q[i]) != q[i] - 1) { snprintf(orc_err, 256, "root %d is not a primitive 2m-th root of unity", i); orc_ctx_destroy(c); return NULL; }
    p->rinv = invmod(p->root, p->q);
    u64 Q = p->q;
    if (pow2) {
      i64 n = m / 2; int lg = ilog2(n); u64 psi = mulmod(p->root, p->root, Q), ipsi = invmod(psi, Q);
      p->psi = malloc(8 * n); p->psi_sh = malloc(8 * n); p->ipsi = malloc(8 * n); p->ipsi_sh = malloc(8 * n);
      for (i64 j = 0; j < n; j++) { u64 e = brv(j, lg); p->psi[j] = powmod(psi, e, Q); p->psi_sh[j] = shoup_pre(p->psi[j], Q); p->ipsi[j] = powmod(ipsi, e, Q); p->ipsi_sh[j] = shoup_pre(p->ipsi[j], Q); }
      p->ninv = invmod(n % Q, Q); p->ninv_sh = shoup_pre(p->ninv, Q);
    } else {
      p->powers = malloc(8 * m); p->ipowers = malloc(8 * m); p->b = calloc(2 * m, 8); p->ib = calloc(2 * m, 8);
      for (i64 j = 0; j < m; j++) { u64 e = (u64)(((u128)j * j) % (2 * m)); p->powers[j] = powmod(p->root, e, Q); p->ipowers[j] = powmod(p->rinv, e, Q); }
      /* b[m-1+-j] = root^{-j^2}  (bluestein.cpp:126-132); inverse direction swaps the roles of root and rInv */
      for (i64 j = 0; j < m; j++) { p->b[m - 1 + j] = p->b[m - 1 - j] = p->ipowers[j]; p->ib[m - 1 + j] = p->ib[m - 1 - j] = p->powers[j]; }
      p->minv = invmod(m % Q, Q);
    }
  }
  return c;
}
i64 orc_phim(const orc_ctx* c) { return c->phim; }
void orc_set_slow_dft(orc_ctx* c, int on) { c->use_slow_dft = on; }
void orc_get_tables(const orc_ctx* c, int* zms_idx_out, i64* phi_out) { memcpy(zms_idx_out, c->zms_idx, sizeof(int) * c->m); memcpy(phi_out, c->phi, 8 * (c->phim + 1)); }

/* ---- the length-m DFT at all m-th roots: x[k] = sum_i a[i] (root^2)^{ik} ------------------------------- */
/* tDFT (bluestein.cpp:149-172): the reference's own slow definition */
static void tdft(u64* x, const u64* a, i64 n, u64 w, u64 q) {
  for (i64 k = 0; k < n; k++) { u64 base = powmod(w, k, q), term = 1, sum = 0; for (i64 i = 0; i < n; i++) { sum = addmod(sum, mulmod(a[i], term, q), q); term = mulmod(term, base, q); } x[k] = sum; }
}
/* tBluesteinFFT (bluestein.cpp:93-144).  The N-point cyclic product of :138 is evaluated only on the output
 * window n-1..2n-2 (:139); wrap-around terms of the cyclic product never reach that window because N >= 2n-1. */
static void bluestein(u64* x, const u64* a, i64 n, const u64* powers, const u64* b, u64 q) {
  int zero = 1; for (i64 i = 0; i < n; i++) zero &= (a[i] == 0);
  if (zero) { memset(x, 0, 8 * n); return; }                            /* :96-97 */
  u64* t = malloc(8 * n);
  for (i64 i = 0; i < n; i++) t[i] = mulmod(a[i], powers[i], q);        /* :111-113 */
  for (i64 k = 0; k < n; k++) {
    u128 acc = 0; const u64* bb = b + (n - 1 + k);
    for (i64 i = 0; i < n; i++) { acc += (u128)mulmod(t[i], bb[-i], q); }  /* sum of < 2^20 residues < 2^84 */
    x[k] = mulmod((u64)(acc % q), powers[k], q);                         /* :139-142 */
  }
  free(t);
}

/* ---- FFT form of tBluesteinFFT, the algorithm the reference really runs for EVERY m (bluestein.cpp:93-144): the N-point cyclic
 * product of :138 (N = 2^ceil(log2(2m-1)), :117) through NTL's fftRep = forward FFTs modulo several word-size FFT primes, pointwise
 * product with the precomputed transform Rb of the chirp (:121-136), inverse FFTs and CRT back to Z_q (:119,135,138-139).  NTL is
 * absent here, so the FFT primes are three primes = 1 mod N just below 2^62 (their product exceeds the largest coefficient
 * m * q^2 < 2^140 of the integer convolution, which is what makes the result exact for any choice).  Used as the like-for-like CPU
 * baseline of bench.py and cross-checked against the direct evaluation in tests/test_oracle_golden.py. */
static void cyc_ntt(u64* a, i64 N, int logN, const u64* w, const u64* wsh, u64 p) {   /* in-place, natural in / natural out */
  for (i64 i = 0; i < N; i++) { i64 j = (i64)brv((u64)i, logN); if (j > i) { u64 t = a[i]; a[i] = a[j]; a[j] = t; } }
  for (i64 len = 1; len < N; len <<= 1) { i64 step = N / (2 * len);
    for (i64 i = 0; i < N; i += 2 * len) for (i64 j = 0; j < len; j++) {
      u64 u = a[i + j], v = mulmod_shoup(a[i + j + len], w[j * step], wsh[j * step], p);
      a[i + j] = addmod(u, v, p); a[i + j + len] = submod(u, v, p); } }
}
static int bluestein_fft_setup(orc_ctx* c) {
  if (c->N) return 0;
  i64 m = c->m; int lg = 0; while ((1ll << lg) < 2 * m - 1) lg++;
  i64 N = 1ll << lg; c->N = N; c->logN = lg;
  u64 cand = ((1ull << 62) / (u64)N) * (u64)N + 1;
  for (int a = 0; a < 3; a++) {
    do { cand -= (u64)N; } while (!is_prime_u64(cand));
    orc_fftprime* f = &c->fp[a]; f->p = cand;
    u64 g = 0; for (u64 s = 2; s < 1000 && !g; s++) { u64 r = powmod(s, (cand - 1) / (u64)N, cand); if (powmod(r, (u64)N / 2, cand) == cand - 1) g = r; }
    if (!g) { snprintf(orc_err, 256, "no N-th root of unity for the auxiliary prime"); return 1; }
    u64 gi = invmod(g, cand);
    f->w = malloc(8 * (N / 2)); f->wsh = malloc(8 * (N / 2)); f->iw = malloc(8 * (N / 2)); f->iwsh = malloc(8 * (N / 2));
    u64 x = 1, y = 1;
    for (i64 j = 0; j < N / 2; j++) { f->w[j] = x; f->wsh[j] = shoup_pre(x, cand); f->iw[j] = y; f->iwsh[j] = shoup_pre(y, cand); x = mulmod(x, g, cand); y = mulmod(y, gi, cand); }
    f->ninv = invmod((u64)N % cand, cand); f->ninv_sh = shoup_pre(f->ninv, cand);
  }
  for (int i = 0; i < c->L; i++) {
    orc_prime* p = &c->pr[i]; u64 Q = p->q;
    if (!p->powers) {      /* power-of-two m keeps only the negacyclic tables: build the chirp tables of bluestein.cpp:103-109,126-132 too */
      p->powers = malloc(8 * m); p->ipowers = malloc(8 * m); p->b = calloc(2 * m, 8); p->ib = calloc(2 * m, 8);
      for (i64 j = 0; j < m; j++) { u64 e = (u64)(((u128)j * j) % (2 * m)); p->powers[j] = powmod(p->root, e, Q); p->ipowers[j] = powmod(p->rinv, e, Q); }
      for (i64 j = 0; j < m; j++) { p->b[m - 1 + j] = p->b[m - 1 - j] = p->ipowers[j]; p->ib[m - 1 + j] = p->ib[m - 1 - j] = p->powers[j]; }
      p->minv = invmod(m % Q, Q);
    }
    for (int a = 0; a < 3; a++) {
      const orc_fftprime* f = &c->fp[a];
      p->Rb[a] = calloc(N, 8); p->iRb[a] = calloc(N, 8);
      for (i64 j = 0; j < 2 * m - 1; j++) { p->Rb[a][j] = p->b[j] % f->p; p->iRb[a][j] = p->ib[j] % f->p; }
      cyc_ntt(p->Rb[a], N, lg, f->w, f->wsh, f->p); cyc_ntt(p->iRb[a], N, lg, f->w, f->wsh, f->p);
    }
  }
  return 0;
}
int orc_set_bluestein_fft(orc_ctx* c, int on) { c->use_bluestein_fft = on; return on ? bluestein_fft_setup(c) : 0; }
static void bluestein_fft(const orc_ctx* c, u64* x, const u64* a, const u64* powers, u64* const Rb[3], u64 q) {
  i64 n = c->m, N = c->N;
  int zero = 1; for (i64 i = 0; i < n; i++) zero &= (a[i] == 0);
  if (zero) { memset(x, 0, 8 * n); return; }                            /* bluestein.cpp:96-97 */
  u64* t = malloc(8 * n); u64* r[3];
  for (i64 i = 0; i < n; i++) t[i] = mulmod(a[i], powers[i], q);        /* :111-113 */
  for (int k = 0; k < 3; k++) {
    const orc_fftprime* f = &c->fp[k]; u64* v = calloc(N, 8); r[k] = v;
    for (i64 i = 0; i < n; i++) v[i] = t[i] % f->p;                     /* TofftRep :119 */
    cyc_ntt(v, N, c->logN, f->w, f->wsh, f->p);
    for (i64 i = 0; i < N; i++) v[i] = mulmod(v[i], Rb[k][i], f->p);    /* mul(Ra, Ra, Rb) :138 */
    cyc_ntt(v, N, c->logN, f->iw, f->iwsh, f->p);                       /* FromfftRep :139 (window n-1 .. 2n-2) */
    for (i64 i = n - 1; i < 2 * n - 1; i++) v[i] = mulmod_shoup(v[i], f->ninv, f->ninv_sh, f->p);
  }
  /* CRT of the three residues (Garner), reduced modulo q on the way: value = r0 + p0 (d1 + p1 d2) */
  u64 p0 = c->fp[0].p, p1 = c->fp[1].p, p2 = c->fp[2].p;
  u64 i01 = invmod(p0 % p1, p1), i012 = invmod(mulmod(p0 % p2, p1 % p2, p2), p2);
  for (i64 k = 0; k < n; k++) {
    u64 r0 = r[0][n - 1 + k], r1 = r[1][n - 1 + k], r2 = r[2][n - 1 + k];
    u64 d1 = mulmod(submod(r1 % p1, r0 % p1, p1), i01, p1);
    u64 v01 = addmod(r0 % p2, mulmod(p0 % p2, d1 % p2, p2), p2);
    u64 d2 = mulmod(submod(r2, v01, p2), i012, p2);
    u64 acc = addmod(d1 % q, mulmod(p1 % q, d2 % q, q), q);
    acc = addmod(r0 % q, mulmod(p0 % q, acc, q), q);
    x[k] = mulmod(acc, powers[k], q);                                   /* :140-142 */
  }
  free(t); for (int k = 0; k < 3; k++) free(r[k]);
}

/* negacyclic NTT, natural-order in and out (power-of-two m): y[j] = sum_k a_k psi^{(2j+1)k} */
static void ntt_pow2_fwd(u64* y, const u64* a_in, i64 n, const orc_prime* p) {
  u64 q = p->q; int lg = ilog2(n); u64* a = malloc(8 * n); memcpy(a, a_in, 8 * n);
  i64 t = n;
  for (i64 mm = 1; mm < n; mm <<= 1) { t >>= 1;
    for (i64 i = 0; i < mm; i++) { u64 s = p->psi[mm + i], sp = p->psi_sh[mm + i]; i64 j1 = 2 * i * t;
      for (i64 j = j1; j < j1 + t; j++) { u64 u = a[j], v = mulmod_shoup(a[j + t], s, sp, q); a[j] = addmod(u, v, q); a[j + t] = submod(u, v, q); } } }
  for (i64 j = 0; j < n; j++) y[j] = a[brv(j, lg)];
  free(a);
}
static void ntt_pow2_inv(u64* x, const u64* y, i64 n, const orc_prime* p) {
  u64 q = p->q; int lg = ilog2(n); u64* a = malloc(8 * n);
  for (i64 j = 0; j < n; j++) a[j] = y[brv(j, lg)];
  i64 t = 1;
  for (i64 mm = n; mm > 1; mm >>= 1) { i64 h = mm >> 1, j1 = 0;
    for (i64 i = 0; i < h; i++) { u64 s = p->ipsi[h + i], sp = p->ipsi_sh[h + i];
      for (i64 j = j1; j < j1 + t; j++) { u64 u = a[j], v = a[j + t]; a[j] = addmod(u, v, q); a[j + t] = mulmod_shoup(submod(u, v, q), s, sp, q); } j1 += 2 * t; }
    t <<= 1; }
  for (i64 j = 0; j < n; j++) x[j] = mulmod_shoup(a[j], p->ninv, p->ninv_sh, q);
  free(a);
}

/* Cmod::FFT on residues (CModulus.cpp:90-107 after conv :96): xres has ncoeffs entries in [0,q) */
void orc_fft_residues(const orc_ctx* c, int i, const u64* xres, i64 ncoeffs, u64* y) {
  const orc_prime* p = &c->pr[i]; i64 m = c->m; u64 q = p->q;
  if (c->use_bluestein_fft) {
    u64* in = calloc(m, 8); u64* out = malloc(8 * m);
    for (i64 k = 0; k < ncoeffs && k < m; k++) in[k] = xres[k];
    bluestein_fft(c, out, in, p->powers, p->Rb, q);
    for (i64 k = 0, j = 0; k < m; k++) if (c->zms_idx[k] >= 0) y[j++] = out[k];
    free(in); free(out); return;
  }
  if (p->pow2 && !c->use_slow_dft) {
    i64 n = m / 2; u64* a = calloc(n, 8);
    /* degree >= m ignored (bluestein.cpp:111-113); X^n = -1 at primitive m-th roots folds n..m-1 */
    for (i64 k = 0; k < ncoeffs && k < m; k++) { if (k < n) a[k] = addmod(a[k], xres[k], q); else a[k - n] = submod(a[k - n], xres[k], q); }
    ntt_pow2_fwd(y, a, n, p); free(a); return;
  }
  u64* in = calloc(m, 8); u64* out = malloc(8 * m);
  for (i64 k = 0; k < ncoeffs && k < m; k++) in[k] = xres[k];
  if (c->use_slow_dft || p->pow2) { u64 w = mulmod(p->root, p->root, q); int zero = 1; for (i64 k = 0; k < m; k++) zero &= !in[k]; if (zero) memset(out, 0, 8 * m); else tdft(out, in, m, w, q); }
  else bluestein(out, in, m, p->powers, p->b, q);
  for (i64 k = 0, j = 0; k < m; k++) if (c->zms_idx[k] >= 0) y[j++] = out[k];   /* CModulus.cpp:103-106 */
  free(in); free(out);
}

/* Cmod::FFT (CModulus.cpp:90-107) on big-int coefficients */
void orc_cmod_fft(const orc_ctx* c, int i, const u64* limbs, int nlimbs, i64 ncoeffs, u64* y) {
  u64* r = malloc(8 * (ncoeffs ? ncoeffs : 1));
  for (i64 k = 0; k < ncoeffs; k++) r[k] = bn_mod_u64(limbs + k * nlimbs, nlimbs, c->pr[i].q);   /* conv(in,x) :96 */
  orc_fft_residues(c, i, r, ncoeffs, y); free(r);
}

/* Cmod::iFFT (CModulus.cpp:110-132): returns phim coefficients in [0,q) */
void orc_cmod_ifft(const orc_ctx* c, int i, const u64* y, u64* x) {
  const orc_prime* p = &c->pr[i]; i64 m = c->m, phim = c->phim; u64 q = p->q;
  if (p->pow2 && !c->use_slow_dft && !c->use_bluestein_fft) { ntt_pow2_inv(x, y, phim, p); return; }
  u64* in = calloc(m, 8); u64* out = malloc(8 * m);
  for (i64 k = 0, j = 0; k < m; k++) if (c->zms_idx[k] >= 0) in[k] = y[j++];      /* :117-121 */
  u64 minv = p->pow2 ? invmod(m % q, q) : p->minv;
  if (c->use_bluestein_fft) bluestein_fft(c, out, in, p->ipowers, p->iRb, q);
  else if (c->use_slow_dft || p->pow2) { u64 w = mulmod(p->rinv, p->rinv, q); int zero = 1; for (i64 k = 0; k < m; k++) zero &= !in[k]; if (zero) memset(out, 0, 8 * m); else tdft(out, in, m, w, q); }
  else bluestein(out, in, m, p->ipowers, p->ib, q);                                /* :124 */
  for (i64 k = 0; k < m; k++) out[k] = mulmod(out[k], minv, q);                    /* :125 */
  /* rem(out, out, Phi_m) over Z_q (:128-129), Phi_m monic */
  for (i64 k = m - 1; k >= phim; k--) { u64 cc = out[k]; if (!cc) continue;
    for (i64 j = 0; j <= phim; j++) { i64 f = c->phi[j]; if (!f) continue; u64 fm = f < 0 ? q - ((u64)(-f) % q) : (u64)f % q; if (fm == q) fm = 0; out[k - phim + j] = submod(out[k - phim + j], mulmod(cc, fm, q), q); } }
  memcpy(x, out, 8 * phim); free(in); free(out);
}

/* ------------------------------------------------------------------ DoubleCRT (rows: [L][phim], all primes) */
/* DoubleCRT(const ZZX&) (DoubleCRT.cpp:244-257) */
void orc_dcrt_from_poly(const orc_ctx* c, const u64* limbs, int nlimbs, i64 ncoeffs, u64* rows) {
  for (int i = 0; i < c->L; i++) orc_cmod_fft(c, i, limbs, nlimbs, ncoeffs, rows + (i64)i * c->phim);
}
/* DoubleCRT::Op (DoubleCRT.cpp:79-113), op: 0 add, 1 sub, 2 mul; a <- a op b */
void orc_dcrt_op(const orc_ctx* c, u64* a, const u64* b, int op) {
  for (int i = 0; i < c->L; i++) { u64 q = c->pr[i].q; u64* ra = a + (i64)i * c->phim; const u64* rb = b + (i64)i * c->phim;
    for (i64 j = 0; j < c->phim; j++) ra[j] = op == 0 ? addmod(ra[j], rb[j], q) : op == 1 ? submod(ra[j], rb[j], q) : mulmod(ra[j], rb[j], q); }
}
/* DoubleCRT::Op(const ZZ&) (DoubleCRT.cpp:115-129); op 3 = operator/= (:407-420); op 4 = operator=(ZZ) (:333-347) */
void orc_dcrt_op_scalar(const orc_ctx* c, u64* a, const u64* num, int nlimbs, int op) {
  for (int i = 0; i < c->L; i++) { u64 q = c->pr[i].q; u64 s = bn_mod_u64(num, nlimbs, q); if (op == 3) s = invmod(s, q); u64* ra = a + (i64)i * c->phim;
    for (i64 j = 0; j < c->phim; j++) ra[j] = op == 0 ? addmod(ra[j], s, q) : op == 1 ? submod(ra[j], s, q) : op == 4 ? s : mulmod(ra[j], s, q); }
}
/* DoubleCRT::Exp (DoubleCRT.cpp:423-434): PowerMod per element; e < 0 inverts first, -1 if some element is 0 (NTL InvMod error) */
int orc_dcrt_exp(const orc_ctx* c, u64* a, i64 e) {
  if (e < 0) for (i64 j = 0; j < (i64)c->L * c->phim; j++) if (a[j] == 0) return -1;
  for (int i = 0; i < c->L; i++) { u64 q = c->pr[i].q; u64* ra = a + (i64)i * c->phim;
    for (i64 j = 0; j < c->phim; j++) { u64 b = e < 0 ? invmod(ra[j], q) : ra[j]; ra[j] = powmod(b, e < 0 ? 0 - (u64)e : (u64)e, q); } }
  return 0;
}
/* DoubleCRT::automorph (DoubleCRT.cpp:439-465); returns -1 if k not in Zm* (:442-443) */
int orc_dcrt_automorph(const orc_ctx* c, u64* a, i64 k) {
  i64 m = c->m; if (k <= 0 || k >= m || c->zms_idx[k] < 0) return -1;
  u64* tmp = malloc(8 * m);
  for (int i = 0; i < c->L; i++) { u64* row = a + (i64)i * c->phim;
    for (i64 j = 1; j < m; j++) if (c->zms_idx[j] >= 0) tmp[j] = row[c->zms_idx[j]];
    for (i64 j = 1; j < m; j++) if (c->zms_idx[j] >= 0) row[c->zms_idx[j]] = tmp[(j * k) % m]; }
  free(tmp); return 0;
}

/* intVecCRT (NumbTh.cpp:307-335) on fixed-width signed big ints vp[n][W]; P non-negative big int */
static void int_vec_crt(u64* vp, const u64* P, int W, const u64* vq, i64 n, u64 q) {
  u64 pinv = invmod(bn_mod_u64_mag(P, W, q), q); u64 q2 = q / 2;
  for (i64 i = 0; i < n; i++) { u64* v = vp + i * W;
    u64 d = mulmod(submod(vq[i], bn_mod_u64(v, W, q), q), pinv, q);
    i64 ds = d > q2 ? (i64)d - (i64)q : (i64)d;
    bn_addmul_i64(v, P, ds, W); }
}

/* DoubleCRT::toPoly (DoubleCRT.cpp:349-398).  idx: ascending prime indices (NULL = all), out[phim][nlimbs]
 * two's complement (values that do not fit are truncated mod 2^(64 nlimbs), like any fixed-width store). */
void orc_dcrt_to_poly(const orc_ctx* c, const u64* rows, const int* idx, int nidx, int positive, u64* out, int nlimbs) {
  i64 n = c->phim; int all[64]; if (!idx) { nidx = c->L; for (int i = 0; i < nidx; i++) all[i] = i; idx = all; }
  if (nidx == 0) { memset(out, 0, 8 * n * nlimbs); return; }
  int W = nidx + 2; u64* vp = calloc(n * W, 8); u64* cur = malloc(8 * n); u64* P = calloc(W, 8);
  u64 p0 = c->pr[idx[0]].q; P[0] = p0;
  orc_cmod_ifft(c, idx[0], rows + (i64)idx[0] * n, cur);
  for (i64 j = 0; j < n; j++) { i64 v = cur[j] > p0 / 2 ? (i64)cur[j] - (i64)p0 : (i64)cur[j]; bn_set_i64(vp + j * W, v, W); }   /* :375-376 */
  for (int t = 1; t < nidx; t++) { u64 q = c->pr[idx[t]].q;
    orc_cmod_ifft(c, idx[t], rows + (i64)idx[t] * n, cur);
    int_vec_crt(vp, P, W, cur, n, q); bn_mul_u64(P, q, W); }                        /* :381-388 */
  for (i64 j = 0; j < n; j++) { u64* v = vp + j * W; if (positive && bn_sign(v, W)) bn_add(v, P, W); bn_copy_ext(out + j * nlimbs, nlimbs, v, W); }
  free(vp); free(cur); free(P);
}

/* ------------------------------------------------------------------ SingleCRT (SingleCRT.cpp): rows [L][phim] of COEFFICIENT residues */
/* SingleCRT::operator=(const ZZX&) (SingleCRT.cpp:239-251): PolyRed(poly, p_i, abs=true) per prime (NumbTh.cpp:210-233) */
void orc_scrt_from_poly(const orc_ctx* c, const u64* limbs, int nlimbs, i64 ncoeffs, u64* rows) {
  i64 n = c->phim; memset(rows, 0, 8 * n * c->L);
  for (int i = 0; i < c->L; i++) for (i64 k = 0; k < ncoeffs && k < n; k++) rows[(i64)i * n + k] = bn_mod_u64(limbs + k * nlimbs, nlimbs, c->pr[i].q);
}
/* SingleCRT::toPoly (SingleCRT.cpp:299-334): first row centred (:320-321), then intVecCRT per further prime (:323-328) */
void orc_scrt_to_poly(const orc_ctx* c, const u64* rows, const int* idx, int nidx, u64* out, int nlimbs) {
  i64 n = c->phim; int all[64]; if (!idx) { nidx = c->L; for (int i = 0; i < nidx; i++) all[i] = i; idx = all; }
  if (nidx == 0) { memset(out, 0, 8 * n * nlimbs); return; }
  int W = nidx + 2; u64* vp = calloc(n * W, 8); u64* P = calloc(W, 8);
  u64 p0 = c->pr[idx[0]].q; P[0] = p0; const u64* r0 = rows + (i64)idx[0] * n;
  for (i64 j = 0; j < n; j++) { i64 v = r0[j] > p0 / 2 ? (i64)r0[j] - (i64)p0 : (i64)r0[j]; bn_set_i64(vp + j * W, v, W); }
  for (int t = 1; t < nidx; t++) { u64 q = c->pr[idx[t]].q; int_vec_crt(vp, P, W, rows + (i64)idx[t] * n, n, q); bn_mul_u64(P, q, W); }
  for (i64 j = 0; j < n; j++) bn_copy_ext(out + j * nlimbs, nlimbs, vp + j * W, W);
  free(vp); free(P);
}
/* SingleCRT::Op(const ZZ&, NTL::add / sub / mul) (SingleCRT.cpp:137-153) and operator/= (:279-296); op: 0 add, 1 sub, 2 mul, 3 div.
 * add / sub of a ZZX and a scalar change the constant coefficient only.  Returns 1 when the divisor is not invertible. */
int orc_scrt_op_scalar(const orc_ctx* c, u64* rows, const u64* num, int nlimbs, int op) {
  i64 n = c->phim;
  for (int i = 0; i < c->L; i++) { u64 q = c->pr[i].q, v = bn_mod_u64(num, nlimbs, q); u64* r = rows + (i64)i * n;
    if (op == 0) r[0] = addmod(r[0], v, q);
    else if (op == 1) r[0] = submod(r[0], v, q);
    else { if (op == 3) { if (!v) return 1; v = invmod(v, q); } for (i64 j = 0; j < n; j++) r[j] = mulmod(r[j], v, q); } }
  return 0;
}

/* ------------------------------------------------------------------ BGV-style modulus switching (DoubleCRT.cpp:162-208, 518-558)
 * Dead code in fhe-si (no callers) but part of the DoubleCRT surface.  Rows are in the full layout [L][phim];
 * index sets are ascending lists of prime indices.  Big integers here are magnitudes in little-endian limbs. */
static void bn_mul_mag(u64* out, const u64* a, int na, const u64* b, int nb) {       /* out[na+nb] = a * b */
  memset(out, 0, 8 * (na + nb));
  for (int i = 0; i < na; i++) { u64 c = 0; for (int j = 0; j < nb; j++) { u128 t = (u128)a[i] * b[j] + out[i + j] + c; out[i + j] = (u64)t; c = (u64)(t >> 64); } out[i + nb] = c; }
}
static int bn_cmp_mag(const u64* a, const u64* b, int n) { for (int i = n - 1; i >= 0; i--) if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1; return 0; }
static void bn_mod_mag(u64* r, int nr, const u64* a, int na, const u64* m) {            /* r[nr] = a mod m, m in nr limbs, m < 2^(64 nr - 1) */
  memset(r, 0, 8 * nr);
  for (int bit = 64 * na - 1; bit >= 0; bit--) {
    u64 c = (a[bit / 64] >> (bit % 64)) & 1;
    for (int i = 0; i < nr; i++) { u64 t = r[i]; r[i] = (t << 1) | c; c = t >> 63; }
    if (bn_cmp_mag(r, m, nr) >= 0) bn_sub(r, m, nr);
  }
}
static void prod_of_primes(const orc_ctx* c, const int* idx, int nidx, u64* out, int W) { memset(out, 0, 8 * W); out[0] = 1; for (int i = 0; i < nidx; i++) bn_mul_u64(out, c->pr[idx[i]].q, W); }

/* DoubleCRT::addPrimesAndScale (DoubleCRT.cpp:162-208): scale the rows of cur_idx by factor = F * (F^-1 mod p), F = product of the
 * added primes; rows of add_idx are zero-filled (:200-205).  Returns the logarithm of the factor (:172,182). */
double orc_dcrt_add_primes_and_scale(const orc_ctx* c, u64* rows, const int* cur_idx, int ncur, const int* add_idx, int nadd, u64 p) {
  if (nadd == 0) return 0.0;                                                          /* :165 */
  i64 n = c->phim; int W = nadd + 2; u64 factor[W]; prod_of_primes(c, add_idx, nadd, factor, W);
  double lf = 0.0; for (int i = 0; i < nadd; i++) lf += log((double)c->pr[add_idx[i]].q);
  u64 prodInv = invmod(bn_mod_u64_mag(factor, W, p), p);                             /* :176-179 (p prime or at least coprime: InvMod) */
  bn_mul_u64(factor, prodInv, W); lf += log((double)prodInv);                         /* :180-182 */
  for (int t = 0; t < ncur; t++) { u64 q = c->pr[cur_idx[t]].q, f = bn_mod_u64_mag(factor, W, q); u64* row = rows + (i64)cur_idx[t] * n;
    for (i64 j = 0; j < n; j++) row[j] = mulmod(row[j], f, q); }                      /* :188-197 */
  for (int t = 0; t < nadd; t++) memset(rows + (i64)add_idx[t] * n, 0, 8 * n);        /* :200-205 */
  return lf;
}

/* DoubleCRT::scaleDownToSet (DoubleCRT.cpp:518-558).  cur_idx: the object's index set, s_idx: the target set; on return the rows of
 * (cur & s) hold the result (the other rows are no longer part of the object).  Returns 0, or 1 when an assertion of :525-526 fails. */
int orc_dcrt_scale_down_to_set(const orc_ctx* c, u64* rows, const int* cur_idx, int ncur, const int* s_idx, int ns, u64 p) {
  i64 n = c->phim; int keep[64], diff[64], nk = 0, nd = 0;
  for (int t = 0; t < ncur; t++) { int in = 0; for (int u = 0; u < ns; u++) in |= (s_idx[u] == cur_idx[t]); if (in) keep[nk++] = cur_idx[t]; else diff[nd++] = cur_idx[t]; }
  if (!nk || !nd) return 1;                                                           /* :525-526 */
  int W = nd + 3; u64 D[W]; prod_of_primes(c, diff, nd, D, W);                        /* diffProd :528 */
  u64 dp = bn_mod_u64_mag(D, W, p);
  for (int t = 0; t < ncur; t++) { u64 q = c->pr[cur_idx[t]].q, f = dp % q; u64* row = rows + (i64)cur_idx[t] * n; for (i64 j = 0; j < n; j++) row[j] = mulmod(row[j], f, q); }   /* :529 */
  u64* delta = malloc(8 * n * W); orc_dcrt_to_poly(c, rows, diff, nd, 0, delta, W);   /* :531-532 */
  /* factor = diffProd * InvMod(diffProd % p, p) (:538); delta[i] = delta[i]*factor - delta[i] (:539-543), then ReduceCoefficientsSlow
   * modulo diffProd * p (:545, Util.cpp:35-43: c %= mod -- NTL's remainder is non-negative for a positive modulus -- then c -= mod
   * when c > mod/2) */
  u64 fm1[W]; memcpy(fm1, D, 8 * W); bn_mul_u64(fm1, invmod(dp, p), W); { u64 one[W]; memset(one, 0, 8 * W); one[0] = 1; bn_sub(fm1, one, W); }   /* factor - 1 >= 0 */
  u64 M[W]; memcpy(M, D, 8 * W); bn_mul_u64(M, p, W);
  u64 halfM[W]; memcpy(halfM, M, 8 * W); for (int i = 0; i < W; i++) halfM[i] = (M[i] >> 1) | (i + 1 < W ? M[i + 1] << 63 : 0);
  u64* e = malloc(8 * n * W); u64 mag[W], prod[2 * W], r[W];
  for (i64 j = 0; j < n; j++) {
    memcpy(mag, delta + j * W, 8 * W); int neg = bn_sign(mag, W); if (neg) bn_neg(mag, W);
    bn_mul_mag(prod, mag, W, fm1, W); bn_mod_mag(r, W, prod, 2 * W, M);
    int zero = 1; for (int i = 0; i < W; i++) zero &= !r[i];
    if (neg && !zero) { u64 t[W]; memcpy(t, M, 8 * W); bn_sub(t, r, W); memcpy(r, t, 8 * W); }
    if (bn_cmp_mag(r, halfM, W) > 0) bn_sub(r, M, W);                                 /* two's complement from here on */
    memcpy(e + j * W, r, 8 * W);
  }
  /* removePrimes(diff); *this += delta; *this /= diffProd  (:555-557) */
  u64* row_e = malloc(8 * n);
  for (int t = 0; t < nk; t++) { int i = keep[t]; u64 q = c->pr[i].q; u64* row = rows + (i64)i * n;
    orc_cmod_fft(c, i, e, W, n, row_e);
    u64 dinv = invmod(bn_mod_u64_mag(D, W, q), q);
    for (i64 j = 0; j < n; j++) row[j] = mulmod(addmod(row[j], row_e[j], q), dinv, q); }
  free(delta); free(e); free(row_e);
  return 0;
}

/* ------------------------------------------------------------------ Util.cpp / Ciphertext.cpp / FHE-SI.cpp */
/* Reduce (Util.cpp:3-26) in place on a two's complement value of `n` limbs */
static void reduce_logq(u64* v, int n, int logQ, int positive) {
  int sign_bit = positive ? 0 : (int)((v[(logQ - 1) / 64] >> ((logQ - 1) % 64)) & 1);
  for (int i = 0; i < n; i++) { int lo = i * 64; if (lo >= logQ) v[i] = sign_bit ? ~0ull : 0; else if (lo + 64 > logQ) { u64 mask = (1ull << (logQ - lo)) - 1; v[i] = sign_bit ? (v[i] | ~mask) : (v[i] & mask); } }
}
void orc_reduce_coeffs(u64* poly, i64 ncoeffs, int nlimbs, int logQ, int positive) { for (i64 j = 0; j < ncoeffs; j++) reduce_logq(poly + j * nlimbs, nlimbs, logQ, positive); }

/* Ciphertext::ScaleDown per component (Ciphertext.cpp:194-218): rows of one tProd component -> nlimbs_out-limb
 * coefficients round-half-up(x / 2^logQ) then centered mod 2^logQ.  floor((2x+q)/(2q)) == (x + q/2) >> logQ. */
void orc_scale_down(const orc_ctx* c, const u64* rows, int logQ, u64* out, int nlimbs_out) {
  i64 n = c->phim; int W = c->L + 3; u64* big = malloc(8 * n * W); orc_dcrt_to_poly(c, rows, NULL, 0, 0, big, W);
  u64 half[W]; memset(half, 0, sizeof(half)); half[(logQ - 1) / 64] = 1ull << ((logQ - 1) % 64);
  for (i64 j = 0; j < n; j++) { u64* v = big + j * W; bn_add(v, half, W); bn_sar(v, W, logQ); reduce_logq(v, W, logQ, 0); bn_copy_ext(out + j * nlimbs_out, nlimbs_out, v, W); }
  free(big);
}

/* Ciphertext::ByteDecompPart (Ciphertext.cpp:82-105): digit d of coefficient j -> digits[d][j] (word each) */
void orc_byte_decomp_part(const u64* poly, i64 ncoeffs, int nlimbs, int logQ, int nd, int decomp_bytes, u64* digits) {
  int bits = 8 * decomp_bytes; u64 tmp[nlimbs + 1];
  for (i64 j = 0; j < ncoeffs; j++) { memcpy(tmp, poly + j * nlimbs, 8 * nlimbs); tmp[nlimbs] = bn_sign(tmp, nlimbs) ? ~0ull : 0; reduce_logq(tmp, nlimbs + 1, logQ, 1);
    for (int d = 0; d < nd; d++) { int lo = bits * d, w = lo / 64, b = lo % 64; u64 v = tmp[w] >> b; if (b + bits > 64 && w + 1 <= nlimbs) v |= tmp[w + 1] << (64 - b); digits[(i64)d * ncoeffs + j] = v & ((1ull << bits) - 1); } }
}

/* Ciphertext::operator*= (Ciphertext.cpp:167-192) for two 2-part ciphertexts: a, b are [2][phim][nlimbs];
 * tprod out is [3][L][phim] */
void orc_ct_mul(const orc_ctx* c, const u64* a, const u64* b, int nlimbs, u64 p, u64* tprod) {
  i64 n = c->phim; i64 rs = (i64)c->L * n; int W = nlimbs + 1;
  u64* c1 = malloc(8 * 2 * rs); u64* c2 = malloc(8 * 2 * rs); u64* lifted = malloc(8 * n * W); u64* tmp = malloc(8 * rs);
  for (int i = 0; i < 2; i++) {
    for (i64 j = 0; j < n; j++) { u64* v = lifted + j * W; bn_copy_ext(v, W, a + ((i64)i * n + j) * nlimbs, nlimbs); int s = bn_sign(v, W); if (s) bn_neg(v, W); bn_mul_u64(v, p, W); if (s) bn_neg(v, W); }   /* poly * p (:171) */
    orc_dcrt_from_poly(c, lifted, W, n, c1 + i * rs);
    orc_dcrt_from_poly(c, b + (i64)i * n * nlimbs, nlimbs, n, c2 + i * rs);
  }
  memset(tprod, 0, 8 * 3 * rs);
  for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) { memcpy(tmp, c1 + i * rs, 8 * rs); orc_dcrt_op(c, tmp, c2 + j * rs, 2); orc_dcrt_op(c, tprod + (i + j) * rs, tmp, 0); }   /* :179-186 */
  free(c1); free(c2); free(lifted); free(tmp);
}

/* KeySwitchSI::ApplyKeySwitch (FHE-SI.cpp:241-260) on a 3-component scaled-up ciphertext.
 * ksm: [2][3*nd][L][phim] (matrix[0]=b, [1]=A; column index i*nd+j, FHE-SI.cpp:176-177,206-208).
 * out: [2][phim][nlimbs] */
void orc_apply_key_switch(const orc_ctx* c, const u64* ksm, const u64* tprod, int ncomp, int logQ, int decomp_bytes, u64* out, int nlimbs) {
  i64 n = c->phim; i64 rs = (i64)c->L * n; int nd = (logQ + 8 * decomp_bytes - 1) / (8 * decomp_bytes); int ncol = ncomp * nd;
  u64* parts = malloc(8 * ncomp * n * nlimbs); u64* dig = malloc(8 * (i64)ncol * n); u64* bd = malloc(8 * (i64)ncol * rs); u64* acc = malloc(8 * rs); u64* tmp = malloc(8 * rs);
  for (int i = 0; i < ncomp; i++) orc_scale_down(c, tprod + i * rs, logQ, parts + (i64)i * n * nlimbs, nlimbs);                      /* ScaleDown :243 */
  for (int i = 0; i < ncomp; i++) orc_byte_decomp_part(parts + (i64)i * n * nlimbs, n, nlimbs, logQ, nd, decomp_bytes, dig + (i64)i * nd * n);  /* ByteDecomp :244 */
  for (int k = 0; k < ncol; k++) orc_dcrt_from_poly(c, dig + (i64)k * n, 1, n, bd + (i64)k * rs);                                     /* :246-249 (digits < 2^24: one non-negative limb) */
  int W = c->L + 3; u64* big = malloc(8 * n * W);
  for (int r = 0; r < 2; r++) { const u64* key = ksm + (i64)r * ncol * rs;
    memcpy(acc, key, 8 * rs); orc_dcrt_op(c, acc, bd, 2);                                                                              /* DotProduct Util.h:79-98 */
    for (int k = 1; k < ncol; k++) { memcpy(tmp, key + (i64)k * rs, 8 * rs); orc_dcrt_op(c, tmp, bd + (i64)k * rs, 2); orc_dcrt_op(c, acc, tmp, 0); }
    orc_dcrt_to_poly(c, acc, NULL, 0, 0, big, W);                                                                                      /* :255 */
    for (i64 j = 0; j < n; j++) { reduce_logq(big + j * W, W, logQ, 0); bn_copy_ext(out + ((i64)r * n + j) * nlimbs, nlimbs, big + j * W, W); }   /* :256 */
  }
  free(parts); free(dig); free(bd); free(acc); free(tmp); free(big);
}

/* The metric's unit of work (Test_AddMul.cpp:59-67): operator*= then ApplyKeySwitch */
void orc_ct_mul_relin(const orc_ctx* c, const u64* ksm, const u64* a, const u64* b, int nlimbs, int logQ, u64 p, int decomp_bytes, u64* out) {
  i64 rs = (i64)c->L * c->phim; u64* tprod = malloc(8 * 3 * rs);
  orc_ct_mul(c, a, b, nlimbs, p, tprod);
  orc_apply_key_switch(c, ksm, tprod, 3, logQ, decomp_bytes, out, nlimbs);
  free(tprod);
}

/* ------------------------------------------------------------------ ciphertext algebra used by Matrix<Ciphertext> / Regression */
/* Ciphertext::operator+= on unscaled ciphertexts (Ciphertext.cpp:123-134): parts add + ReduceCoefficients.  a, b: [nparts][phim][nlimbs] */
void orc_ct_add(const orc_ctx* c, u64* a, const u64* b, int nparts, int nlimbs, int logQ) {
  i64 n = c->phim; int W = nlimbs + 1; u64 x[W], y[W];
  for (i64 j = 0; j < (i64)nparts * n; j++) { bn_copy_ext(x, W, a + j * nlimbs, nlimbs); bn_copy_ext(y, W, b + j * nlimbs, nlimbs); bn_add(x, y, W); reduce_logq(x, W, logQ, 0); memcpy(a + j * nlimbs, x, 8 * nlimbs); }
}
/* Ciphertext::operator*=(long) on an unscaled ciphertext (Ciphertext.cpp:232-237, CiphertextPart::operator*= :21-27) */
void orc_ct_mul_long(const orc_ctx* c, u64* a, i64 l, int nparts, int nlimbs, int logQ) {
  i64 n = c->phim; int W = nlimbs + 2; u64 x[W]; u64 mag = l < 0 ? (u64)(-(l + 1)) + 1 : (u64)l;
  for (i64 j = 0; j < (i64)nparts * n; j++) { bn_copy_ext(x, W, a + j * nlimbs, nlimbs); int s = bn_sign(x, W); if (s) bn_neg(x, W); bn_mul_u64(x, mag, W); if (s != (l < 0)) bn_neg(x, W);
    reduce_logq(x, W, logQ, 0); memcpy(a + j * nlimbs, x, 8 * nlimbs); }
}
/* Ciphertext::operator+=(const ZZX&) on an unscaled ciphertext (Ciphertext.cpp:147-156): scaledConstant[i] = (other[i] << logQ) / p with
 * NTL's floor division (the quotient of a negative numerator rounds towards minus infinity), parts[0] += scaledConstant,
 * ReduceCoefficients.  a: [nparts][phim][nlimbs], only part 0 changes; poly: [phim] machine-word coefficients */
void orc_ct_add_const(const orc_ctx* c, u64* a, const i64* poly, int nlimbs, int logQ, u64 p) {
  i64 n = c->phim; int W = nlimbs + 3; u64 x[W], q[W];
  for (i64 j = 0; j < n; j++) {
    i64 v = poly[j]; u64 mag = v < 0 ? (u64)(-(v + 1)) + 1 : (u64)v;
    memset(x, 0, 8 * W); x[0] = mag; bn_shl(x, W, logQ);                         /* |other| << logQ  (NTL shifts the magnitude) */
    u64 rem = 0; for (int i = W - 1; i >= 0; i--) { u128 cur = ((u128)rem << 64) | x[i]; q[i] = (u64)(cur / p); rem = (u64)(cur % p); }
    if (v < 0) { if (rem) { u64 one[W]; memset(one, 0, 8 * W); one[0] = 1; bn_add(q, one, W); } bn_neg(q, W); }   /* floor(-|x| / p) = -ceil(|x| / p) */
    bn_copy_ext(x, W, a + j * nlimbs, nlimbs); bn_add(x, q, W); reduce_logq(x, W, logQ, 0); memcpy(a + j * nlimbs, x, 8 * nlimbs);
  }
}
/* Ciphertext::operator*=(const ZZX&) on an unscaled ciphertext (Ciphertext.cpp:245-249) = CiphertextPart::operator*=(const ZZX&) (:29-36) on
 * every part: poly *= other as integer polynomials, rem(poly, poly, PhimX), Reduce of every coefficient.  Schoolbook product and long
 * division by the monic Phi_m on fixed-width big integers.  a: [nparts][phim][nlimbs]; poly: [phim] machine words */
void orc_ct_mul_poly(const orc_ctx* c, u64* a, const i64* poly, int nparts, int nlimbs, int logQ) {
  i64 n = c->phim; int W = nlimbs + 3; i64 len = 2 * n - 1;
  u64* prod = malloc(8 * (size_t)len * W); u64 x[W];
  for (int part = 0; part < nparts; part++) {
    u64* ap = a + (i64)part * n * nlimbs;
    memset(prod, 0, 8 * (size_t)len * W);
    for (i64 i = 0; i < n; i++) { bn_copy_ext(x, W, ap + i * nlimbs, nlimbs); int neg = bn_sign(x, W); if (neg) bn_neg(x, W);
      for (i64 j = 0; j < n; j++) { i64 b = poly[j]; if (!b) continue; bn_addmul_i64(prod + (i + j) * W, x, neg ? -b : b, W); } }
    for (i64 d = len - 1; d >= n; d--) {                                           /* rem by the monic Phi_m (degree n = phi(m)) */
      u64* lead = prod + d * W; int neg = bn_sign(lead, W); memcpy(x, lead, 8 * W); if (neg) bn_neg(x, W);
      for (i64 t = 0; t < n; t++) { i64 f = c->phi[t]; if (f) bn_addmul_i64(prod + (d - n + t) * W, x, neg ? f : -f, W); }
      memset(lead, 0, 8 * W);
    }
    for (i64 i = 0; i < n; i++) { reduce_logq(prod + i * W, W, logQ, 0); memcpy(ap + i * nlimbs, prod + i * W, 8 * nlimbs); }
  }
  free(prod);
}
/* Ciphertext::operator>>= on an unscaled ciphertext (Ciphertext.cpp:264-269; CiphertextPart::operator>>= :54-59):
 * DoubleCRT(poly) >>= k; toPoly.  in: [nparts][phim][nlimbs], out: [nparts][phim][nlimbs_out] (centred modulo the chain) */
int orc_ct_automorph(const orc_ctx* c, const u64* in, i64 k, int nparts, int nlimbs, u64* out, int nlimbs_out) {
  i64 n = c->phim; i64 rs = (i64)c->L * n; u64* rows = malloc(8 * rs); int rc = 0;
  for (int i = 0; i < nparts && !rc; i++) { orc_dcrt_from_poly(c, in + (i64)i * n * nlimbs, nlimbs, n, rows); rc = orc_dcrt_automorph(c, rows, k);
    if (!rc) orc_dcrt_to_poly(c, rows, NULL, 0, 0, out + (i64)i * n * nlimbs_out, nlimbs_out); }
  free(rows); return rc;
}
/* KeySwitchSI::ApplyKeySwitch (FHE-SI.cpp:241-260) on an UNSCALED ciphertext of ncomp parts: ScaleDown returns at once
 * (Ciphertext.cpp:195), ByteDecomp takes the positive residue mod 2^logQ (:94).  parts: [ncomp][phim][nlimbs_in] */
void orc_apply_key_switch_parts(const orc_ctx* c, const u64* ksm, const u64* parts, int ncomp, int nlimbs_in, int logQ, int decomp_bytes, u64* out, int nlimbs) {
  i64 n = c->phim; i64 rs = (i64)c->L * n; int nd = (logQ + 8 * decomp_bytes - 1) / (8 * decomp_bytes); int ncol = ncomp * nd;
  u64* dig = malloc(8 * (i64)ncol * n); u64* bd = malloc(8 * (i64)ncol * rs); u64* acc = malloc(8 * rs); u64* tmp = malloc(8 * rs);
  for (int i = 0; i < ncomp; i++) orc_byte_decomp_part(parts + (i64)i * n * nlimbs_in, n, nlimbs_in, logQ, nd, decomp_bytes, dig + (i64)i * nd * n);
  for (int k = 0; k < ncol; k++) orc_dcrt_from_poly(c, dig + (i64)k * n, 1, n, bd + (i64)k * rs);
  int W = c->L + 3; u64* big = malloc(8 * n * W);
  for (int r = 0; r < 2; r++) { const u64* key = ksm + (i64)r * ncol * rs;
    memcpy(acc, key, 8 * rs); orc_dcrt_op(c, acc, bd, 2);
    for (int k = 1; k < ncol; k++) { memcpy(tmp, key + (i64)k * rs, 8 * rs); orc_dcrt_op(c, tmp, bd + (i64)k * rs, 2); orc_dcrt_op(c, acc, tmp, 0); }
    orc_dcrt_to_poly(c, acc, NULL, 0, 0, big, W);
    for (i64 j = 0; j < n; j++) { reduce_logq(big + j * W, W, logQ, 0); bn_copy_ext(out + ((i64)r * n + j) * nlimbs, nlimbs, big + j * W, W); }
  }
  free(dig); free(bd); free(acc); free(tmp); free(big);
}

/* ------------------------------------------------------------------ counter-based randomness for sampling on the device
 * The reference draws from NTL's sequential PRNG (NumbTh.cpp:340-404 via RandomBnd, FHE-SI.cpp:14-25,174-190), which only an NTL process
 * can reproduce.  The device draws instead from Philox-4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3",
 * SC'11) keyed by a seed, counter = (coefficient j, object index low / high word, purpose << 16 | block); this is the checker's own
 * statement of that definition (the product states it in fhe-si_amd/csrc/philox.h, the Python model in fhesi_pyref.py).
 * Known answers of the generator: tests/test_oracle_golden.py. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; r++) {
    if (r) { k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    u64 p0 = (u64)0xD2511F53u * c0, p1 = (u64)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static void draw(u64 seed, u64 object, uint32_t j, uint32_t purpose, uint32_t block, uint32_t w[4]) {
  uint32_t ctr[4] = {j, (uint32_t)object, (uint32_t)(object >> 32), purpose << 16 | block}, key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  orc_philox4x32_10(ctr, key, w);
}
/* floor(2^64 P(|X| <= k)), X = round(N(0, 3.2^2)): the distribution of sampleGaussian's Box-Muller + floor(x + 0.5) (NumbTh.cpp:377-404) with
 * FHEContext.h:106's stdev, sampled by inversion on integers */
static const u64 gauss_cdf[30] = {
  0x1fc936cfb902b000ull, 0x5c5a3878a5513000ull, 0x90ba6b457f6e7800ull, 0xb9d6e65c2e45a000ull, 0xd7212f1da26f6200ull, 0xea12314c4c373a00ull,
  0xf53070328c8acd80ull, 0xfb1cdac9f40b6980ull, 0xfdfa2ace3c107960ull, 0xff3c09d0d6606540ull, 0xffbc45110abb0cdcull, 0xffeaa36c3c86f3c9ull,
  0xfff9db4fc8bf1e60ull, 0xfffe63d9a0b88f51ull, 0xffff9da028c31301ull, 0xffffea9fbab7b7e9ull, 0xfffffbc5e81f8a58ull, 0xffffff3d57e1db7bull,
  0xffffffe027626c9eull, 0xfffffffb4348bc92ull, 0xffffffff5c0429c2ull, 0xffffffffebd8d7e9ull, 0xfffffffffdbfd855ull, 0xffffffffffc58a4cull,
  0xfffffffffffa9c86ull, 0xffffffffffff8c81ull, 0xfffffffffffff738ull, 0xffffffffffffff65ull, 0xfffffffffffffff7ull, 0xffffffffffffffffull };
static i64 gaussian_of(const uint32_t w[4]) {
  u64 u = (u64)w[0] | (u64)w[1] << 32; int k = 0;
  for (int i = 0; i < 30; i++) k += u > gauss_cdf[i];
  return (w[2] & 1) ? -(i64)k : (i64)k;
}
/* purposes: 0 binary r, 1 / 2 noise of part 0 / 1, 3 key-switch column polynomial, 4 its error, 5 sampleHWt draws, 6 sampleGaussian */
/* the randomness of one Encrypt (FHE-SI.cpp:14-25): small [phim] binary, noise [2][phim] Gaussian samples before the multiplication by p */
void orc_draw_encrypt(const orc_ctx* c, u64 seed, u64 index, i64* small, i64* noise) {
  uint32_t w[4];
  for (i64 j = 0; j < c->phim; j++) {
    draw(seed, index, (uint32_t)j, 0, 0, w); small[j] = w[0] & 1;
    draw(seed, index, (uint32_t)j, 1, 0, w); noise[j] = gaussian_of(w);
    draw(seed, index, (uint32_t)j, 2, 0, w); noise[c->phim + j] = gaussian_of(w);
  }
}
/* the randomness of one key-switch column (FHE-SI.cpp:174-190): a [phim][nlimbs] = SampleRandom(2^logQ) (Util.cpp:49-55: RandomBnd(q) - q/2) from
 * logQ random bits (limb i = words 2 (i mod 2), 2 (i mod 2) + 1 of block i / 2), err [phim] Gaussian */
void orc_draw_keygen(const orc_ctx* c, u64 seed, u64 index, int nlimbs, int logQ, u64* a, i64* err) {
  uint32_t w[4]; int W = nlimbs + 1; u64 x[W], half[W];
  for (i64 j = 0; j < c->phim; j++) {
    memset(x, 0, 8 * W);
    for (int i = 0; i * 64 < logQ; i++) { draw(seed, index, (uint32_t)j, 3, (uint32_t)(i / 2), w); x[i] = (u64)w[2 * (i % 2)] | (u64)w[2 * (i % 2) + 1] << 32; }
    if (logQ % 64) x[logQ / 64] &= (1ull << (logQ % 64)) - 1;                     /* logQ random bits: RandomBnd(2^logQ) */
    memset(half, 0, 8 * W); half[(logQ - 1) / 64] = 1ull << ((logQ - 1) % 64); bn_sub(x, half, W);   /* - q / 2 */
    memcpy(a + j * nlimbs, x, 8 * nlimbs);
    draw(seed, index, (uint32_t)j, 4, 0, w); err[j] = gaussian_of(w);
  }
}
/* sampleHWt (NumbTh.cpp:340-360): draws t = 0, 1, ... pick a position and a sign; positions already set are skipped */
void orc_draw_hwt(const orc_ctx* c, u64 seed, u64 index, i64 hwt, i64* poly) {
  uint32_t w[4]; i64 n = c->phim; memset(poly, 0, 8 * n); if (hwt > n) hwt = n;
  uint32_t t = 0;
  for (i64 i = 0; i < hwt; t++) { draw(seed, index, t, 5, 0, w); u64 u = ((u64)w[0] | (u64)w[1] << 32) % (u64)n; if (!poly[u]) { poly[u] = (w[2] & 1) ? 1 : -1; i++; } }
}
void orc_draw_gaussian(const orc_ctx* c, u64 seed, u64 index, i64* poly) { uint32_t w[4]; for (i64 j = 0; j < c->phim; j++) { draw(seed, index, (uint32_t)j, 6, 0, w); poly[j] = gaussian_of(w); } }

/* ------------------------------------------------------------------ Encrypt / Decrypt with explicit randomness */
/* FHESIPubKey::Encrypt (FHE-SI.cpp:10-36).  pk: [2][L][phim] rows; small: the binary polynomial r (:14-18); noise: [2][phim]
 * Gaussian samples before the multiplication by p (:24-25); msg: [phim] in [0,p).  out: [2][phim][nlimbs] */
void orc_encrypt(const orc_ctx* c, const u64* pk, const i64* small, const i64* noise, const i64* msg, int logQ, u64 p, u64* out, int nlimbs) {
  i64 n = c->phim; i64 rs = (i64)c->L * n; u64* r = malloc(8 * rs); u64* e = malloc(8 * rs); u64* ct = malloc(8 * rs); u64 pl[1] = {p};
  int W = c->L + 3; u64* big = malloc(8 * n * W);
  orc_dcrt_from_poly(c, (const u64*)small, 1, n, r);
  /* delta = floor(2^logQ / p) (:31) as a W-limb integer: schoolbook division of 2^logQ by the word p */
  u64 delta[W]; memset(delta, 0, sizeof(delta)); { u128 rem = 0; for (int i = W - 1; i >= 0; i--) { u64 limb = (i == logQ / 64) ? (1ull << (logQ % 64)) : 0; u128 cur = (rem << 64) | limb; delta[i] = (u64)(cur / p); rem = cur % p; } }
  for (int i = 0; i < 2; i++) {
    orc_dcrt_from_poly(c, (const u64*)(noise + (i64)i * n), 1, n, e); orc_dcrt_op_scalar(c, e, pl, 1, 2);                 /* e *= p */
    memcpy(ct, pk + (i64)i * rs, 8 * rs); orc_dcrt_op(c, ct, r, 2); orc_dcrt_op(c, ct, e, 0);                              /* pk[i]*r + e (:26-27) */
    orc_dcrt_to_poly(c, ct, NULL, 0, 0, big, W);
    for (i64 j = 0; j < n; j++) { u64* v = big + j * W;
      if (i == 0) bn_addmul_i64(v, delta, msg[j], W);                                                                      /* += delta * msg (:32-33) */
      reduce_logq(v, W, logQ, 0); bn_copy_ext(out + ((i64)i * n + j) * nlimbs, nlimbs, v, W); }
  }
  free(r); free(e); free(ct); free(big);
}
/* FHESISecKey::Decrypt (FHE-SI.cpp:93-119) of a 2-part ciphertext: z = c0 + c1*t, m = round(p z / q) mod p with
 * floor((2 p z + q) / (2 q)).  t_rows: [L][phim] = sKeys[1]; parts: [2][phim][nlimbs]; msg_out: [phim] in [0,p) */
void orc_decrypt(const orc_ctx* c, const u64* t_rows, const u64* parts, int nlimbs, int logQ, u64 p, i64* msg_out) {
  i64 n = c->phim; i64 rs = (i64)c->L * n; u64* c0 = malloc(8 * rs); u64* c1 = malloc(8 * rs);
  orc_dcrt_from_poly(c, parts, nlimbs, n, c0); orc_dcrt_from_poly(c, parts + n * nlimbs, nlimbs, n, c1);
  orc_dcrt_op(c, c1, t_rows, 2); orc_dcrt_op(c, c0, c1, 0);                  /* DotProduct with (1, t) (:105-107) */
  int W = c->L + 4; u64* big = malloc(8 * n * W); orc_dcrt_to_poly(c, c0, NULL, 0, 0, big, W);
  u64 q[W]; memset(q, 0, sizeof(q)); q[logQ / 64] = 1ull << (logQ % 64);
  for (i64 j = 0; j < n; j++) { u64* v = big + j * W; int s = bn_sign(v, W); if (s) bn_neg(v, W); bn_mul_u64(v, 2 * p, W); if (s) bn_neg(v, W);   /* 2 p z */
    bn_add(v, q, W); bn_sar(v, W, logQ + 1);                                 /* floor((2pz + q) / 2q) */
    msg_out[j] = (i64)bn_mod_u64(v, W, p); }
  free(c0); free(c1); free(big);
}

/* KeySwitchSI::Init (FHE-SI.cpp:153-209) with the randomness made explicit.  src: [nsrc][L][phim] rows of the source key components,
 * t_rows: rows of dst[1]; a: [ncol][phim][nlimbs] the SampleRandom polynomials (:174-175), err: [ncol][phim] the Gaussian errors
 * (:189-190), column ind = i * ndigits + j.  ksm: [2][ncol][L][phim], ksm[0] = b, ksm[1] = A (:206-208). */
void orc_keyswitch_init(const orc_ctx* c, const u64* src, int nsrc, const u64* t_rows, int logQ, int decomp_bytes, const u64* a, int nlimbs, const i64* err, u64* ksm) {
  i64 n = c->phim; int L = c->L, nd = (logQ + 8 * decomp_bytes - 1) / (8 * decomp_bytes), ncol = nsrc * nd;
  int W = L + 3 + (nd * 8 * decomp_bytes + 63) / 64;                 /* sCoeff grows by 8 decompSize bits per digit (:198-200) */
  u64* sco = malloc(8 * n * W); u64* bco = malloc(8 * n * W); u64* tmp = malloc(8 * (i64)L * n); u64 e[W];
  for (int i = 0; i < nsrc; i++) {
    orc_dcrt_to_poly(c, src + (i64)i * L * n, NULL, 0, 0, sco, W);                                   /* s[i].toPoly(sCoeff[i]) :163-166 */
    for (int j = 0; j < nd; j++) { int ind = i * nd + j;
      u64* A = ksm + ((i64)ncol + ind) * L * n; u64* B = ksm + (i64)ind * L * n;
      orc_dcrt_from_poly(c, a + (i64)ind * n * nlimbs, nlimbs, n, A);                                /* A[ind] = DoubleCRT(poly) :176-178 */
      memcpy(tmp, A, 8 * (i64)L * n); orc_dcrt_op(c, tmp, t_rows, 2);                                /* b[ind] = A[ind]; b[ind] *= t :179,184 */
      orc_dcrt_to_poly(c, tmp, NULL, 0, 0, bco, W);                                                  /* b[ind].toPoly(bCoeff) :186-187 */
      for (i64 k = 0; k < n; k++) { u64* v = bco + k * W; bn_set_i64(e, err[(i64)ind * n + k], W); bn_add(v, e, W); bn_add(v, sco + k * W, W);    /* :192-194 */
        bn_shl(sco + k * W, W, 8 * decomp_bytes);                                                    /* sCoeff[i].rep[k] <<= 8 decompSize :196-198 */
        reduce_logq(v, W, logQ, 0); }                                                                /* ReduceCoefficients :200 */
      orc_dcrt_from_poly(c, bco, W, n, B);                                                           /* b[ind] = DoubleCRT(bCoeff) :201 */
      u64 m1[1] = {~0ull}; orc_dcrt_op_scalar(c, A, m1, 1, 2);                                       /* A[ind] *= -1 :181 */
    }
  }
  free(sco); free(bco); free(tmp);
}

