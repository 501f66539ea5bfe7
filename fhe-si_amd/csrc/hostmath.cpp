// hostmath.cpp -- host-side number theory for table construction (setup only; nothing here is on the hot path).
// Mirrors what FHEcontext / PAlgebra / Cmodulus compute at setup in the reference:
//   PAlgebra::init (PAlgebra.cpp:40-56), Cyclotomic (NumbTh.cpp:142-158), ProbPrime check (FHEContext.cpp:34).
#include "fhesi_internal.h"

#include <cstdarg>
#include <cstring>

static thread_local char g_err[512] = "";
void fhesi_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* fhesi_last_error(void) { return g_err; }

namespace hm {

u64 mulmod(u64 a, u64 b, u64 q) { return (u64)(((u128)a * b) % q); }

u64 powmod(u64 a, u64 e, u64 q) {
  u64 r = 1 % q;
  a %= q;
  for (; e; e >>= 1) {
    if (e & 1) r = mulmod(r, a, q);
    a = mulmod(a, a, q);
  }
  return r;
}

u64 invmod(u64 a, u64 q) { return powmod(a % q, q - 2, q); }

bool is_prime(u64 n) {
  static const u64 bases[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};   // deterministic below 3.3e24
  if (n < 2) return false;
  for (u64 p : bases)
    if (n % p == 0) return n == p;
  u64 d = n - 1;
  int s = 0;
  while (!(d & 1)) { d >>= 1; ++s; }
  for (u64 a : bases) {
    u64 x = powmod(a, d, n);
    if (x == 1 || x == n - 1) continue;
    bool composite = true;
    for (int r = 1; r < s && composite; ++r) {
      x = mulmod(x, x, n);
      if (x == n - 1) composite = false;
    }
    if (composite) return false;
  }
  return true;
}

u64 shoup(u64 w, u64 q) { return (u64)(((u128)w << 64) / q); }
u64 shoup63(u64 w, u64 q) { return (u64)(((u128)w << 63) / q); }

u64 brv(u64 x, int bits) {
  u64 r = 0;
  for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
  return r;
}

int ilog2_ceil(i64 n) {
  int k = 0;
  while ((1ll << k) < n) ++k;
  return k;
}

static u64 gcd(u64 a, u64 b) {
  while (b) { u64 t = a % b; a = b; b = t; }
  return a;
}

std::vector<int> zms_idx(i64 m, i64* phim) {
  std::vector<int> idx(m, -1);
  int k = 0;
  for (i64 i = 0; i < m; ++i)
    if (gcd((u64)i, (u64)m) == 1) idx[i] = k++;
  *phim = k;
  return idx;
}

static int mobius(i64 n) {
  int mu = 1;
  for (i64 p = 2; p * p <= n; ++p) {
    if (n % p == 0) {
      n /= p;
      if (n % p == 0) return 0;
      mu = -mu;
    }
  }
  return n > 1 ? -mu : mu;
}

// Phi_m = prod_{d | m} (X^{m/d} - 1)^{mu(d)}: the factors with mu = 1 are multiplied in, those with mu = -1 divided out ONE AT A TIME --
// both are O(degree) for a binomial X^e - 1 (quotient: Q[i] = Q[i - e] - A[i]), and every partial quotient is exact because the product
// of the mu = -1 binomials divides the numerator.  (A dense division by that product was quadratic: 46 s at m = 2^19, 194 s at the
// largest ring FHEContext.cpp:89 admits.)  Coefficients of cyclotomic polynomials for m <= 2^20 fit easily in 64 bits.
std::vector<i64> cyclotomic(i64 m) {
  std::vector<i64> t(1, 1);
  std::vector<i64> neg;
  for (i64 d = 1; d <= m; ++d) {
    if (m % d) continue;
    const int mu = mobius(d);
    const i64 e = m / d;
    if (mu == 1) {
      std::vector<i64> r(t.size() + e, 0);
      for (size_t i = 0; i < t.size(); ++i) { r[i + e] += t[i]; r[i] -= t[i]; }
      t.swap(r);
    } else if (mu == -1) neg.push_back(e);
  }
  for (const i64 e : neg) {
    std::vector<i64> q(t.size() - (size_t)e);
    for (size_t i = 0; i < q.size(); ++i) q[i] = (i >= (size_t)e ? q[i - e] : 0) - t[i];
    t.swap(q);
  }
  return t;
}

// Psi_m = (X^m - 1) / Phi_m = prod_{mu(d) = -1} (X^{m/d} - 1) / prod_{mu(d) = 1, d > 1} (X^{m/d} - 1), degree m - phi(m): the same
// binomial products and exact binomial divisions.  1 / Phi_m = -Psi_m (1 + X^m + X^2m + ...) as a power series, which is what makes the
// quotient of a division by Phi_m one product with Psi_m (bluestein.hip, the reduction modulo Phi_m for general m).
std::vector<i64> cyclotomic_cofactor(i64 m) {
  std::vector<i64> t(1, 1);
  std::vector<i64> pos;
  for (i64 d = 1; d <= m; ++d) {
    if (m % d) continue;
    const int mu = mobius(d);
    const i64 e = m / d;
    if (mu == -1) {
      std::vector<i64> r(t.size() + e, 0);
      for (size_t i = 0; i < t.size(); ++i) { r[i + e] += t[i]; r[i] -= t[i]; }
      t.swap(r);
    } else if (mu == 1 && d > 1) pos.push_back(e);
  }
  for (const i64 e : pos) {
    std::vector<i64> q(t.size() - (size_t)e);
    for (size_t i = 0; i < q.size(); ++i) q[i] = (i >= (size_t)e ? q[i - e] : 0) - t[i];
    t.swap(q);
  }
  return t;
}

bool is_primitive_2m_root(u64 root, i64 m, u64 q) {
  if (root == 0 || root >= q) return false;
  // order divides 2m; it is exactly 2m iff root^(2m/f) != 1 for every prime f | 2m
  if (powmod(root, 2 * (u64)m, q) != 1) return false;
  u64 e = 2 * (u64)m, t = e;
  for (u64 f = 2; f * f <= t; ++f) {
    if (t % f == 0) {
      if (powmod(root, e / f, q) == 1) return false;
      while (t % f == 0) t /= f;
    }
  }
  if (t > 1 && powmod(root, e / t, q) == 1) return false;
  return true;
}

u64 bn_mod(const u64* limbs, int nlimbs, u64 q) {
  bool neg = limbs[nlimbs - 1] >> 63;
  u64 r = 0;
  if (!neg) {
    for (int i = nlimbs - 1; i >= 0; --i) r = (u64)((((u128)r << 64) | limbs[i]) % q);
    return r;
  }
  // magnitude = ~x + 1
  std::vector<u64> mag(limbs, limbs + nlimbs);
  u64 c = 1;
  for (int i = 0; i < nlimbs; ++i) { u64 v = ~mag[i] + c; c = (c && v == 0); mag[i] = v; }
  for (int i = nlimbs - 1; i >= 0; --i) r = (u64)((((u128)r << 64) | mag[i]) % q);
  return r ? q - r : 0;
}

std::vector<u64> bn_mul_small(const std::vector<u64>& a, u64 b) {
  std::vector<u64> r(a.size() + 1, 0);
  u64 c = 0;
  for (size_t i = 0; i < a.size(); ++i) {
    u128 s = (u128)a[i] * b + c;
    r[i] = (u64)s;
    c = (u64)(s >> 64);
  }
  r[a.size()] = c;
  while (r.size() > 1 && r.back() == 0) r.pop_back();
  return r;
}

}  // namespace hm
