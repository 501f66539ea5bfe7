// host_helpers_literal.h -- TEST HARNESS ONLY.  Restatements of host helpers of the reference that nothing on the hot path calls
// (SURVEY section 2 marks them out of scope): NumbTh.h:43-66,127-131,202 / NumbTh.cpp:20-200,421-429 (factorize, phi_N, mobius, ord, primroot,
// Cyclotomic, largestCoeff, argmax / argmin) and Util.h:68-76,100-111 / Util.cpp:33-43 (ComputeLog, TensorProduct, ReduceCoefficientsSlow).
// They lived in fhe-si_amd/host/ until round 5; tests/host/test_wire.cpp is their only user (self-checks of the host arithmetic).
#pragma once
#include "../../fhe-si_amd/host/fhesi_host.h"
namespace fhesi {
// the small number-theory helpers PAlgebra and the drivers use (NumbTh.h:43-66,202; NumbTh.cpp:20-200,421-429), on machine words
inline void factorize(std::vector<long>& factors, long N) { factors.clear(); for (long f = 2; f * f <= N; ++f) if (N % f == 0) { factors.push_back(f); while (N % f == 0) N /= f; } if (N > 1) factors.push_back(N); }   // distinct primes, ascending
inline int phi_N(int N) { std::vector<long> f; factorize(f, N); long r = N; for (long q : f) r = r / q * (q - 1); return (int)r; }
inline int mobius(int n) { int r = 1; for (int f = 2; f * f <= n; ++f) if (n % f == 0) { n /= f; if (n % f == 0) return 0; r = -r; } return n > 1 ? -r : r; }
inline int ord(int N, int p) { int o = 0; while (N % p == 0) { ++o; N /= p; } return o; }                                       // the exponent of p in N
inline int primroot(int N, int phiN) {                                                                                          // smallest g >= 2 whose order modulo N is phiN
  std::vector<long> f; factorize(f, phiN);
  for (int g = 2;; ++g) { bool ok = true; for (long q : f) if (PowerMod((uint64_t)g, (uint64_t)(phiN / q), (uint64_t)N) == 1) { ok = false; break; } if (ok) return g; }
}
inline ZZX Cyclotomic(int N) {                                                                                                  // Phi_N = prod_{d | N} (X^(N/d) - 1)^mu(d), exact divisions on machine words
  std::vector<long> num{1}, den{1};
  auto times = [](std::vector<long>& a, int e) { std::vector<long> r(a.size() + e, 0); for (size_t i = 0; i < a.size(); ++i) { r[i + e] += a[i]; r[i] -= a[i]; } a.swap(r); };   // a *= (X^e - 1)
  for (int d = 1; d <= N; ++d) if (N % d == 0) { const int mu = mobius(d); if (mu == 1) times(num, N / d); else if (mu == -1) times(den, N / d); }
  std::vector<long> q(num.size() - den.size() + 1, 0);                                                                         // den is monic
  for (long i = (long)q.size() - 1; i >= 0; --i) { q[i] = num[i + den.size() - 1]; for (size_t j = 0; j < den.size(); ++j) num[i + j] -= q[i] * den[j]; }
  ZZX F; F.rep.resize(q.size()); for (size_t i = 0; i < q.size(); ++i) F.rep[i] = ZZ(q[i]); F.normalize();
  return F;
}
inline ZZ largestCoeff(const ZZX& f) { ZZ mx; for (auto& c : f.rep) { ZZ a = c; a.neg = false; if (mx < a) mx = a; } return mx; }
template <class T> long argmax(std::vector<T>& v) { if (v.empty()) return -1; long b = 0; for (size_t i = 1; i < v.size(); ++i) if (v[b] < v[i]) b = (long)i; return b; }   // NumbTh.h:127-131
template <class T> long argmin(std::vector<T>& v) { if (v.empty()) return -1; long b = 0; for (size_t i = 1; i < v.size(); ++i) if (v[i] < v[b]) b = (long)i; return b; }
inline void ReduceCoefficientsSlow(ZZX& poly, const ZZ& modulus, bool positive = false) {   // Util.cpp:33-43: any modulus; NTL's % is non-negative for a positive modulus
  const ZZ half = modulus / ZZ(2L);
  for (auto& c : poly.rep) { c = c % modulus; if (!positive && c > half) c -= modulus; }
  poly.normalize();
}
inline void ReduceCoefficientsSlow(ZZX& poly, unsigned modulus, bool positive = false) { ReduceCoefficientsSlow(poly, ZZ((unsigned long)modulus), positive); }
template <typename T> unsigned ComputeLog(T val) { unsigned lg = 0; while (val != 0) { val >>= 1; ++lg; } return lg - 1; }            // Util.h:68-76
template <typename T> void TensorProduct(std::vector<T>& res, const std::vector<T>& v1, const std::vector<T>& v2) {                // Util.h:100-111
  res.resize(v1.size() * v2.size());
  size_t ind = 0;
  for (size_t i = 0; i < v1.size(); ++i) for (size_t j = 0; j < v2.size(); ++j) { res[ind] = v1[i]; res[ind++] *= v2[j]; }
}
}  // namespace fhesi
