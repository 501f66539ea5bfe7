// ring_elem.h -- TEST HARNESS: the plaintext side of the Regression / Statistics drivers.  An element of Z_p[X]/Phi_m with the operations
// Matrix<T>, SumBatchedData and the drivers use (+=, *=, *= long, >>= as the automorphism X -> X^k); word arithmetic, p < 2^31.
#pragma once
#include "../../fhe-si_amd/host/fhesi_host.h"

using namespace fhesi;

struct RingElem {
  static const FHEcontext* ctx;
  static std::vector<long> phi;        // Phi_m mod p, degree phi(m), monic
  std::vector<long> c;
  RingElem() : c(ctx ? ctx->zMstar.phiM() : 0, 0) {}
  static long P() { return ctx->ModulusP().to_long(); }
  static void init(const FHEcontext* cx) { ctx = cx; const ZZX& f = cx->zMstar.PhimX(); phi.assign(cx->zMstar.phiM() + 1, 0); for (long i = 0; i <= deg(f); ++i) phi[i] = rem(f.rep[i], P()); }
  static void reduce(std::vector<long>& a) {       // a mod Phi_m, in place; result has phi(m) entries
    const long df = (long)phi.size() - 1, p = P();
    for (long i = (long)a.size() - 1; i >= df; --i) { const long t = a[i]; if (!t) continue; for (long j = 0; j <= df; ++j) a[i - df + j] = (a[i - df + j] + (p - t) * phi[j]) % p; }
    a.resize(df, 0);
  }
  RingElem& operator+=(const RingElem& o) { const long p = P(); for (size_t i = 0; i < c.size(); ++i) c[i] = (c[i] + o.c[i]) % p; return *this; }
  RingElem& operator*=(const RingElem& o) {
    const long p = P(); std::vector<long> r(2 * c.size(), 0);
    for (size_t i = 0; i < c.size(); ++i) if (c[i]) for (size_t j = 0; j < o.c.size(); ++j) r[i + j] = (r[i + j] + c[i] * o.c[j]) % p;
    reduce(r); c = r; return *this;
  }
  RingElem& operator*=(long l) { const long p = P(); l = ((l % p) + p) % p; for (auto& v : c) v = v * l % p; return *this; }
  RingElem& operator>>=(long k) {       // X -> X^k modulo Phi_m
    const long m = ctx->zMstar.M(), p = P(); std::vector<long> r(m, 0);
    for (size_t i = 0; i < c.size(); ++i) { const long e = (long)((i * (unsigned long)k) % m); r[e] = (r[e] + c[i]) % p; }
    reduce(r); c = r; return *this;
  }
  bool operator==(const RingElem& o) const { return c == o.c; }
};
inline const FHEcontext* RingElem::ctx = nullptr;
inline std::vector<long> RingElem::phi;

