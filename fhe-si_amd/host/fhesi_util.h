// fhesi_util.h -- part of the C++ mirror of the reference's class surface (see fhesi_host.h, which includes the parts in order; not a
// standalone header): the samplers of NumbTh.cpp:340-404 on the documented PRNG and Util.h / Util.cpp (Reduce, ReduceCoefficients, TensorProduct-era helpers, DotProduct).
#pragma once

namespace fhesi {

// ---------------------------------------------------------------- samplers (NumbTh.cpp:340-404) on the documented PRNG
inline void sampleHWt(ZZX& poly, long Hwt, long n) {
  poly.rep.assign(n, ZZ()); if (Hwt > n) Hwt = n; long i = 0;
  while (i < Hwt) { long u = RandomBnd(n); if (poly.rep[u].is_zero()) { long b = (long)(global_rng().next() & 2) - 1; poly.rep[u] = ZZ(b); ++i; } }
  poly.normalize();
}
inline void sampleSmall(ZZX& poly, long n) {                          // NumbTh.cpp:361-375: 0 with probability 1/2, else +-1
  poly.rep.assign(n, ZZ());
  for (long i = 0; i < n; ++i) { const uint64_t u = global_rng().next(); if (u & 1) poly.rep[i] = ZZ((long)(u & 2) - 1); }
  poly.normalize();
}
inline void sampleGaussian(ZZX& poly, long n, double stdev) {
  static const double Pi = 4.0 * std::atan(1.0); static const long bignum = 0xfffffff;
  poly.rep.assign(n, ZZ());
  for (long i = 0; i < n; i += 2) {
    double r1 = (1 + RandomBnd(bignum)) / ((double)bignum + 1), r2 = (1 + RandomBnd(bignum)) / ((double)bignum + 1);
    double theta = 2 * Pi * r1, rr = std::sqrt(-2.0 * std::log(r2)) * stdev;
    poly.rep[i] = ZZ((long)std::floor(rr * std::cos(theta) + 0.5));
    if (i + 1 < n) poly.rep[i + 1] = ZZ((long)std::floor(rr * std::sin(theta) + 0.5));
  }
  poly.normalize();
}
inline void SampleRandom(ZZX& poly, const ZZ& modulus, unsigned degn) {   // Util.cpp:49-55
  ZZ offset = modulus / ZZ(2L); poly.rep.assign(degn, ZZ());
  for (unsigned i = 0; i < degn; ++i) poly.rep[i] = RandomBnd(modulus) - offset;
  poly.normalize();
}
inline void DoubleCRT::sampleSmall() { ZZX p; fhesi::sampleSmall(p, context.zMstar.phiM()); *this = p; }           // DoubleCRT.h:308-311
inline void DoubleCRT::sampleHWt(long Hwt) { ZZX p; fhesi::sampleHWt(p, Hwt, context.zMstar.phiM()); *this = p; }
inline void DoubleCRT::sampleGaussian(double sd) { if (sd == 0.0) sd = context.stdev; ZZX p; fhesi::sampleGaussian(p, context.zMstar.phiM(), sd); *this = p; }

// ---------------------------------------------------------------- Util.cpp
inline void Reduce(ZZ& val, unsigned logQ, bool positive = false) {   // Util.cpp:3-26
  ZZ Q = ZZ(1L) << (long)logQ, r = val % Q;        // canonical residue in [0, 2^logQ)
  if (!positive && r.bit(logQ - 1)) r -= Q;
  val = r;
}
inline void ReduceCoefficients(ZZX& poly, unsigned logQ, bool positive = false) { for (auto& c : poly.rep) Reduce(c, logQ, positive); poly.normalize(); }
template <typename T> void DotProduct(T& res, const std::vector<T>& v1, const std::vector<T>& v2) {   // Util.h:79-98
  if (v1.empty()) return;
  res = v1[0]; res *= v2[0];
  for (size_t i = 1; i < v1.size(); ++i) { T val = v1[i]; val *= v2[i]; res += val; }
}

}  // namespace fhesi
