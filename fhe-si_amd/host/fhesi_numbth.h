// fhesi_numbth.h -- part of the C++ mirror of the reference's class surface (see fhesi_host.h, which includes the parts in order; not a
// standalone header): the documented PRNG and the number theory of NumbTh.cpp (mcMod, PowerMod-style helpers, primitive roots, FindM-era utilities) on zz.h.
#pragma once

namespace fhesi {

// ---------------------------------------------------------------- PRNG (replaces srand48 / SetSeed / RandomBnd / lrand48)
class SplitMix64 {
  uint64_t s;
 public:
  explicit SplitMix64(uint64_t seed = 0) : s(seed) {}
  void seed(uint64_t v) { s = v; }
  uint64_t next() { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
  ZZ bits(long nbits) { ZZ r; long words = (nbits + 63) / 64; r.mag.resize(words); for (long i = 0; i < words; ++i) r.mag[i] = next(); if (nbits % 64) r.mag[words - 1] &= (1ull << (nbits % 64)) - 1; r.trim(); return r; }
  ZZ bnd(const ZZ& n) { if (n <= ZZ(1L)) return ZZ(); long k = (n - ZZ(1L)).bits(); for (;;) { ZZ v = bits(k); if (v < n) return v; } }
  long bnd(long n) { return bnd(ZZ(n)).to_long(); }
};
inline SplitMix64& global_rng() { static SplitMix64 g(0); return g; }
inline void SetSeed(uint64_t seed) { global_rng().seed(seed); }
inline ZZ RandomBnd(const ZZ& n) { return global_rng().bnd(n); }
inline long RandomBnd(long n) { return global_rng().bnd(n); }

// ---------------------------------------------------------------- number theory (NumbTh.cpp)
inline uint64_t MulMod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((unsigned __int128)a * b) % q); }
inline uint64_t PowerMod(uint64_t a, uint64_t e, uint64_t q) { uint64_t r = 1 % q; a %= q; for (; e; e >>= 1) { if (e & 1) r = MulMod(r, a, q); a = MulMod(a, a, q); } return r; }
inline bool ProbPrime(uint64_t n) {
  static const uint64_t b[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
  if (n < 2) return false;
  for (uint64_t p : b) if (n % p == 0) return n == p;
  uint64_t d = n - 1; int s = 0; while (!(d & 1)) { d >>= 1; ++s; }
  for (uint64_t a : b) { uint64_t x = PowerMod(a, d, n); if (x == 1 || x == n - 1) continue; bool comp = true; for (int r = 1; r < s && comp; ++r) { x = MulMod(x, x, n); if (x == n - 1) comp = false; } if (comp) return false; }
  return true;
}
// FindPrimitiveRoot (NumbTh.cpp:85-118): the reference tries random bases; this mirror takes the smallest base that
// passes the same order test so that row values are reproducible.
inline long FindPrimitiveRoot(long q, unsigned long e) {
  if ((q - 1) % e) return 0;
  std::vector<unsigned long> facts; unsigned long t = e;
  for (unsigned long f = 2; f * f <= t; ++f) if (t % f == 0) { facts.push_back(f); while (t % f == 0) t /= f; }
  if (t > 1) facts.push_back(t);
  for (uint64_t s = 2; s < 1000; ++s) {
    uint64_t r = PowerMod(s, (q - 1) / e, q);
    if (PowerMod(r, e, q) != 1) continue;
    bool ok = true; for (unsigned long f : facts) if (PowerMod(r, e / f, q) == 1) ok = false;
    if (ok) return (long)r;
  }
  Error("FindPrimitiveRoot(): gave up after 1000 trials");
}

// PolyRed (NumbTh.cpp:209-232): coefficients modulo q into (-q/2, q/2] (q = 2: the sign of the input is kept), or [0, q) with abs
inline void PolyRed(ZZX& out, const ZZX& in, const ZZ& q, bool abs = false) {
  ZZX r; r.rep.resize(in.rep.size()); const ZZ q2 = q >> 1, two(2L);
  for (size_t i = 0; i < in.rep.size(); ++i) {
    ZZ c = in.rep[i] % q;                                     // non-negative, like NTL's % for a positive modulus
    if (!abs) { if (q != two) { if (c > q2) c -= q; } else if (in.rep[i].neg && !c.is_zero()) c = ZZ(-1L); }
    r.rep[i] = c;
  }
  r.normalize(); out = r;
}
inline void PolyRed(ZZX& out, const ZZX& in, int q, bool abs = false) { PolyRed(out, in, ZZ((long)q), abs); }
inline void PolyRed(ZZX& F, int q, bool abs = false) { PolyRed(F, F, q, abs); }
inline void PolyRed(ZZX& F, const ZZ& q, bool abs = false) { PolyRed(F, F, q, abs); }

}  // namespace fhesi
