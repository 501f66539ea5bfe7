// fhesi_host.h -- C++ mirror of the reference's class surface for the DoubleCRT path, with every body bound to the
// C ABI of include/fhesi_hip.h (the HIP library).  Same class and method names, argument meaning and error behaviour
// (NTL-style Error(msg) -> abort) as the reference so that code written against
//   PAlgebra (PAlgebra.h:53-88), IndexSet (IndexSet.h:26-127), Cmodulus (CModulus.h:42-170), FHEcontext (FHEContext.h:40-205),
//   DoubleCRT (DoubleCRT.h:83-365), CiphertextPart / Ciphertext (Ciphertext.h:10-97), FHESISecKey / FHESIPubKey /
//   KeySwitchSI (FHE-SI.h), Reduce / ReduceCoefficients / DotProduct (Util.h)
// reads the same.  NTL is not available here, so ZZ / ZZX come from zz.h and randomness from a documented SplitMix64
// stream (NTL's SetSeed/RandomBnd and lrand48 streams are not reproducible anyway).
// Rows live in HBM behind fhesi_dcrt handles; getMap()-style access materialises them on demand.
#pragma once
#include <cassert>
#include <cmath>
#include <cstdlib>
#include <map>
#include <memory>
#include <set>
#include <atomic>
#include <vector>

#include "../../include/fhesi_hip.h"
#include "zz.h"

namespace fhesi {

typedef std::vector<long> vec_long;

inline void ck(int rc) { if (rc) Error(fhesi_last_error()); }

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
template <class T> long argmax(std::vector<T>& v) { if (v.empty()) return -1; long b = 0; for (size_t i = 1; i < v.size(); ++i) if (v[b] < v[i]) b = (long)i; return b; }   // NumbTh.h:127-131
template <class T> long argmin(std::vector<T>& v) { if (v.empty()) return -1; long b = 0; for (size_t i = 1; i < v.size(); ++i) if (v[i] < v[b]) b = (long)i; return b; }

// ---------------------------------------------------------------- IndexSet (IndexSet.h:26-127), ordered set of prime indices
class IndexSet {
  std::set<long> s;
 public:
  IndexSet() {}
  IndexSet(long lo, long hi) { for (long i = lo; i <= hi; ++i) s.insert(i); }
  explicit IndexSet(long j) { s.insert(j); }
  static const IndexSet& emptySet() { static IndexSet e; return e; }
  long first() const { return s.empty() ? 0 : *s.begin(); }
  long last() const { return s.empty() ? -1 : *s.rbegin(); }
  long next(long j) const { auto it = s.upper_bound(j); return it == s.end() ? last() + 1 : *it; }
  long card() const { return (long)s.size(); }
  bool contains(long j) const { return s.count(j) != 0; }
  void insert(long j) { s.insert(j); }
  void insert(const IndexSet& o) { s.insert(o.s.begin(), o.s.end()); }
  void remove(long j) { s.erase(j); }
  void remove(const IndexSet& o) { for (long j : o.s) s.erase(j); }
  void retain(const IndexSet& o) { for (auto it = s.begin(); it != s.end();) { if (!o.s.count(*it)) it = s.erase(it); else ++it; } }   // intersection, in place (IndexSet.h:122-123)
  bool disjointFrom(const IndexSet& o) const { for (long j : s) if (o.s.count(j)) return false; return true; }                    // IndexSet.h:94-95
  void clear() { s.clear(); }
  bool operator==(const IndexSet& o) const { return s == o.s; }
  bool operator!=(const IndexSet& o) const { return s != o.s; }
  bool contains(const IndexSet& o) const { for (long j : o.s) if (!s.count(j)) return false; return true; }
  std::vector<int32_t> vec() const { return std::vector<int32_t>(s.begin(), s.end()); }
  friend IndexSet operator|(const IndexSet& a, const IndexSet& b) { IndexSet r = a; r.insert(b); return r; }
  friend IndexSet operator&(const IndexSet& a, const IndexSet& b) { IndexSet r; for (long j : a.s) if (b.s.count(j)) r.s.insert(j); return r; }
  friend IndexSet operator/(const IndexSet& a, const IndexSet& b) { IndexSet r; for (long j : a.s) if (!b.s.count(j)) r.s.insert(j); return r; }   // set minus
  friend bool operator>=(const IndexSet& a, const IndexSet& b) { return a.contains(b); }
  friend bool operator>(const IndexSet& a, const IndexSet& b) { return a.contains(b) && a != b; }
};
inline long card(const IndexSet& s) { return s.card(); }
inline bool disjoint(const IndexSet& a, const IndexSet& b) { return a.disjointFrom(b); }

// ---------------------------------------------------------------- PAlgebra (PAlgebra.h:53-88)
class PAlgebra {
  unsigned m = 0, g = 0, phim = 0;
  ZZX Phi_mX;
  std::vector<long> zmsIdx;
 public:
  void init(unsigned mm, unsigned gg, const std::vector<int32_t>& idx, const std::vector<int64_t>& phi) {
    m = mm; g = gg; zmsIdx.assign(idx.begin(), idx.end()); phim = 0;
    for (long v : zmsIdx) if (v >= 0) ++phim;
    Phi_mX.rep.clear(); for (int64_t c : phi) Phi_mX.rep.push_back(ZZ((long)c)); Phi_mX.normalize();
  }
  unsigned M() const { return m; }
  unsigned G() const { return g; }
  unsigned phiM() const { return phim; }
  const ZZX& PhimX() const { return Phi_mX; }
  int indexInZmstar(unsigned t) const { return (t > 0 && t < m) ? (int)zmsIdx[t] : -1; }
  bool inZmStar(unsigned t) const { return t > 0 && t < m && zmsIdx[t] > -1; }
};

class FHEcontext;
inline void drop_ct_engine(const FHEcontext*);   // fhesi_engine.h: the arena of device-resident ciphertexts goes before the device context does

// ---------------------------------------------------------------- Cmodulus (CModulus.h:42-170)
class Cmodulus {
  const FHEcontext* ctx;
  long q, root;
  int index;
 public:
  Cmodulus(const FHEcontext* c, long qq, long rt, int idx) : ctx(c), q(qq), root(rt), index(idx) {}
  const long& getQ() const { return q; }
  const long& getRoot() const { return root; }
  void FFT(vec_long& y, const ZZX& x) const;    // y = FFT(x)       (CModulus.cpp:90-107)
  void iFFT(ZZX& x, const vec_long& y) const;   // x = FFT^{-1}(y)  (CModulus.cpp:110-132)
};

// ---------------------------------------------------------------- FHEcontext (FHEContext.h:40-205, FHEContext.cpp)
class FHEcontext {
  std::vector<Cmodulus> moduli;
  mutable fhesi_ctx* dev = nullptr;
  ZZ ptxtP;
  unsigned generator = 0;
  int device;
 public:
  PAlgebra zMstar;
  IndexSet ctxtPrimes, specialPrimes;
  double stdev = 3.2;
  int spNbits = 60;                    // NTL_SP_NBITS of the NTL build being mirrored: where AddPrimesBySize starts (FHEContext.cpp:92); 50 in NTL 5.x / 6.x, 60 today
  ZZ modulusQ;
  unsigned logQ = 0, decompSize = 3, ndigits = 0;

  FHEcontext(unsigned m, unsigned logQ_, unsigned p, unsigned gen, unsigned decomp = 3, int device_ = 0) : device(device_) { Init(m, logQ_, ZZ((long)p), gen, decomp); }
  ~FHEcontext() { drop_ct_engine(this); if (dev) fhesi_ctx_destroy(dev); }
  FHEcontext(const FHEcontext&) = delete;
  void Init(unsigned m, unsigned logQ_, const ZZ& p, unsigned gen, unsigned decomp = 3) {   // FHEContext.h:105-118
    m_ = m; logQ = logQ_; modulusQ = ZZ(1L) << (long)logQ_; decompSize = decomp;
    ndigits = (logQ_ + 8 * decomp - 1) / (8 * decomp);
    ptxtP = p; generator = gen;
    // PAlgebra tables: phi(m) and zmsIdx are needed before any prime exists (SetUpSIContext sizes the chain with phi(m))
    std::vector<int32_t> idx(m, -1); int k = 0;
    for (unsigned i = 0; i < m; ++i) { unsigned a = i, b = m; while (b) { unsigned t2 = a % b; a = b; b = t2; } if (a == 1) idx[i] = k++; }
    zMstar.init(m, gen, idx, std::vector<int64_t>());
  }
  unsigned Generator() const { return generator; }
  const ZZ& ModulusP() const { return ptxtP; }
  long ithPrime(unsigned i) const { return i < moduli.size() ? moduli[i].getQ() : 0; }
  const Cmodulus& ithModulus(unsigned i) const { return moduli[i]; }
  long numPrimes() const { return (long)moduli.size(); }
  bool inChain(long p) const { for (auto& c : moduli) if (c.getQ() == p) return true; return false; }
  ZZ productOfPrimes(const IndexSet& s) const { ZZ p(1L); for (long i = s.first(); i <= s.last(); i = s.next(i)) p *= ZZ(ithPrime(i)); return p; }
  ZZ productOfPrimes() const { return productOfPrimes(ctxtPrimes); }
  double logOfPrime(unsigned i) const { return std::log((double)ithPrime(i)); }                                                   // FHEContext.h:178
  double logOfProduct(const IndexSet& s) const {                                                                                  // FHEContext.h:181-189
    if (s.last() >= numPrimes()) Error("FHEContext::logOfProduct: IndexSet has too many rows");
    double ans = 0.0; for (long i = s.first(); i <= s.last(); i = s.next(i)) ans += logOfPrime((unsigned)i); return ans;
  }
  bool isZeroDivisor(const ZZ& num) const { for (auto& c : moduli) if (rem(num, c.getQ()) == 0) return true; return false; }      // FHEContext.h:152-156

  void AddPrime(long p, bool special, long root = 0) {   // FHEContext.cpp:30-43
    if (dev) Error("FHEcontext::AddPrime: the chain is already bound to the device");
    long twoM = 2 * (long)zMstar.M();
    if (!(ProbPrime(p) && p % twoM == 1 && !inChain(p))) Error("FHEcontext::AddPrime: assertion ProbPrime(p) && p % twoM == 1 && !inChain(p) failed");
    if (!root) root = FindPrimitiveRoot(p, (unsigned long)twoM);       // CModulus.cpp:69-76
    long i = (long)moduli.size();
    moduli.push_back(Cmodulus(this, p, root, (int)i));
    if (special) specialPrimes.insert(i); else ctxtPrimes.insert(i);
  }
  void SetUpSIContext(long xi = 1) {   // FHEContext.cpp:83-85
    AddPrimesBySize(log(modulusQ) * 2 + log(ModulusP()) + std::log((double)zMstar.phiM()) * 2 + std::log(2.0) + std::log((double)xi), false, spNbits);
  }
  double AddPrimesBySize(double totalSize, bool special, int sp_nbits = 60) {   // FHEContext.cpp:88-115
    if (!zMstar.M() || zMstar.M() > (1u << 20)) Error("AddModuli1: m undefined or larger than 2^20");
    long p = (long)((1ull << sp_nbits) - 1), twoM = 2 * (long)zMstar.M();
    p -= p % twoM; p += twoM + 1;
    bool lastPrime = false; double sizeLeft = totalSize;
    while (sizeLeft > 0.0) {
      if (sizeLeft < std::log((double)p) && !lastPrime) { lastPrime = true; p = (long)std::ceil(std::exp(sizeLeft)); p -= (p % twoM) - 1; twoM = -twoM; }
      do { p -= twoM; } while (!ProbPrime(p));
      if (!inChain(p)) { AddPrime(p, special); sizeLeft -= std::log((double)p); }
    }
    return totalSize - sizeLeft;
  }
  // nPrimes primes = 1 mod 2m ASCENDING from p (FHEContext.cpp:118-141); returns the natural log of their product
  double AddPrimesByNumber(long nPrimes, long p = 1, bool special = false) {
    if (!zMstar.M() || zMstar.M() > (1u << 20)) Error("FHEcontext::AddModuli2: m undefined or larger than 2^20");
    const long twoM = 2 * (long)zMstar.M();
    if (p < 1) p = 1;
    p -= (p % twoM) - 1;
    double sizeSoFar = 0.0;
    while (nPrimes > 0) {
      do { p += twoM; } while (!ProbPrime((uint64_t)p));
      if (!inChain(p)) { AddPrime(p, special); --nPrimes; sizeSoFar += std::log((double)p); }
    }
    return sizeSoFar;
  }
  // the device context is created on first use, from the finished chain
  fhesi_ctx* handle() const {
    if (!dev) {
      std::vector<uint64_t> q, r;
      for (auto& c : moduli) { q.push_back((uint64_t)c.getQ()); r.push_back((uint64_t)c.getRoot()); }
      if (q.empty()) Error("FHEcontext: no primes in the chain");
      ck(fhesi_ctx_create(&dev, m_, (int32_t)q.size(), q.data(), r.data(), device));
      std::vector<int32_t> idx(m_); std::vector<int64_t> phi(fhesi_ctx_phim(dev) + 1);
      ck(fhesi_ctx_zms_idx(dev, idx.data())); ck(fhesi_ctx_phi_m(dev, phi.data()));
      const_cast<PAlgebra&>(zMstar).init(m_, generator, idx, phi);
    }
    return dev;
  }
  // another device context with the same chain and roots on GPU `dev_index` (one per GPU in the multi-GPU model); the caller owns it
  fhesi_ctx* replica(int dev_index) const {
    std::vector<uint64_t> q, r;
    for (auto& c : moduli) { q.push_back((uint64_t)c.getQ()); r.push_back((uint64_t)c.getRoot()); }
    fhesi_ctx* h = nullptr;
    ck(fhesi_ctx_create(&h, m_, (int32_t)q.size(), q.data(), r.data(), dev_index));
    ck(fhesi_ctx_copy_options(h, handle()));            // the replica runs the forms selected on this context (checker and A/B switches included)
    return h;
  }
  int deviceIndex() const { return device; }
 private:
  unsigned m_ = 0;
};
extern FHEcontext* activeContext;   // FHEContext.cpp:21
inline double AddPrimesBySize(FHEcontext& c, double totalSize, bool special = false) { return c.AddPrimesBySize(totalSize, special, c.spNbits); }   // FHEContext.h: free functions of the same name
inline double AddPrimesByNumber(FHEcontext& c, long nPrimes, long p = 1, bool special = false) { return c.AddPrimesByNumber(nPrimes, p, special); }

// ---------------------------------------------------------------- ZZX <-> limb buffers
inline int limbs_for(const ZZX& p) { long b = 1; for (auto& c : p.rep) b = std::max(b, c.bits() + 1); return (int)((b + 63) / 64); }
inline std::vector<uint64_t> to_limbs(const ZZX& p, int nl) { std::vector<uint64_t> v(std::max<size_t>(1, p.rep.size()) * nl, 0); for (size_t i = 0; i < p.rep.size(); ++i) p.rep[i].to_limbs(&v[i * nl], nl); return v; }
// n coefficients of p as nl two's complement limbs each (zero above the degree), without a ZZ copy per coefficient; and back
inline void poly_to_limbs(const ZZX& p, uint64_t* dst, long n, int nl) {
  const long have = std::min<long>(n, (long)p.rep.size());
  for (long j = 0; j < have; ++j) p.rep[j].to_limbs(dst + (size_t)j * nl, nl);
  if (have < n) std::fill(dst + (size_t)have * nl, dst + (size_t)n * nl, (uint64_t)0);
}
inline void limbs_to_poly(ZZX& p, const uint64_t* src, long n, int nl) { p.rep.resize(n); for (long j = 0; j < n; ++j) p.rep[j] = ZZ::from_limbs(src + (size_t)j * nl, nl); p.normalize(); }
inline ZZX from_limbs(const std::vector<uint64_t>& v, long n, int nl) { ZZX p; p.rep.resize(n); for (long i = 0; i < n; ++i) p.rep[i] = ZZ::from_limbs(&v[i * nl], nl); p.normalize(); return p; }

inline void Cmodulus::FFT(vec_long& y, const ZZX& x) const {
  fhesi_ctx* h = ctx->handle(); long n = fhesi_ctx_phim(h);
  int nl = limbs_for(x); std::vector<uint64_t> lim = to_limbs(x, nl), out(n);
  ck(fhesi_cmod_fft(h, index, lim.data(), nl, (int64_t)x.rep.size(), out.data()));
  y.assign(out.begin(), out.end());
}
inline void Cmodulus::iFFT(ZZX& x, const vec_long& y) const {
  fhesi_ctx* h = ctx->handle(); long n = fhesi_ctx_phim(h);
  if ((long)y.size() != n) Error("Cmodulus::iFFT: bad row length");
  std::vector<uint64_t> in(y.begin(), y.end()), out(n);
  ck(fhesi_cmod_ifft(h, index, in.data(), out.data()));
  x.rep.assign(n, ZZ()); for (long i = 0; i < n; ++i) x.rep[i] = ZZ((unsigned long)out[i]); x.normalize();
}

// ---------------------------------------------------------------- DoubleCRT (DoubleCRT.h:83-365, DoubleCRT.cpp)
enum { OP_ADD = FHESI_OP_ADD, OP_SUB = FHESI_OP_SUB, OP_MUL = FHESI_OP_MUL, OP_DIV = FHESI_OP_DIV, OP_SET = FHESI_OP_SET };
class SingleCRT;
class DoubleCRT {
  const FHEcontext& context;
  fhesi_dcrt* h = nullptr;
  void alloc(const IndexSet& s) { auto v = s.vec(); if (v.empty()) Error("DoubleCRT: empty index set"); ck(fhesi_dcrt_alloc(context.handle(), v.data(), (int32_t)v.size(), &h)); }
  DoubleCRT& Op(const DoubleCRT& other, int op, bool matchIndexSets = true) {   // DoubleCRT.cpp:79-113
    if (&context != &other.context) Error("DoubleCRT::Op: incompatible objects");
    if (matchIndexSets && !(getIndexSet() >= other.getIndexSet())) addPrimes(other.getIndexSet() / getIndexSet());
    if (getIndexSet() > other.getIndexSet()) { DoubleCRT tmp(other); tmp.addPrimes(getIndexSet() / other.getIndexSet()); ck(fhesi_dcrt_op(h, tmp.h, op)); }
    else if (getIndexSet() == other.getIndexSet()) ck(fhesi_dcrt_op(h, other.h, op));
    else { DoubleCRT tmp(other); tmp.removePrimes(other.getIndexSet() / getIndexSet()); ck(fhesi_dcrt_op(h, tmp.h, op)); }   // !matchIndexSets: this object's set rules
    return *this;
  }
  DoubleCRT& Op(const ZZ& num, int op) { int nl = (int)(num.bits() / 64 + 2); std::vector<uint64_t> v(nl); num.to_limbs(v.data(), nl); ck(fhesi_dcrt_op_scalar(h, v.data(), nl, op)); return *this; }   // :115-129
  DoubleCRT& Op(const ZZX& poly, int op) { DoubleCRT other(poly, context, getIndexSet()); return Op(other, op); }   // :131-137
 public:
  DoubleCRT(const DoubleCRT& o) : context(o.context) { alloc(o.getIndexSet()); ck(fhesi_dcrt_copy(h, o.h)); }
  DoubleCRT(const ZZX& poly, const FHEcontext& c, const IndexSet& s) : context(c) { alloc(s); *this = poly; }
  DoubleCRT(const ZZX& poly, const FHEcontext& c) : context(c) { alloc(c.ctxtPrimes); *this = poly; }
  explicit DoubleCRT(const ZZX& poly) : context(*activeContext) { alloc(context.ctxtPrimes); *this = poly; }
  DoubleCRT(const FHEcontext& c, const IndexSet& s) : context(c) { alloc(s); }
  explicit DoubleCRT(const FHEcontext& c) : context(c) { alloc(c.ctxtPrimes); }
  DoubleCRT() : context(*activeContext) { alloc(context.ctxtPrimes); }
  ~DoubleCRT() { if (h) fhesi_dcrt_free(h); }

  DoubleCRT& operator=(const DoubleCRT& o) { if (&context != &o.context) Error("DoubleCRT assigment: incompatible contexts"); ck(fhesi_dcrt_copy(h, o.h)); return *this; }   // :313-320
  DoubleCRT& operator=(const ZZX& poly) { int nl = limbs_for(poly); auto v = to_limbs(poly, nl); ck(fhesi_dcrt_from_poly(h, v.data(), nl, (int64_t)poly.rep.size())); return *this; }   // :323-331
  DoubleCRT& operator=(const ZZ& num) { return Op(num, OP_SET); }   // :333-347
  DoubleCRT& operator=(long num) { return *this = ZZ(num); }
  DoubleCRT& operator=(const SingleCRT& scrt);                                  // :484-496
  void toSingleCRT(SingleCRT& scrt, const IndexSet& s) const;                    // :498-510
  void toSingleCRT(SingleCRT& scrt) const;                                       // :512-515

  void toPoly(ZZX& p, const IndexSet& s, bool positive = false) const {   // :349-404
    IndexSet s1 = getIndexSet() & s;
    if (card(s1) == 0) { clear(p); return; }
    int nl = (int)card(s1) + 2; long n = context.zMstar.phiM(); auto idx = s1.vec();
    std::vector<uint64_t> out((size_t)n * nl);
    ck(fhesi_dcrt_to_poly(h, idx.data(), (int32_t)idx.size(), positive ? 1 : 0, out.data(), nl));
    p = from_limbs(out, n, nl);
  }
  void toPoly(ZZX& p, bool positive = false) const { toPoly(p, getIndexSet(), positive); }
  bool operator==(const DoubleCRT& o) const { if (&context != &o.context) return false; int32_t eq = 0; ck(fhesi_dcrt_equal(h, o.h, &eq)); return eq != 0; }
  bool operator!=(const DoubleCRT& o) const { return !(*this == o); }
  DoubleCRT& SetZero() { return *this = ZZ(); }
  DoubleCRT& SetOne() { return *this = 1L; }
  void addPrimes(const IndexSet& s1) { auto v = s1.vec(); if (v.empty()) return; ck(fhesi_dcrt_add_primes(h, v.data(), (int32_t)v.size())); }       // :142-156
  void removePrimes(const IndexSet& s1) { auto v = s1.vec(); if (v.empty()) return; ck(fhesi_dcrt_remove_primes(h, v.data(), (int32_t)v.size())); }  // DoubleCRT.h:197-199
  DoubleCRT& operator+=(const DoubleCRT& o) { return Op(o, OP_ADD); }
  DoubleCRT& operator+=(const ZZX& p) { return Op(p, OP_ADD); }
  DoubleCRT& operator+=(const ZZ& n) { return Op(n, OP_ADD); }
  DoubleCRT& operator+=(long n) { return Op(ZZ(n), OP_ADD); }
  DoubleCRT& operator-=(const DoubleCRT& o) { return Op(o, OP_SUB); }
  DoubleCRT& operator-=(const ZZX& p) { return Op(p, OP_SUB); }
  DoubleCRT& operator-=(const ZZ& n) { return Op(n, OP_SUB); }
  DoubleCRT& operator-=(long n) { return Op(ZZ(n), OP_SUB); }
  DoubleCRT& operator*=(const DoubleCRT& o) { return Op(o, OP_MUL); }
  DoubleCRT& operator*=(const ZZX& p) { return Op(p, OP_MUL); }
  DoubleCRT& operator*=(const ZZ& n) { return Op(n, OP_MUL); }
  DoubleCRT& operator*=(long n) { return Op(ZZ(n), OP_MUL); }
  void Add(const DoubleCRT& o, bool match = true) { Op(o, OP_ADD, match); }
  void Sub(const DoubleCRT& o, bool match = true) { Op(o, OP_SUB, match); }
  void Mul(const DoubleCRT& o, bool match = true) { Op(o, OP_MUL, match); }
  DoubleCRT& operator/=(const ZZ& n) { return Op(n, OP_DIV); }   // :407-420
  DoubleCRT& operator/=(long n) { return Op(ZZ(n), OP_DIV); }
  void Exp(long e) { ck(fhesi_dcrt_exp(h, e)); }   // :423-434
  void automorph(long k) { if (!context.zMstar.inZmStar((unsigned)k)) Error("DoubleCRT::automorph: k not in Zm*"); ck(fhesi_dcrt_automorph(h, k)); }   // :439-465
  // BGV-style modulus switching (no callers in fhe-si, kept for the class surface): device kernels behind the C ABI
  double addPrimesAndScale(const IndexSet& s1) {   // DoubleCRT.cpp:162-208
    std::vector<int32_t> v; for (long i = s1.first(); i <= s1.last(); i = s1.next(i)) v.push_back((int32_t)i);
    double lf = 0.0;
    ck(fhesi_dcrt_add_primes_and_scale(h, v.data(), (int32_t)v.size(), (uint64_t)context.ModulusP().to_long(), &lf));
    return lf;
  }
  void scaleDownToSet(const IndexSet& s) {   // DoubleCRT.cpp:518-558
    std::vector<int32_t> v; for (long i = s.first(); i <= s.last(); i = s.next(i)) v.push_back((int32_t)i);
    ck(fhesi_dcrt_scale_down_to_set(h, v.data(), (int32_t)v.size(), (uint64_t)context.ModulusP().to_long()));
  }
  DoubleCRT& operator>>=(long k) { automorph(k); return *this; }
  const FHEcontext& getContext() const { return context; }
  IndexSet getIndexSet() const { int32_t n = 0; std::vector<int32_t> v(64); ck(fhesi_dcrt_index_set(h, v.data(), &n)); IndexSet s; for (int i = 0; i < n; ++i) s.insert(v[i]); return s; }
  // getMap(): rows materialised from HBM (the reference's IndexMap<vec_long>, DoubleCRT.h:302)
  std::map<long, vec_long> getMap() const {
    std::map<long, vec_long> m; long n = context.zMstar.phiM(); std::vector<uint64_t> row(n); IndexSet s = getIndexSet();
    for (long i = s.first(); i <= s.last(); i = s.next(i)) { ck(fhesi_dcrt_download_row(h, (int32_t)i, row.data())); m[i] = vec_long(row.begin(), row.end()); }
    return m;
  }
  void setMap(const std::map<long, vec_long>& m) {   // DoubleCRT.h: replace index set and rows (Import, Serialization.cpp:67-81)
    IndexSet s; for (auto& kv : m) s.insert(kv.first);
    if (h) { ck(fhesi_dcrt_free(h)); h = nullptr; }
    alloc(s);
    for (auto& kv : m) { if ((long)kv.second.size() != (long)context.zMstar.phiM()) Error("DoubleCRT::setMap: bad row length"); setRow(kv.first, kv.second); }
  }
  void setRow(long i, const vec_long& r) { std::vector<uint64_t> v(r.begin(), r.end()); ck(fhesi_dcrt_upload_row(h, (int32_t)i, v.data())); }
  fhesi_dcrt* handle() const { return h; }
  void randomize() { IndexSet s = getIndexSet(); long n = context.zMstar.phiM(); for (long i = s.first(); i <= s.last(); i = s.next(i)) { vec_long r(n); for (long j = 0; j < n; ++j) r[j] = RandomBnd(context.ithPrime(i)); setRow(i, r); } }   // :468-481
  void sampleSmall();
  void sampleHWt(long Hwt);
  void sampleGaussian(double stdev = 0.0);
  ZZ getCoefficientModulus() const { return context.productOfPrimes(); }
};
inline ZZX to_ZZX(const DoubleCRT& d) { ZZX p; d.toPoly(p); return p; }
inline void conv(DoubleCRT& d, const ZZX& p) { d = p; }                 // DoubleCRT.h:368-378
inline DoubleCRT to_DoubleCRT(const ZZX& p) { return DoubleCRT(p); }
inline void conv(ZZX& p, const DoubleCRT& d) { d.toPoly(p); }

// ---------------------------------------------------------------- SingleCRT (SingleCRT.h:41-175, SingleCRT.cpp)
// Coefficient-domain RNS form: per prime of the index set, the polynomial's coefficients modulo that prime, resident in HBM.
// Same member names and argument meaning as the reference; every operation is a C-ABI call on device rows.  One deliberate
// difference: the reference's SingleCRT::addPrimes stores the UNREDUCED polynomial in the new rows (`map[i] = poly;` instead of
// `poly1`, SingleCRT.cpp:262-266), which its own verify() would reject; the mirror stores the reduced residues the comment there
// describes.
class SingleCRT {
  const FHEcontext& context;
  fhesi_dcrt* h = nullptr;
  friend class DoubleCRT;
  void alloc(const IndexSet& s) { auto v = s.vec(); if (v.empty()) Error("SingleCRT: empty index set"); ck(fhesi_scrt_alloc(context.handle(), v.data(), (int32_t)v.size(), &h)); }
  void realloc(const IndexSet& s) { if (h) { ck(fhesi_dcrt_free(h)); h = nullptr; } alloc(s); }
  SingleCRT& Op(const SingleCRT& other, int op, bool matchIndexSets = true) {   // SingleCRT.cpp:61-103
    if (&context != &other.context) Error("SingleCRT::Op: incomopatible objects");
    if (matchIndexSets && !(getIndexSet() >= other.getIndexSet())) addPrimes(other.getIndexSet() / getIndexSet());
    if (getIndexSet() > other.getIndexSet()) { SingleCRT tmp(other); tmp.addPrimes(getIndexSet() / other.getIndexSet()); ck(fhesi_dcrt_op(h, tmp.h, op)); }
    else if (getIndexSet() == other.getIndexSet()) ck(fhesi_dcrt_op(h, other.h, op));
    else { SingleCRT tmp(other); tmp.removePrimes(other.getIndexSet() / getIndexSet()); ck(fhesi_dcrt_op(h, tmp.h, op)); }
    return *this;
  }
  SingleCRT& Op(const ZZX& poly, int op) { SingleCRT other(poly, context, getIndexSet()); return Op(other, op); }   // :105-135: PolyRed per prime, then AddMod / SubMod
  SingleCRT& Op(const ZZ& num, int op) { int nl = (int)(num.bits() / 64 + 2); std::vector<uint64_t> v(nl); num.to_limbs(v.data(), nl); ck(fhesi_scrt_op_scalar(h, v.data(), nl, op)); return *this; }   // :137-153
 public:
  SingleCRT(const ZZX& poly, const FHEcontext& c, const IndexSet& s) : context(c) { alloc(s); *this = poly; }
  SingleCRT(const ZZX& poly, const FHEcontext& c) : context(c) { alloc(IndexSet(0, c.numPrimes() - 1)); *this = poly; }
  explicit SingleCRT(const ZZX& poly) : context(*activeContext) { alloc(IndexSet(0, context.numPrimes() - 1)); *this = poly; }
  SingleCRT(const FHEcontext& c, const IndexSet& s) : context(c) { alloc(s); }
  explicit SingleCRT(const FHEcontext& c) : context(c) { alloc(IndexSet(0, c.numPrimes() - 1)); }
  SingleCRT() : context(*activeContext) { alloc(IndexSet(0, context.numPrimes() - 1)); }
  SingleCRT(const SingleCRT& o) : context(o.context) { alloc(o.getIndexSet()); ck(fhesi_dcrt_copy(h, o.h)); }
  ~SingleCRT() { if (h) fhesi_dcrt_free(h); }

  SingleCRT& operator=(const SingleCRT& o) { if (&context != &o.context) Error("SingleCRT assignment: context mismatch"); ck(fhesi_dcrt_copy(h, o.h)); return *this; }   // :219-228
  SingleCRT& operator=(const DoubleCRT& d) { d.toSingleCRT(*this); return *this; }                                                                                 // :231-235
  SingleCRT& operator=(const ZZX& poly) {                                                                                                                            // :239-251
    ZZX p = poly; p.normalize();
    if ((long)p.rep.size() > (long)context.zMstar.phiM()) Error("SingleCRT = ZZX: degree >= phi(m) is outside the device row layout");
    int nl = limbs_for(p); auto v = to_limbs(p, nl);
    ck(fhesi_scrt_from_poly(h, v.data(), nl, (int64_t)p.rep.size()));
    return *this;
  }
  SingleCRT& operator=(const ZZ& num) { ZZX p; p.rep.assign(1, num); p.normalize(); return *this = p; }
  SingleCRT& operator=(long num) { return *this = ZZ(num); }
  bool operator==(const SingleCRT& o) const { if (&context != &o.context) return false; int32_t eq = 0; ck(fhesi_dcrt_equal(h, o.h, &eq)); return eq != 0; }
  bool operator!=(const SingleCRT& o) const { return !(*this == o); }
  SingleCRT& setZero() { return *this = ZZ(); }
  SingleCRT& setOne() { return *this = 1L; }
  void addPrimes(const IndexSet& s1) {                                                                                                                               // :254-268
    assert(card(s1 & getIndexSet()) == 0);
    if (card(s1) == 0) return;
    ZZX poly; toPoly(poly);
    IndexSet uni = getIndexSet() | s1;
    SingleCRT grown(context, uni);
    long n = context.zMstar.phiM(); std::vector<uint64_t> row(n);
    IndexSet old = getIndexSet();
    for (long i = old.first(); i <= old.last(); i = old.next(i)) { ck(fhesi_dcrt_download_row(h, (int32_t)i, row.data())); ck(fhesi_dcrt_upload_row(grown.h, (int32_t)i, row.data())); }
    SingleCRT fresh(poly, context, s1);
    for (long i = s1.first(); i <= s1.last(); i = s1.next(i)) { ck(fhesi_dcrt_download_row(fresh.h, (int32_t)i, row.data())); ck(fhesi_dcrt_upload_row(grown.h, (int32_t)i, row.data())); }
    realloc(uni); ck(fhesi_dcrt_copy(h, grown.h));
  }
  void removePrimes(const IndexSet& s1) { auto v = s1.vec(); if (v.empty()) return; ck(fhesi_dcrt_remove_primes(h, v.data(), (int32_t)v.size())); }   // SingleCRT.h:117-119
  SingleCRT& operator+=(const SingleCRT& o) { return Op(o, OP_ADD); }
  SingleCRT& operator+=(const ZZX& p) { return Op(p, OP_ADD); }
  SingleCRT& operator+=(const ZZ& n) { return Op(n, OP_ADD); }
  SingleCRT& operator+=(long n) { return Op(ZZ(n), OP_ADD); }
  SingleCRT& operator-=(const SingleCRT& o) { return Op(o, OP_SUB); }
  SingleCRT& operator-=(const ZZX& p) { return Op(p, OP_SUB); }
  SingleCRT& operator-=(const ZZ& n) { return Op(n, OP_SUB); }
  SingleCRT& operator-=(long n) { return Op(ZZ(n), OP_SUB); }
  void Add(const SingleCRT& o, bool match = true) { Op(o, OP_ADD, match); }
  void Sub(const SingleCRT& o, bool match = true) { Op(o, OP_SUB, match); }
  SingleCRT& operator++() { return *this += 1L; }
  SingleCRT& operator--() { return *this -= 1L; }
  void operator++(int) { *this += 1L; }
  void operator--(int) { *this -= 1L; }
  SingleCRT& operator*=(const ZZ& n) { return Op(n, OP_MUL); }
  SingleCRT& operator*=(long n) { return Op(ZZ(n), OP_MUL); }
  SingleCRT& operator/=(const ZZ& n) { return Op(n, OP_DIV); }                                                                                                       // :279-296
  SingleCRT& operator/=(long n) { return Op(ZZ(n), OP_DIV); }
  void toPoly(ZZX& p, const IndexSet& s) const {                                                                                                                     // :299-334
    IndexSet s1 = getIndexSet() & s;
    if (card(s1) == 0) { clear(p); return; }
    int nl = (int)card(s1) + 2; long n = context.zMstar.phiM(); auto idx = s1.vec();
    std::vector<uint64_t> out((size_t)n * nl);
    ck(fhesi_scrt_to_poly(h, idx.data(), (int32_t)idx.size(), out.data(), nl));
    p = from_limbs(out, n, nl);
  }
  void toPoly(ZZX& p) const { toPoly(p, getIndexSet()); }
  const FHEcontext& getContext() const { return context; }
  IndexSet getIndexSet() const { int32_t n = 0; std::vector<int32_t> v(64); ck(fhesi_dcrt_index_set(h, v.data(), &n)); IndexSet s; for (int i = 0; i < n; ++i) s.insert(v[i]); return s; }
  fhesi_dcrt* handle() const { return h; }
};
inline void conv(SingleCRT& s, const ZZX& p) { s = p; }
inline void conv(ZZX& p, const SingleCRT& s) { s.toPoly(p); }
inline ZZX to_ZZX(const SingleCRT& s) { ZZX p; s.toPoly(p); return p; }
inline void conv(DoubleCRT& d, const SingleCRT& s);                      // DoubleCRT.h:380
inline DoubleCRT& DoubleCRT::operator=(const SingleCRT& scrt) {
  if (&context != &scrt.getContext()) Error("DoubleCRT=SingleCRT -- incompatible contexts");
  ck(fhesi_dcrt_assign_scrt(h, scrt.handle()));
  return *this;
}
inline void DoubleCRT::toSingleCRT(SingleCRT& scrt, const IndexSet& s) const {
  if (&context != &scrt.getContext()) Error("DoubleCRT::toSingleCRT -- incompatible contexts");
  auto v = s.vec();
  if (v.empty()) Error("DoubleCRT::toSingleCRT: empty index set");
  ck(fhesi_scrt_assign_dcrt(scrt.handle(), h, v.data(), (int32_t)v.size()));
}
inline void DoubleCRT::toSingleCRT(SingleCRT& scrt) const { toSingleCRT(scrt, getIndexSet()); }
inline void conv(DoubleCRT& d, const SingleCRT& s) { d = s; }

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
template <typename T> void DotProduct(T& res, const std::vector<T>& v1, const std::vector<T>& v2) {   // Util.h:79-98
  if (v1.empty()) return;
  res = v1[0]; res *= v2[0];
  for (size_t i = 1; i < v1.size(); ++i) { T val = v1[i]; val *= v2[i]; res += val; }
}

}  // namespace fhesi
#include "fhesi_engine.h"   // device-resident, lazily evaluated ciphertext values (needs FHEcontext above)
namespace fhesi {

// ---------------------------------------------------------------- Ciphertext (Ciphertext.h, Ciphertext.cpp)
class CiphertextPart {
  const FHEcontext& context;
 public:
  ZZX poly;
  CiphertextPart() : context(*activeContext) {}
  CiphertextPart(const FHEcontext& c) : context(c) {}
  explicit CiphertextPart(const ZZX& p) : context(*activeContext), poly(p) {}
  CiphertextPart(const CiphertextPart& o) : context(o.context), poly(o.poly) {}
  CiphertextPart& operator=(const CiphertextPart& o) { if (&context != &o.context) Error("Incompatible contexts."); poly = o.poly; return *this; }
  CiphertextPart& operator+=(const ZZX& o) { poly += o; return *this; }
  CiphertextPart& operator+=(const CiphertextPart& o) { poly += o.poly; return *this; }
  CiphertextPart& operator*=(long l) { for (auto& c : poly.rep) { c *= ZZ(l); Reduce(c, context.logQ); } poly.normalize(); return *this; }   // Ciphertext.cpp:21-27
  CiphertextPart& operator*=(const ZZX& o) { poly = mul(poly, o); rem(poly, poly, context.zMstar.PhimX()); for (auto& c : poly.rep) Reduce(c, context.logQ); poly.normalize(); return *this; }   // :29-36 (host form; Ciphertext::operator*=(ZZX) takes the device call)
  CiphertextPart& operator>>=(long k) { DoubleCRT tmp(poly); tmp >>= k; tmp.toPoly(poly); return *this; }                                    // :54-59
  bool operator==(const CiphertextPart& o) const { return poly == o.poly; }
};

// The unscaled parts of a Ciphertext: the reference's `vector<CiphertextPart> parts` (Ciphertext.h:71) with the same access
// (size, [], iteration, assign, push_back, =), whose contents may live in HBM as a CtValue (fhesi_engine.h) instead of in host big
// integers.  Every access through this interface brings them to the host first; a writable access also drops the device image.
class CtParts {
 public:
  typedef std::vector<CiphertextPart> Vec;
 private:
  mutable Vec host_;
  mutable bool onHost = true;          // false: the value is `val` only (always two parts)
  mutable CtRef val;                   // the same two parts in HBM, or the recorded operation that will produce them; null: host only
  void fetch() const {
    CtEngine& e = *val->eng;
    std::vector<uint64_t> lim((size_t)e.words);
    e.download(val, lim.data());
    host_.assign(2, CiphertextPart(e.ctx()));
    for (int part = 0; part < 2; ++part) limbs_to_poly(host_[part].poly, &lim[(size_t)part * e.n * e.nl], e.n, e.nl);
    onHost = true;
  }
 public:
  const Vec& host() const { if (!onHost) fetch(); return host_; }
  Vec& host() { if (!onHost) fetch(); val.reset(); return host_; }
  size_t size() const { return onHost ? host_.size() : 2; }
  bool empty() const { return size() == 0; }
  CiphertextPart& operator[](size_t i) { return host()[i]; }
  const CiphertextPart& operator[](size_t i) const { return host()[i]; }
  Vec::iterator begin() { return host().begin(); }
  Vec::iterator end() { return host().end(); }
  Vec::const_iterator begin() const { return host().begin(); }
  Vec::const_iterator end() const { return host().end(); }
  void clear() { host_.clear(); onHost = true; val.reset(); }
  void assign(size_t cnt, const CiphertextPart& v) { clear(); host_.assign(cnt, v); }
  void push_back(const CiphertextPart& v) { host().push_back(v); }
  CtParts& operator=(const Vec& v) { clear(); host_ = v; return *this; }
  operator const Vec&() const { return host(); }
  // the device side
  bool resident() const { return (bool)val; }
  const CtRef& value() const { return val; }
  void set_value(CtRef v) { host_.clear(); onHost = false; val = std::move(v); }     // the value lives in HBM from now on
  void cache_value(CtRef v) const { val = std::move(v); }                              // ... in both places
};

class Ciphertext {
  const FHEcontext* context;
  // scaled up (Ciphertext.cpp:167-192): the tensor product as DoubleCRT objects, or -- while nobody has looked at it -- as the list of
  // products of device-resident ciphertexts it is the sum of (multiplied out by the key switch that consumes it, fhesi_engine.h)
  mutable std::vector<DoubleCRT> tProd;
  mutable CtTerms terms;
  bool scaledUp = false;
  friend class KeySwitchSI;            // ApplyKeySwitch hands the scaled-up rows to the fused device call without a round trip through the host
  friend class FHESISecKey;
  friend class FHESIPubKey;
  CtEngine& engine() const { return ct_engine(*context); }
  bool lazy2() const { return LazyCiphertexts() && !scaledUp && parts.size() == 2; }
  // multiply the recorded products out into tProd (someone wants the rows themselves)
  void materialise() const {
    if (terms.empty()) return;
    CtEngine& e = engine(); e.flush();
    const long n = e.n, L = context->numPrimes();
    void* tp; ck(fhesi_dev_alloc(e.h, (size_t)3 * L * n * 8, &tp));
    tProd.clear();
    for (auto& t : terms) {
      int rc = fhesi_ct_mul_dev(e.h, (uint64_t)context->ModulusP().to_long(), e.ptr(t.first->slot), e.ptr(t.second->slot), e.nl, 1, (uint64_t*)tp);
      std::vector<DoubleCRT> one(3, DoubleCRT(*context));
      for (int i = 0; i < 3 && !rc; ++i) rc = fhesi_dev_copy(e.h, fhesi_dcrt_device_ptr(one[i].handle()), (const uint64_t*)tp + (size_t)i * L * n, (size_t)L * n * 8);
      if (rc) { fhesi_dev_free(e.h, tp); ck(rc); }
      if (tProd.empty()) tProd = one; else for (int i = 0; i < 3; ++i) tProd[i] += one[i];
    }
    ck(fhesi_dev_free(e.h, tp));
    terms.clear();
  }
 public:
  CtParts parts;
  Ciphertext() : context(activeContext) {}
  Ciphertext(const FHEcontext& c) : context(&c) {}
  void Initialize(unsigned n, const FHEcontext& c) { context = &c; parts.assign(n, CiphertextPart(c)); }
  unsigned size() const { return scaledUp ? (terms.empty() ? (unsigned)tProd.size() : 3u) : (unsigned)parts.size(); }
  CiphertextPart& operator[](unsigned i) { return parts[i]; }
  CiphertextPart GetPart(unsigned i) const { return parts[i]; }
  bool isScaledUp() const { return scaledUp; }
  void Clear() { tProd.clear(); terms.clear(); scaledUp = false; parts.clear(); }     // Ciphertext.cpp:226-230
  void SetTensorRepresentation(std::vector<DoubleCRT>& repr) { parts.clear(); terms.clear(); std::swap(tProd, repr); scaledUp = true; }   // Ciphertext.cpp:220-224
  // this unscaled two-part ciphertext as a value in HBM (uploaded once, then shared by every copy and every product that uses it)
  CtRef device_value() const {
    if (parts.resident()) return parts.value();
    if (scaledUp || parts.size() != 2) Error("Ciphertext::device_value: expects an unscaled 2-part ciphertext");
    CtEngine& e = engine();
    std::vector<uint64_t> lim((size_t)e.words);
    for (int part = 0; part < 2; ++part) poly_to_limbs(parts.host()[part].poly, &lim[(size_t)part * e.n * e.nl], e.n, e.nl);
    parts.cache_value(e.upload(lim.data()));
    return parts.value();
  }
  void set_device_value(CtRef v) { tProd.clear(); terms.clear(); scaledUp = false; parts.set_value(std::move(v)); }

  Ciphertext& operator+=(const Ciphertext& o) {   // Ciphertext.cpp:123-145
    assert(scaledUp == o.scaledUp);
    if (!scaledUp) {
      if (lazy2() && o.parts.size() == 2 && (parts.resident() || o.parts.resident())) { CtRef a = device_value(), b = o.device_value(); parts.set_value(engine().add(a, b)); return *this; }
      CtParts::Vec& mine = parts.host(); const CtParts::Vec& theirs = o.parts.host();
      unsigned i = 0;
      for (; i < mine.size() && i < theirs.size(); ++i) { mine[i] += theirs[i]; ReduceCoefficients(mine[i].poly, context->logQ); }
      for (; i < theirs.size(); ++i) mine.push_back(theirs[i]);
    } else {
      if (tProd.empty() && o.tProd.empty()) { CtTerms add = o.terms; terms.insert(terms.end(), add.begin(), add.end()); return *this; }   // both still recorded: the sum of all their products
      materialise(); o.materialise();
      unsigned i = 0;
      for (; i < tProd.size() && i < o.tProd.size(); ++i) tProd[i] += o.tProd[i];
      for (; i < o.tProd.size(); ++i) tProd.push_back(o.tProd[i]);
    }
    return *this;
  }
  Ciphertext& operator*=(const Ciphertext& o) {   // Ciphertext.cpp:167-192
    if (!scaledUp && !o.scaledUp && parts.size() == 2 && o.parts.size() == 2) {
      // two fresh ciphertexts (every multiplication the reference's drivers perform)
      if (LazyCiphertexts()) {          // recorded: the key switch that follows takes the sum of such products in one device call
        CtRef a = device_value(), b = o.device_value();
        terms.assign(1, std::make_pair(a, b)); tProd.clear(); parts.clear(); scaledUp = true;
        return *this;
      }
      // at once: the lift by p, the four DoubleCRT conversions and the tensor products as ONE device call (fhesi_ct_mul_dev) instead of
      // 4 + 4 + 4 object operations; the same rows, bit for bit (tests/host/test_wire.cpp compares with MulObjects below)
      fhesi_ctx* h = context->handle();
      const long n = context->zMstar.phiM(), L = context->numPrimes(); const int nl = (int)((context->logQ + 63) / 64);
      std::vector<uint64_t> host((size_t)2 * 2 * n * nl, 0);
      for (int part = 0; part < 2; ++part) { poly_to_limbs(parts[part].poly, &host[(size_t)part * n * nl], n, nl); poly_to_limbs(o.parts[part].poly, &host[(size_t)(2 + part) * n * nl], n, nl); }
      void *in, *tp; ck(fhesi_dev_alloc(h, host.size() * 8, &in)); ck(fhesi_dev_alloc(h, (size_t)3 * L * n * 8, &tp));
      ck(fhesi_dev_upload(h, in, host.data(), host.size() * 8));
      int rc = fhesi_ct_mul_dev(h, (uint64_t)context->ModulusP().to_long(), (const uint64_t*)in, (const uint64_t*)in + (size_t)2 * n * nl, nl, 1, (uint64_t*)tp);
      if (!rc) { tProd.assign(3, DoubleCRT(*context)); for (int i = 0; i < 3 && !rc; ++i) rc = fhesi_dev_copy(h, fhesi_dcrt_device_ptr(tProd[i].handle()), (const uint64_t*)tp + (size_t)i * L * n, (size_t)L * n * 8); }
      fhesi_dev_free(h, in); fhesi_dev_free(h, tp);
      ck(rc);
      parts.clear(); scaledUp = true;
      return *this;
    }
    return MulObjects(o);
  }
  Ciphertext& MulObjects(const Ciphertext& o) {   // the reference's loop, one DoubleCRT object at a time
    std::vector<DoubleCRT> c1, c2;
    for (auto& p : parts) c1.push_back(DoubleCRT(p.poly * context->ModulusP(), *context));
    for (auto& p : o.parts) c2.push_back(DoubleCRT(p.poly, *context));
    tProd.assign(c1.size() + c2.size() - 1, DoubleCRT(*context)); terms.clear();
    for (size_t i = 0; i < c1.size(); ++i)
      for (size_t j = 0; j < c2.size(); ++j) { DoubleCRT tmp = c1[i]; tmp *= c2[j]; tProd[i + j] += tmp; }
    parts.clear(); scaledUp = true;
    return *this;
  }
  Ciphertext& operator*=(long l) {   // Ciphertext.cpp:232-243
    if (lazy2() && parts.resident()) { parts.set_value(engine().scale(parts.value(), l)); return *this; }
    if (!scaledUp) for (auto& p : parts) p *= l; else { materialise(); for (auto& t : tProd) t *= l; }
    return *this;
  }
  // operator+=(const ZZX&) (Ciphertext.cpp:147-161): the constant is scaled by q / p with NTL's floor division and added to part 0
  // (unscaled: device call fhesi_ct_add_const_dev when the coefficients are machine words, else the same arithmetic on the host), or to
  // tProd[0] (scaled-up: DoubleCRT += ZZX).  The std::vector<long> overloads take the role of the reference's ZZ_pX ones (:158-160, :256-258):
  // the mirror's Plaintext holds its message as machine words.
  Ciphertext& operator+=(const ZZX& other) {
    std::vector<int64_t> small;
    if (!scaledUp && words_of(other, small)) { with_parts_on_device([&](uint64_t* dev, int nl) {
        ck(fhesi_ct_add_const_dev(context->handle(), (int32_t)context->logQ, (uint64_t)context->ModulusP().to_long(), dev, (int32_t)parts.size(), nl, 1, small.data(), 1)); });
      return *this; }
    ZZX sc(other);
    for (auto& c : sc.rep) { c <<= (long)context->logQ; c /= context->ModulusP(); }     // floor division, like NTL
    sc.normalize();
    if (!scaledUp) { parts[0] += sc; ReduceCoefficients(parts[0].poly, context->logQ); } else { materialise(); tProd[0] += sc; }
    return *this;
  }
  Ciphertext& operator+=(const std::vector<long>& msg) { return *this += words_to_ZZX(msg); }
  // operator*=(const ZZX&) (Ciphertext.cpp:245-258): unscaled -- every part times the polynomial over the integers, modulo Phi_m, Reduce
  // (CiphertextPart::operator*=(ZZX), :29-36; device call fhesi_ct_mul_poly_dev); scaled-up -- tProd[i] *= DoubleCRT(other)
  Ciphertext& operator*=(const ZZX& other) {
    if (scaledUp) { materialise(); DoubleCRT o(other, *context); for (auto& t : tProd) t *= o; return *this; }
    std::vector<int64_t> small;
    if (words_of(other, small)) with_parts_on_device([&](uint64_t* dev, int nl) { ck(fhesi_ct_mul_poly_dev(context->handle(), (int32_t)context->logQ, dev, (int32_t)parts.size(), nl, 1, small.data(), 1)); });
    else for (auto& p : parts) p *= other;
    return *this;
  }
  Ciphertext& operator*=(const std::vector<long>& msg) { return *this *= words_to_ZZX(msg); }
 private:
  static ZZX words_to_ZZX(const std::vector<long>& v) { ZZX p; p.rep.resize(v.size()); for (size_t i = 0; i < v.size(); ++i) p.rep[i] = ZZ(v[i]); p.normalize(); return p; }
  // the polynomial as phi(m) machine words, if every coefficient fits one (a ZZ_pX message always does)
  bool words_of(const ZZX& p, std::vector<int64_t>& out) const {
    const long n = context->zMstar.phiM();
    if ((long)p.rep.size() > n) return false;
    out.assign(n, 0);
    for (size_t i = 0; i < p.rep.size(); ++i) { if (p.rep[i].bits() > 62) return false; out[i] = (int64_t)p.rep[i].to_long(); }
    return true;
  }
  // the unscaled parts as one device ciphertext [nparts][phi(m)][nl] around a device call: on a copy of the value's arena slot (the result
  // stays in HBM), or -- recording off, or not two parts -- through a temporary buffer and back to the host
  template <class Fn> void with_parts_on_device(Fn fn) {
    if (lazy2()) { CtEngine& e = engine(); const long s = e.clone_slot(device_value()); fn(e.ptr(s), e.nl); e.publish(s, 1); parts.set_value(e.wrap(s)); return; }
    const long n = context->zMstar.phiM(); const int nl = (int)((context->logQ + 63) / 64); const size_t np = parts.size();
    std::vector<uint64_t> host(np * n * nl, 0);
    for (size_t i = 0; i < np; ++i) poly_to_limbs(parts[i].poly, &host[(i * n) * nl], n, nl);
    void* dev; ck(fhesi_dev_alloc(context->handle(), host.size() * 8, &dev)); ck(fhesi_dev_upload(context->handle(), dev, host.data(), host.size() * 8));
    fn((uint64_t*)dev, nl);
    ck(fhesi_dev_download(context->handle(), host.data(), dev, host.size() * 8)); ck(fhesi_dev_free(context->handle(), dev));
    for (size_t i = 0; i < np; ++i) limbs_to_poly(parts[i].poly, &host[(i * n) * nl], n, nl);
  }
 public:
  Ciphertext& operator>>=(long k) {   // Ciphertext.cpp:264-275
    if (lazy2() && parts.resident()) {
      if (!context->zMstar.inZmStar((unsigned)k)) Error("DoubleCRT::automorph: k not in Zm*");
      parts.set_value(engine().automorph(parts.value(), k)); return *this;
    }
    if (!scaledUp) for (auto& p : parts) p >>= k; else { materialise(); for (auto& t : tProd) t >>= k; }
    return *this;
  }
  void ScaleDown() {   // Ciphertext.cpp:194-218
    if (!scaledUp) return;
    materialise();
    ZZ q = context->modulusQ, q2 = q * ZZ(2L);
    parts.clear();
    for (auto& t : tProd) {
      ZZX part; t.toPoly(part);
      for (auto& c : part.rep) { c *= ZZ(2L); c += q; c /= q2; }     // floor division, like NTL
      part.normalize(); ReduceCoefficients(part, context->logQ);
      CiphertextPart cp(*context); cp.poly = part; parts.push_back(cp);
    }
    scaledUp = false; tProd.clear();
  }
  Ciphertext& ByteDecomp() {   // Ciphertext.cpp:82-121: part-major, digit-minor
    std::vector<CiphertextPart> orig = parts.host(); const unsigned nd = context->ndigits, bits = 8 * context->decompSize;
    parts.assign(orig.size() * nd, CiphertextPart(*context));
    ZZ mask = (ZZ(1L) << (long)bits) - ZZ(1L);
    for (size_t pi = 0; pi < orig.size(); ++pi)
      for (long i = 0; i <= deg(orig[pi].poly); ++i) {
        ZZ c = coeff(orig[pi].poly, i); Reduce(c, context->logQ, true);
        for (unsigned d = 0; d < nd; ++d) { ZZ dig = c >> (long)(bits * d); ZZ low; low.mag = dig.mag; if (low.mag.size() > 1) low.mag.resize(1); if (!low.mag.empty()) low.mag[0] &= (bits >= 64 ? ~0ull : ((1ull << bits) - 1)); low.trim(); if (!low.is_zero()) SetCoeff(parts[pi * nd + d].poly, i, low); }
      }
    return *this;
  }
};

// ---------------------------------------------------------------- Plaintext (coefficient form only; slot packing is out of scope)
struct Plaintext { std::vector<long> message; };

// ---------------------------------------------------------------- FHE-SI.cpp: keys and key switching
class FHESISecKey {
  const FHEcontext& context;
  std::vector<DoubleCRT> sKeys;
 public:
  FHESISecKey(const FHEcontext& c) : context(c) { Init(c); }
  void Init(const FHEcontext& c) { sKeys.assign(2, DoubleCRT(c)); sKeys[0] = 1L; sKeys[1].sampleHWt(64); }   // FHE-SI.cpp:86-91
  const std::vector<DoubleCRT>& GetRepresentation() const { return sKeys; }
  void UpdateRepresentation(const std::vector<DoubleCRT>& r) { sKeys = r; }
  const FHEcontext& GetContext() const { return context; }
  size_t GetSize() const { return sKeys.size(); }
  // Decrypt for many unscaled 2-part ciphertexts in one device call (fhesi_decrypt_batch); same values as repeated Decrypt calls
  void DecryptBatch(std::vector<Plaintext>& ptxts, const std::vector<Ciphertext>& ctxts) const {
    const long n = context.zMstar.phiM(), count = (long)ctxts.size(); const int nl = (int)((context.logQ + 63) / 64);
    std::vector<int64_t> msg((size_t)count * n);
    if (LazyCiphertexts() && count) {
      // the ciphertexts as values in HBM (whatever was recorded for them runs now), gathered into one run of the arena
      CtEngine& e = ct_engine(context);
      std::vector<CtRef> vals; for (auto& c : ctxts) vals.push_back(c.device_value());
      e.flush();
      std::vector<int32_t> idx; for (auto& v : vals) { e.force(v); idx.push_back((int32_t)v->slot); }
      const long run = e.alloc_run(count);
      ck(fhesi_ct_gather_dev(e.h, e.pool(), idx.data(), count, e.words, e.ptr(run)));
      int rc = fhesi_decrypt_batch(e.h, sKeys[1].handle(), (int32_t)context.logQ, (uint64_t)context.ModulusP().to_long(), e.ptr(run), nl, count, msg.data());
      e.free_run(run, count);
      ck(rc);
    } else {
      std::vector<uint64_t> host((size_t)count * 2 * n * nl);
      for (long c = 0; c < count; ++c) for (int part = 0; part < 2; ++part) for (long j = 0; j < n; ++j) coeff(ctxts[c].GetPart((unsigned)part).poly, j).to_limbs(&host[((c * 2 + part) * n + j) * nl], nl);
      void* dev; ck(fhesi_dev_alloc(context.handle(), host.size() * 8, &dev)); ck(fhesi_dev_upload(context.handle(), dev, host.data(), host.size() * 8));
      ck(fhesi_decrypt_batch(context.handle(), sKeys[1].handle(), (int32_t)context.logQ, (uint64_t)context.ModulusP().to_long(), (const uint64_t*)dev, nl, count, msg.data()));
      ck(fhesi_dev_free(context.handle(), dev));
    }
    ptxts.assign(count, Plaintext());
    for (long c = 0; c < count; ++c) ptxts[c].message.assign(msg.begin() + c * n, msg.begin() + (c + 1) * n);
  }
  void Decrypt(Plaintext& ptxt, const Ciphertext& ctxt) const {   // FHE-SI.cpp:93-119
    if (LazyCiphertexts() && !ctxt.isScaledUp() && ctxt.parts.resident() && sKeys.size() == 2) {
      // the ciphertext lives in HBM: the same dot product with (1, t), rounding and reduction as ONE device call on it (fhesi_decrypt_batch)
      CtEngine& e = ct_engine(context); CtRef v = ctxt.parts.value(); e.force(v);
      std::vector<int64_t> msg((size_t)e.n);
      ck(fhesi_decrypt_batch(e.h, sKeys[1].handle(), (int32_t)context.logQ, (uint64_t)context.ModulusP().to_long(), e.ptr(v->slot), e.nl, 1, msg.data()));
      ptxt.message.assign(msg.begin(), msg.end());
      return;
    }
    std::vector<DoubleCRT> cp, sp;
    for (size_t i = 0; i < sKeys.size(); ++i) { cp.push_back(DoubleCRT(ctxt.GetPart((unsigned)i).poly, context)); sp.push_back(sKeys[i]); }
    DoubleCRT tmp(context); DotProduct(tmp, cp, sp);
    ZZX z; tmp.toPoly(z);
    ZZ p = context.ModulusP(), q = context.modulusQ, q2 = q * ZZ(2L);
    ptxt.message.assign(context.zMstar.phiM(), 0);
    for (long i = 0; i <= deg(z); ++i) { ZZ c = z.rep[i]; c *= ZZ(2L) * p; c += q; c /= q2; ptxt.message[i] = rem(c, p.to_long()); }
  }
};
// One stream of on-device randomness (csrc/philox.h): the secret seed, the public seed of the key polynomials, and ONE monotonically
// increasing object counter shared by every Encrypt and every KeySwitchSI that draws from it -- callers never pick indices, so a pair
// (seed, index) cannot be handed out twice.  Both seeds must be uniformly random and the secret one stays secret; Philox is not a CSPRNG
// (64-bit key): see include/fhesi_hip.h for what that is good for.
struct SeedSequence {
  const uint64_t seed, public_seed;
  SeedSequence(uint64_t secret, uint64_t pub, uint64_t first = 0) : seed(secret), public_seed(pub), next(first) { if (secret == pub) Error("SeedSequence: the public seed must differ from the secret seed"); }
  uint64_t take(uint64_t count) { return next.fetch_add(count); }      // first index of a fresh range of `count` objects
  uint64_t used() const { return next.load(); }
 private:
  std::atomic<uint64_t> next;
};

class FHESIPubKey {
  const FHEcontext& context;
  std::vector<DoubleCRT> publicKey;
 public:
  FHESIPubKey(const FHESISecKey& sk) : context(sk.GetContext()) { Init(sk); }
  const FHEcontext& GetContext() const { return context; }
  const std::vector<DoubleCRT>& GetRepresentation() const { return publicKey; }
  void UpdateRepresentation(const std::vector<DoubleCRT>& r) { publicKey = r; }
  void Init(const FHESISecKey& sk) {   // FHE-SI.cpp:42-63
    ZZX c0, c1; sampleGaussian(c0, context.zMstar.phiM(), context.stdev); SampleRandom(c1, context.modulusQ, context.zMstar.phiM());
    ZZX tmp; sk.GetRepresentation()[1].toPoly(tmp); tmp = mul(tmp, c1);
    c0 += tmp; rem(c0, c0, context.zMstar.PhimX()); c1 *= ZZ(-1L);
    ReduceCoefficients(c0, context.logQ); ReduceCoefficients(c1, context.logQ);
    publicKey.clear(); publicKey.push_back(DoubleCRT(c0, context)); publicKey.push_back(DoubleCRT(c1, context));
  }
  // Encrypt for many plaintexts in one device call (fhesi_encrypt_batch).  The randomness is drawn here, per plaintext, in the
  // order Encrypt draws it (r, noise of part 0, noise of part 1), so the ciphertexts equal those of repeated Encrypt calls.
  void EncryptBatch(std::vector<Ciphertext>& ctxts, const std::vector<Plaintext>& ptxts) const {
    const long n = context.zMstar.phiM(), count = (long)ptxts.size(); const int nl = (int)((context.logQ + 63) / 64);
    std::vector<int64_t> rnd((size_t)count * 3 * n), msg((size_t)count * n, 0);
    for (long c = 0; c < count; ++c) {
      for (long j = 0; j < n; ++j) rnd[(c * 3) * n + j] = RandomBnd(2L);
      for (int i = 0; i < 2; ++i) { ZZX e; sampleGaussian(e, n, context.stdev); for (long j = 0; j < n; ++j) rnd[(c * 3 + 1 + i) * n + j] = coeff(e, j).to_long(); }
      for (size_t k = 0; k < ptxts[c].message.size() && (long)k < n; ++k) msg[c * n + k] = ptxts[c].message[k];
    }
    if (LazyCiphertexts() && count) {            // the ciphertexts stay in HBM, as consecutive slots of the arena
      CtEngine& e = ct_engine(context); const long first = e.alloc_run(count);
      int rc = fhesi_encrypt_batch(e.h, publicKey[0].handle(), publicKey[1].handle(), (int32_t)context.logQ, (uint64_t)context.ModulusP().to_long(), rnd.data(), msg.data(), count, e.ptr(first), nl);
      if (rc) { e.free_run(first, count); ck(rc); }
      e.publish(first, count);
      ctxts.assign(count, Ciphertext(context));
      for (long c = 0; c < count; ++c) ctxts[c].set_device_value(e.wrap(first + c));
      return;
    }
    void* dev; ck(fhesi_dev_alloc(context.handle(), (size_t)count * 2 * n * nl * 8, &dev));
    ck(fhesi_encrypt_batch(context.handle(), publicKey[0].handle(), publicKey[1].handle(), (int32_t)context.logQ, (uint64_t)context.ModulusP().to_long(), rnd.data(), msg.data(),
                           count, (uint64_t*)dev, nl));
    std::vector<uint64_t> host((size_t)count * 2 * n * nl);
    ck(fhesi_dev_download(context.handle(), host.data(), dev, host.size() * 8)); ck(fhesi_dev_free(context.handle(), dev));
    ctxts.assign(count, Ciphertext(context));
    for (long c = 0; c < count; ++c) {
      ctxts[c].Initialize(2, context);
      for (int part = 0; part < 2; ++part) { ZZX poly; poly.rep.resize(n); for (long j = 0; j < n; ++j) poly.rep[j] = ZZ::from_limbs(&host[((c * 2 + part) * n + j) * nl], nl); poly.normalize(); ctxts[c][part].poly = poly; }
    }
  }
  // ... with r and the noise drawn ON THE DEVICE from the counter-based generator (fhesi_encrypt_batch_seeded, csrc/philox.h): plaintext i
  // uses the streams of object index first + i, so a batch can be split or repeated anywhere and give the same ciphertexts
  // (no default index: an (seed, index) pair used twice repeats r, e0, e1 -- the difference of the two ciphertexts is delta (m1 - m2) in the clear)
  void EncryptBatchSeeded(std::vector<Ciphertext>& ctxts, const std::vector<Plaintext>& ptxts, SeedSequence& seq) const { EncryptBatchSeeded(ctxts, ptxts, seq.seed, seq.take(ptxts.size())); }
  void EncryptBatchSeeded(std::vector<Ciphertext>& ctxts, const std::vector<Plaintext>& ptxts, uint64_t seed, uint64_t first_obj) const {
    const long n = context.zMstar.phiM(), count = (long)ptxts.size(); const int nl = (int)((context.logQ + 63) / 64);
    std::vector<int64_t> msg((size_t)count * n, 0);
    for (long c = 0; c < count; ++c) for (size_t k = 0; k < ptxts[c].message.size() && (long)k < n; ++k) msg[c * n + k] = ptxts[c].message[k];
    if (LazyCiphertexts() && count) {
      CtEngine& e = ct_engine(context); const long first = e.alloc_run(count);
      int rc = fhesi_encrypt_batch_seeded(e.h, publicKey[0].handle(), publicKey[1].handle(), (int32_t)context.logQ, (uint64_t)context.ModulusP().to_long(), seed, first_obj, msg.data(), count, e.ptr(first), nl);
      if (rc) { e.free_run(first, count); ck(rc); }
      e.publish(first, count);
      ctxts.assign(count, Ciphertext(context));
      for (long c = 0; c < count; ++c) ctxts[c].set_device_value(e.wrap(first + c));
      return;
    }
    void* dev; ck(fhesi_dev_alloc(context.handle(), (size_t)count * 2 * n * nl * 8, &dev));
    ck(fhesi_encrypt_batch_seeded(context.handle(), publicKey[0].handle(), publicKey[1].handle(), (int32_t)context.logQ, (uint64_t)context.ModulusP().to_long(), seed, first_obj, msg.data(), count, (uint64_t*)dev, nl));
    std::vector<uint64_t> host((size_t)count * 2 * n * nl);
    ck(fhesi_dev_download(context.handle(), host.data(), dev, host.size() * 8)); ck(fhesi_dev_free(context.handle(), dev));
    ctxts.assign(count, Ciphertext(context));
    for (long c = 0; c < count; ++c) { ctxts[c].Initialize(2, context); for (int part = 0; part < 2; ++part) limbs_to_poly(ctxts[c][part].poly, &host[((size_t)(c * 2 + part) * n) * nl], n, nl); }
  }
  // Encrypt (FHE-SI.cpp:10-36).  With recording on, the randomness is drawn here exactly as below and the arithmetic is the device call of
  // EncryptBatch on one plaintext; the ciphertext stays in HBM (the same bits: tests/host/test_wire.cpp compares EncryptBatch with EncryptObjects)
  void Encrypt(Ciphertext& ctxt, const Plaintext& ptxt) const {
    if (!LazyCiphertexts()) { EncryptObjects(ctxt, ptxt); return; }
    std::vector<Ciphertext> one;
    EncryptBatch(one, std::vector<Plaintext>(1, ptxt));
    ctxt = one[0];
  }
  void EncryptObjects(Ciphertext& ctxt, const Plaintext& ptxt) const {   // the reference's body, one DoubleCRT object at a time
    ctxt.Initialize(2, context);
    ZZX small; small.rep.assign(context.zMstar.phiM(), ZZ());
    for (auto& c : small.rep) c = ZZ(RandomBnd(2L));
    small.normalize();
    DoubleCRT r(small, context), e(context);
    std::vector<DoubleCRT> ct = publicKey;
    for (size_t i = 0; i < ct.size(); ++i) { e.sampleGaussian(); e *= context.ModulusP(); ct[i] *= r; ct[i] += e; ct[i].toPoly(ctxt[(unsigned)i].poly); }
    ZZ delta = context.modulusQ / context.ModulusP(); ZZX msg;
    for (size_t k = 0; k < ptxt.message.size(); ++k) SetCoeff(msg, (long)k, ZZ(ptxt.message[k]));
    ctxt[0] += delta * msg;
    for (size_t i = 0; i < ct.size(); ++i) ReduceCoefficients(ctxt[(unsigned)i].poly, context.logQ);
  }
};
class KeySwitchSI {
  const FHEcontext& context;
  std::vector<std::vector<DoubleCRT>> keySwitchMatrix;
  bool objectAtATime = false;          // checker mode: build the matrix with the reference's per-object loop (InitObjects)
  void InitAny(const FHESISecKey& src, const FHESISecKey& dst) { if (objectAtATime) InitObjects(src, dst); else Init(src, dst); }
 public:
  struct ObjectAtATime {};
  KeySwitchSI(const FHESISecKey& s, ObjectAtATime) : context(s.GetContext()), objectAtATime(true) { InitS2(s); }
  KeySwitchSI(const FHESISecKey& s) : context(s.GetContext()) { InitS2(s); }
  KeySwitchSI(const FHESISecKey& src, const FHESISecKey& dst) : context(src.GetContext()) { Init(src, dst); }
  const std::vector<std::vector<DoubleCRT>>& GetRepresentation() const { return keySwitchMatrix; }
  void UpdateRepresentation(const std::vector<std::vector<DoubleCRT>>& rep) { keySwitchMatrix = rep; drop_device_key(); }
  const FHEcontext& GetContext() const { return context; }
  // FHE-SI.cpp:153-209.  The randomness is drawn here in the reference's order (per column: the SampleRandom polynomial, then the
  // Gaussian error); the arithmetic of all columns -- 2 ncol L forward and ncol L inverse row transforms, products, CRT, the shifted
  // key term and the reduction modulo 2^logQ -- is ONE device call (fhesi_keyswitch_init_batch).  InitObjects below is the same
  // computation one DoubleCRT object at a time, as the reference writes it; both give identical matrices (tests/host/test_wire.cpp).
  void Init(const FHESISecKey& src, const FHESISecKey& dst) {
    const std::vector<DoubleCRT>& s = src.GetRepresentation();
    const size_t n = src.GetSize(); const long phim = context.zMstar.phiM(), L = context.numPrimes();
    const long ncol = (long)(context.ndigits * n); const int nl = (int)((context.logQ + 63) / 64);
    std::vector<uint64_t> a((size_t)ncol * phim * nl); std::vector<int64_t> err((size_t)ncol * phim);
    for (long ind = 0; ind < ncol; ++ind) {
      ZZX poly; SampleRandom(poly, context.modulusQ, phim);
      for (long k = 0; k < phim; ++k) coeff(poly, k).to_limbs(&a[((size_t)ind * phim + k) * nl], nl);
      ZZX e; sampleGaussian(e, phim, context.stdev);
      for (long k = 0; k < phim; ++k) err[(size_t)ind * phim + k] = coeff(e, k).to_long();
    }
    fhesi_ksk* k = nullptr;
    ck(fhesi_ksk_create(context.handle(), (int32_t)n, (int32_t)context.ndigits, &k));
    std::vector<const fhesi_dcrt*> hs; for (auto& d : s) hs.push_back(d.handle());
    int rc = fhesi_keyswitch_init_batch(k, hs.data(), (int32_t)n, dst.GetRepresentation()[1].handle(), (int32_t)context.logQ, (int32_t)context.decompSize, a.data(), nl, err.data());
    if (rc) { fhesi_ksk_free(k); ck(rc); }
    const uint64_t* rows = (const uint64_t*)fhesi_ksk_device_ptr(k); const size_t rowWords = (size_t)L * phim;
    keySwitchMatrix.assign(2, std::vector<DoubleCRT>());
    for (int r = 0; r < 2; ++r)
      for (long col = 0; col < ncol; ++col) {
        DoubleCRT d(context);
        ck(fhesi_dev_copy(context.handle(), fhesi_dcrt_device_ptr(d.handle()), rows + ((size_t)r * ncol + col) * rowWords, rowWords * 8));
        keySwitchMatrix[r].push_back(d);
      }
    devKey = std::make_shared<DeviceKey>(k, (int)n, (int)context.ndigits);   // the device object the matrix was generated in serves the fused calls as it is
  }
  // the same matrix with the column randomness drawn on the device (fhesi_keyswitch_init_batch_seeded): column c <-> object index first + c;
  // the public polynomials a from public_seed, the secret errors from seed.  KeySwitchSI(sk, seq) takes its index range from a SeedSequence.
  struct Seeded { uint64_t seed, public_seed, first; };
  KeySwitchSI(const FHESISecKey& s, SeedSequence& seq) : KeySwitchSI(s, Seeded{seq.seed, seq.public_seed, seq.take((uint64_t)(s.GetContext().ndigits * (s.GetRepresentation().size() * 2 - 1)))}) {}
  KeySwitchSI(const FHESISecKey& s, Seeded sd) : context(s.GetContext()) {
    std::vector<DoubleCRT> sKeys = s.GetRepresentation(), tKeys(sKeys.size() * 2 - 1, sKeys[1]);
    tKeys[0] = sKeys[0];
    for (size_t i = 2; i < tKeys.size(); ++i) tKeys[i] *= tKeys[i - 1];
    InitSeeded(tKeys, s, sd);
  }
  void InitSeeded(const std::vector<DoubleCRT>& s, const FHESISecKey& dst, Seeded sd) {
    const size_t n = s.size(); const long phim = context.zMstar.phiM(), L = context.numPrimes(); const long ncol = (long)(context.ndigits * n);
    fhesi_ksk* k = nullptr;
    ck(fhesi_ksk_create(context.handle(), (int32_t)n, (int32_t)context.ndigits, &k));
    std::vector<const fhesi_dcrt*> hs; for (auto& d : s) hs.push_back(d.handle());
    int rc = fhesi_keyswitch_init_batch_seeded(k, hs.data(), (int32_t)n, dst.GetRepresentation()[1].handle(), (int32_t)context.logQ, (int32_t)context.decompSize, sd.seed, sd.public_seed, sd.first);
    if (rc) { fhesi_ksk_free(k); ck(rc); }
    const uint64_t* rows = (const uint64_t*)fhesi_ksk_device_ptr(k); const size_t rowWords = (size_t)L * phim;
    keySwitchMatrix.assign(2, std::vector<DoubleCRT>());
    for (int r = 0; r < 2; ++r) for (long col = 0; col < ncol; ++col) {
      DoubleCRT d(context);
      ck(fhesi_dev_copy(context.handle(), fhesi_dcrt_device_ptr(d.handle()), rows + ((size_t)r * ncol + col) * rowWords, rowWords * 8));
      keySwitchMatrix[r].push_back(d);
    }
    devKey = std::make_shared<DeviceKey>(k, (int)n, (int)context.ndigits);
  }
  void InitObjects(const FHESISecKey& src, const FHESISecKey& dst) {   // the reference's loop, one object at a time
    std::vector<DoubleCRT> s = src.GetRepresentation(); std::vector<ZZX> sCoeff(s.size());
    for (size_t i = 0; i < s.size(); ++i) s[i].toPoly(sCoeff[i]);
    DoubleCRT t = dst.GetRepresentation()[1]; size_t n = src.GetSize();
    std::vector<DoubleCRT> A, b;
    for (size_t i = 0; i < n; ++i)
      for (unsigned j = 0; j < context.ndigits; ++j) {
        ZZX poly; SampleRandom(poly, context.modulusQ, context.zMstar.phiM());
        DoubleCRT a(poly, context), bb = a; a *= -1L; bb *= t;
        ZZX bCoeff; bb.toPoly(bCoeff);
        ZZX err; sampleGaussian(err, context.zMstar.phiM(), context.stdev);
        bCoeff += err; bCoeff += sCoeff[i];
        for (auto& c : sCoeff[i].rep) c <<= (long)(8 * context.decompSize);
        ReduceCoefficients(bCoeff, context.logQ);
        A.push_back(a); b.push_back(DoubleCRT(bCoeff, context));
      }
    drop_device_key();
    keySwitchMatrix.clear(); keySwitchMatrix.push_back(b); keySwitchMatrix.push_back(A);
  }
  void InitS2(const FHESISecKey& s) {   // FHE-SI.cpp:211-227
    std::vector<DoubleCRT> sKeys = s.GetRepresentation(), tKeys(sKeys.size() * 2 - 1, sKeys[1]);
    tKeys[0] = sKeys[0];
    for (size_t i = 2; i < tKeys.size(); ++i) tKeys[i] *= tKeys[i - 1];
    FHESISecKey tensored(s.GetContext()); tensored.UpdateRepresentation(tKeys);
    InitAny(tensored, s);
  }
  KeySwitchSI(const FHESISecKey& s, unsigned k) : context(s.GetContext()) { InitAutomorph(s, k); }     // FHE-SI.h: key for X -> X^k
  void InitAutomorph(const FHESISecKey& s, unsigned k) {   // FHE-SI.cpp:229-239
    std::vector<DoubleCRT> sKeys = s.GetRepresentation();
    FHESISecKey automorphedKey(s.GetContext());           // (its constructor draws a key that is replaced below, as in the reference)
    for (auto& sk : sKeys) sk.automorph((long)k);
    automorphedKey.UpdateRepresentation(sKeys);
    InitAny(automorphedKey, s);
  }
  // ApplyKeySwitch (FHE-SI.cpp:241-260).  The reference's body -- ScaleDown, ByteDecomp, one DoubleCRT per digit polynomial, two DotProducts,
  // toPoly, ReduceCoefficients -- is ApplyKeySwitchObjects below, one object at a time (2 s per call at the metric ring: the digits alone
  // are 66 polynomials through the host).  ApplyKeySwitch itself hands the ciphertext to the fused device call with the matrix resident in
  // HBM as one object (built on first use): the same bits (tests/host/test_wire.cpp compares the two), about 100 times faster.
  void ApplyKeySwitch(Ciphertext& ctxt) const {
    const size_t ncomp = keySwitchMatrix.empty() ? 0 : keySwitchMatrix[0].size() / context.ndigits;
    if (objectAtATime || ncomp < 2 || ctxt.size() != ncomp) { ApplyKeySwitchObjects(ctxt); return; }
    if (LazyCiphertexts()) {
      CtEngine& e = ct_engine(context);
      if (ctxt.scaledUp && !ctxt.terms.empty() && ncomp == 3) {       // a sum of recorded products: multiplied out and key-switched in one call of the next evaluation
        CtRef v = e.ks_sum(std::move(ctxt.terms), device_key_ref());
        ctxt.set_device_value(v);
        return;
      }
      if (!ctxt.scaledUp && ncomp == 2 && ctxt.parts.size() == 2 && ctxt.parts.resident()) {   // after an automorphism (Regression.h:170-172)
        CtRef in = ctxt.parts.value();
        CtRef v = (in->kind == CtValue::AUTO && in->pending()) ? e.auto_ks(in->a, in->s, device_key_ref()) : e.auto_ks(in, 1, device_key_ref());
        ctxt.set_device_value(v);
        return;
      }
    }
    ctxt.materialise();
    fhesi_ctx* h = context.handle(); fhesi_ksk* k = device_key();
    const long n = context.zMstar.phiM(), L = context.numPrimes(); const int nl = (int)((context.logQ + 63) / 64);
    void* out; ck(fhesi_dev_alloc(h, (size_t)2 * n * nl * 8, &out));
    if (ctxt.scaledUp) {
      void* rows; ck(fhesi_dev_alloc(h, ncomp * L * n * 8, &rows));
      for (size_t i = 0; i < ncomp; ++i) ck(fhesi_dev_copy(h, (uint64_t*)rows + i * L * n, fhesi_dcrt_device_ptr(ctxt.tProd[i].handle()), (size_t)L * n * 8));
      int rc = fhesi_apply_key_switch_dev(h, k, (int32_t)context.logQ, (int32_t)context.decompSize, (const uint64_t*)rows, 1, (uint64_t*)out, nl);
      fhesi_dev_free(h, rows);
      if (rc) { fhesi_dev_free(h, out); ck(rc); }
    } else {
      // an unscaled ciphertext (after an automorphism): ScaleDown returns at once (Ciphertext.cpp:195), ByteDecomp takes the positive residues
      std::vector<uint64_t> host(ncomp * n * nl, 0);
      for (size_t i = 0; i < ncomp; ++i) poly_to_limbs(ctxt.parts[i].poly, &host[(i * n) * nl], n, nl);
      void* in; ck(fhesi_dev_alloc(h, host.size() * 8, &in)); ck(fhesi_dev_upload(h, in, host.data(), host.size() * 8));
      int rc = fhesi_ct_automorph_key_switch_dev(h, k, (int32_t)context.logQ, (int32_t)context.decompSize, 1, (const uint64_t*)in, nl, 1, (uint64_t*)out, nl);
      fhesi_dev_free(h, in);
      if (rc) { fhesi_dev_free(h, out); ck(rc); }
    }
    std::vector<uint64_t> res((size_t)2 * n * nl);
    ck(fhesi_dev_download(h, res.data(), out, res.size() * 8)); ck(fhesi_dev_free(h, out));
    ctxt.tProd.clear(); ctxt.scaledUp = false; ctxt.parts.assign(2, CiphertextPart(context));
    for (int r = 0; r < 2; ++r) limbs_to_poly(ctxt.parts[r].poly, &res[(size_t)r * n * nl], n, nl);
  }
  void ApplyKeySwitchObjects(Ciphertext& ctxt) const {   // the reference's loop, one object at a time
    ctxt.ScaleDown(); ctxt.ByteDecomp();
    std::vector<DoubleCRT> bd; for (auto& p : ctxt.parts) bd.push_back(DoubleCRT(p.poly, context));
    std::vector<CiphertextPart> newCtxt(keySwitchMatrix.size(), CiphertextPart(context));
    for (size_t i = 0; i < keySwitchMatrix.size(); ++i) { DoubleCRT dp(context); DotProduct(dp, keySwitchMatrix[i], bd); dp.toPoly(newCtxt[i].poly); ReduceCoefficients(newCtxt[i].poly, context.logQ); }
    ctxt.parts = newCtxt;
  }
  // a[i] *= b[i]; ApplyKeySwitch(a[i]) for every i in ONE device call (fhesi_ct_mul_relin_batch): what a loop over a Matrix<Ciphertext> row or a
  // vector of ciphertexts should call instead of the two statements per object -- the objects cross the host boundary once per batch
  void MulRelinBatch(std::vector<Ciphertext>& a, const std::vector<Ciphertext>& b) const {
    if (a.size() != b.size()) Error("MulRelinBatch: the operand vectors differ in length");
    const size_t count = a.size(); if (!count) return;
    if (LazyCiphertexts() && !objectAtATime) {    // recorded: the two statements per object become one wave at the next evaluation, operands and results in HBM
      for (size_t c = 0; c < count; ++c) {
        if (a[c].isScaledUp() || b[c].isScaledUp() || a[c].size() != 2 || b[c].size() != 2) Error("MulRelinBatch: operands must be unscaled two-part ciphertexts");
        a[c] *= b[c]; ApplyKeySwitch(a[c]);
      }
      return;
    }
    const long n = context.zMstar.phiM(); const int nl = (int)((context.logQ + 63) / 64);
    std::vector<uint64_t> ha(count * 2 * n * nl, 0), hb(ha.size(), 0), ho(ha.size());
    for (size_t c = 0; c < count; ++c) {
      if (a[c].isScaledUp() || b[c].isScaledUp() || a[c].size() != 2 || b[c].size() != 2) Error("MulRelinBatch: operands must be unscaled two-part ciphertexts");
      for (int part = 0; part < 2; ++part) { poly_to_limbs(a[c].parts[part].poly, &ha[((c * 2 + part) * n) * nl], n, nl); poly_to_limbs(b[c].parts[part].poly, &hb[((c * 2 + part) * n) * nl], n, nl); }
    }
    ck(fhesi_ct_mul_relin_batch(context.handle(), device_key(), (int32_t)context.logQ, (uint64_t)context.ModulusP().to_long(), (int32_t)context.decompSize, ha.data(), hb.data(), ho.data(), nl, (int64_t)count));
    for (size_t c = 0; c < count; ++c) for (int part = 0; part < 2; ++part) limbs_to_poly(a[c].parts[part].poly, &ho[((c * 2 + part) * n) * nl], n, nl);
  }
  KeySwitchSI(const KeySwitchSI& o) : context(o.context), keySwitchMatrix(o.keySwitchMatrix), objectAtATime(o.objectAtATime), devKey(o.devKey) {}      // (the device object is immutable once built: shared)
  KeySwitchSI& operator=(const KeySwitchSI& o) { if (&context != &o.context) Error("Incompatible contexts."); keySwitchMatrix = o.keySwitchMatrix; objectAtATime = o.objectAtATime; devKey = o.devKey; return *this; }
 private:
  // keySwitchMatrix as one HBM-resident fhesi_ksk for the fused calls; shared with the recorded operations that will use it (fhesi_engine.h),
  // so a matrix that is replaced or destroyed before they run stays alive until they have
  mutable DeviceKeyRef devKey;
  void drop_device_key() const { devKey.reset(); }
  const DeviceKeyRef& device_key_ref() const {
    if (devKey) return devKey;
    const size_t ncol = keySwitchMatrix[0].size(), ncomp = ncol / context.ndigits; const size_t rowWords = (size_t)context.numPrimes() * context.zMstar.phiM();
    fhesi_ksk* k = nullptr;
    ck(fhesi_ksk_create(context.handle(), (int32_t)ncomp, (int32_t)context.ndigits, &k));
    devKey = std::make_shared<DeviceKey>(k, (int)ncomp, (int)context.ndigits);
    uint64_t* rows = (uint64_t*)fhesi_ksk_device_ptr(k);
    for (int r = 0; r < 2; ++r) for (size_t col = 0; col < ncol; ++col)
      ck(fhesi_dev_copy(context.handle(), rows + ((size_t)r * ncol + col) * rowWords, fhesi_dcrt_device_ptr(keySwitchMatrix[r][col].handle()), rowWords * 8));
    ck(fhesi_ksk_mark_dirty(k));
    return devKey;
  }
  fhesi_ksk* device_key() const { return device_key_ref()->k; }
 public:
  // the matrix as ONE device object, built on first use and shared (the wave executors of fhesi_matrix.h use it instead of a copy of their own)
  const DeviceKeyRef& DeviceMatrix() const { return device_key_ref(); }
};

}  // namespace fhesi
