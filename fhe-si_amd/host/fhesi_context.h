// fhesi_context.h -- part of the C++ mirror of the reference's class surface (see fhesi_host.h, which includes the parts in order; not a
// standalone header): IndexSet (IndexSet.h:26-127), PAlgebra (PAlgebra.h:53-88), Cmodulus (CModulus.h:42-170) and FHEcontext (FHEContext.h:40-205, FHEContext.cpp): the prime chain and the handle of the HIP context.
#pragma once

namespace fhesi {

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
      // (a library of another ABI revision would link and shift arguments silently: include/fhesi_hip.h, FHESI_ABI_VERSION)
      if (fhesi_abi_version() != FHESI_ABI_VERSION) Error("libfhesi_hip.so and fhesi_hip.h disagree on the ABI revision: rebuild");
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

}  // namespace fhesi
