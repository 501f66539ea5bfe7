// fhesi_doublecrt.h -- part of the C++ mirror of the reference's class surface (see fhesi_host.h, which includes the parts in order; not a
// standalone header): ZZX <-> limb buffers, DoubleCRT (DoubleCRT.h:83-365, DoubleCRT.cpp) and SingleCRT (SingleCRT.h:41-175, SingleCRT.cpp) over fhesi_dcrt / fhesi_scrt handles.
#pragma once

namespace fhesi {

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

}  // namespace fhesi
