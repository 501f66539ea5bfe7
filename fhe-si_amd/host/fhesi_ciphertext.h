// fhesi_ciphertext.h -- part of the C++ mirror of the reference's class surface (see fhesi_host.h, which includes the parts in order; not a
// standalone header): CiphertextPart / CtParts / Ciphertext (Ciphertext.h, Ciphertext.cpp) -- recording its operations on device-resident values (fhesi_engine.h) -- and Plaintext (coefficient form).
#pragma once

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

}  // namespace fhesi
