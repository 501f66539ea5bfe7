// zz.h -- minimal arbitrary-precision integer and integer-polynomial types for the host mirror.
//
// The reference builds on NTL's ZZ / ZZX (not installed here, not vendored: SURVEY.md H1).  This header supplies only
// the operations the mirrored classes need, with NTL's semantics where they matter for bit-exactness:
//   * division and remainder by a positive divisor FLOOR (NTL: q = floor(a/b), remainder has the divisor's sign) --
//     Ciphertext::ScaleDown (Ciphertext.cpp:205-210) depends on it for negative coefficients;
//   * operator>> on a negative value shifts the magnitude and keeps the sign (used by Reduce, Util.cpp:14-17);
//   * rem(ZZ, long) is non-negative.
// Sign-magnitude, little-endian 64-bit limbs.  Host-side setup / glue only -- never on the GPU hot path.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace fhesi {

[[noreturn]] inline void Error(const char* msg) {   // role of NTL::Error: print and abort
  std::fprintf(stderr, "%s\n", msg);
  std::abort();
}

class ZZ {
 public:
  typedef unsigned __int128 u128;
  std::vector<uint64_t> mag;   // magnitude, no leading zero limbs
  bool neg = false;

  ZZ() {}
  ZZ(long v) { set(v); }
  ZZ(int v) { set((long)v); }
  ZZ(unsigned long v) { if (v) mag.push_back(v); }
  ZZ(unsigned v) { if (v) mag.push_back(v); }
  static ZZ zero() { return ZZ(); }
  void set(long v) { mag.clear(); neg = v < 0; if (v) mag.push_back(neg ? (uint64_t)(-(v + 1)) + 1 : (uint64_t)v); }
  bool is_zero() const { return mag.empty(); }
  void trim() { while (!mag.empty() && mag.back() == 0) mag.pop_back(); if (mag.empty()) neg = false; }

  static int cmp_mag(const std::vector<uint64_t>& a, const std::vector<uint64_t>& b) {
    if (a.size() != b.size()) return a.size() < b.size() ? -1 : 1;
    for (size_t i = a.size(); i-- > 0;) if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    return 0;
  }
  static void add_mag(std::vector<uint64_t>& a, const std::vector<uint64_t>& b) {
    if (a.size() < b.size()) a.resize(b.size(), 0);
    uint64_t c = 0;
    for (size_t i = 0; i < a.size(); ++i) { u128 s = (u128)a[i] + (i < b.size() ? b[i] : 0) + c; a[i] = (uint64_t)s; c = (uint64_t)(s >> 64); }
    if (c) a.push_back(c);
  }
  static void sub_mag(std::vector<uint64_t>& a, const std::vector<uint64_t>& b) {   // a >= b
    uint64_t br = 0;
    for (size_t i = 0; i < a.size(); ++i) { u128 s = (u128)a[i] - (i < b.size() ? b[i] : 0) - br; a[i] = (uint64_t)s; br = (uint64_t)(s >> 64) & 1; }
  }
  ZZ& operator+=(const ZZ& o) {
    if (neg == o.neg) add_mag(mag, o.mag);
    else if (cmp_mag(mag, o.mag) >= 0) sub_mag(mag, o.mag);
    else { std::vector<uint64_t> t = o.mag; sub_mag(t, mag); mag.swap(t); neg = o.neg; }
    trim(); return *this;
  }
  ZZ operator-() const { ZZ r = *this; if (!r.is_zero()) r.neg = !r.neg; return r; }
  ZZ& operator-=(const ZZ& o) { return *this += -o; }
  ZZ& operator*=(const ZZ& o) {
    if (is_zero() || o.is_zero()) { mag.clear(); neg = false; return *this; }
    std::vector<uint64_t> r(mag.size() + o.mag.size(), 0);
    for (size_t i = 0; i < mag.size(); ++i) {
      uint64_t c = 0;
      for (size_t j = 0; j < o.mag.size(); ++j) { u128 s = (u128)mag[i] * o.mag[j] + r[i + j] + c; r[i + j] = (uint64_t)s; c = (uint64_t)(s >> 64); }
      r[i + o.mag.size()] += c;
    }
    mag.swap(r); neg = neg != o.neg; trim(); return *this;
  }
  ZZ& shl(long k) {
    if (is_zero() || k == 0) return *this;
    size_t w = k / 64; int b = k % 64;
    std::vector<uint64_t> r(mag.size() + w + 1, 0);
    for (size_t i = 0; i < mag.size(); ++i) { r[i + w] |= mag[i] << b; if (b) r[i + w + 1] |= mag[i] >> (64 - b); }
    mag.swap(r); trim(); return *this;
  }
  ZZ& shr(long k) {   // magnitude shift, sign kept (NTL semantics)
    size_t w = k / 64; int b = k % 64;
    if (w >= mag.size()) { mag.clear(); neg = false; return *this; }
    std::vector<uint64_t> r(mag.size() - w, 0);
    for (size_t i = 0; i < r.size(); ++i) { r[i] = mag[i + w] >> b; if (b && i + w + 1 < mag.size()) r[i] |= mag[i + w + 1] << (64 - b); }
    mag.swap(r); trim(); return *this;
  }
  ZZ& operator<<=(long k) { return shl(k); }
  ZZ& operator>>=(long k) { return shr(k); }
  long bits() const { return mag.empty() ? 0 : 64 * (long)(mag.size() - 1) + (64 - __builtin_clzll(mag.back())); }
  bool bit(long i) const { size_t w = i / 64; return w < mag.size() && ((mag[w] >> (i % 64)) & 1); }

  // truncated magnitude division: |a| = qm*|b| + rm
  static void divmod_mag(const ZZ& a, const ZZ& b, ZZ& qm, ZZ& rm) {
    if (b.is_zero()) Error("ZZ: division by zero");
    qm = ZZ(); rm = ZZ();
    if (cmp_mag(a.mag, b.mag) < 0) { rm.mag = a.mag; return; }
    if (b.mag.size() == 1) {
      qm.mag.assign(a.mag.size(), 0);
      uint64_t d = b.mag[0], r = 0;
      for (size_t i = a.mag.size(); i-- > 0;) { u128 cur = ((u128)r << 64) | a.mag[i]; qm.mag[i] = (uint64_t)(cur / d); r = (uint64_t)(cur % d); }
      qm.trim(); if (r) rm.mag.push_back(r);
      return;
    }
    // binary long division (host glue only; sizes are a few dozen limbs)
    long nb = a.bits();
    qm.mag.assign(a.mag.size(), 0);
    for (long i = nb - 1; i >= 0; --i) {
      rm.shl(1);
      if (a.bit(i)) { if (rm.mag.empty()) rm.mag.push_back(1); else rm.mag[0] |= 1; }
      if (cmp_mag(rm.mag, b.mag) >= 0) { sub_mag(rm.mag, b.mag); rm.trim(); qm.mag[i / 64] |= 1ull << (i % 64); }
    }
    qm.trim(); rm.trim();
  }
  // floor division (NTL): remainder takes the sign of the divisor
  static void DivRem(ZZ& q, ZZ& r, const ZZ& a, const ZZ& b) {
    ZZ qm, rm; divmod_mag(a, b, qm, rm);
    bool qneg = a.neg != b.neg;
    if (qneg && !rm.is_zero()) { qm += ZZ(1L); ZZ t; t.mag = b.mag; t -= rm; rm = t; }   // floor adjust
    qm.neg = qneg && !qm.is_zero();
    rm.neg = b.neg && !rm.is_zero();
    q = qm; r = rm;
  }
  ZZ& operator/=(const ZZ& b) { ZZ q, r; DivRem(q, r, *this, b); return *this = q; }
  ZZ& operator%=(const ZZ& b) { ZZ q, r; DivRem(q, r, *this, b); return *this = r; }

  // two's complement export / import with a fixed number of limbs
  void to_limbs(uint64_t* out, int n) const {
    for (int i = 0; i < n; ++i) out[i] = i < (int)mag.size() ? mag[i] : 0;
    if (neg) { uint64_t c = 1; for (int i = 0; i < n; ++i) { uint64_t v = ~out[i] + c; c = (c && v == 0); out[i] = v; } }
  }
  static ZZ from_limbs(const uint64_t* in, int n) {
    ZZ r; r.mag.assign(in, in + n);
    if (n && (in[n - 1] >> 63)) { uint64_t c = 1; for (int i = 0; i < n; ++i) { uint64_t v = ~r.mag[i] + c; c = (c && v == 0); r.mag[i] = v; } r.neg = true; }
    r.trim(); return r;
  }
  long to_long() const { long v = mag.empty() ? 0 : (long)mag[0]; return neg ? -v : v; }
  std::string str() const {
    if (is_zero()) return "0";
    ZZ t = *this; t.neg = false; std::string s;
    ZZ ten19((unsigned long)10000000000000000000ull);
    while (!t.is_zero()) { ZZ q, r; divmod_mag(t, ten19, q, r); char buf[32]; std::snprintf(buf, sizeof buf, q.is_zero() ? "%llu" : "%019llu", (unsigned long long)(r.mag.empty() ? 0 : r.mag[0])); s = std::string(buf) + s; t = q; }
    return (neg ? "-" : "") + s;
  }
};

inline int compare(const ZZ& a, const ZZ& b) {
  if (a.neg != b.neg) return a.neg ? -1 : 1;
  int c = ZZ::cmp_mag(a.mag, b.mag);
  return a.neg ? -c : c;
}
inline bool operator==(const ZZ& a, const ZZ& b) { return compare(a, b) == 0; }
inline bool operator!=(const ZZ& a, const ZZ& b) { return compare(a, b) != 0; }
inline bool operator<(const ZZ& a, const ZZ& b) { return compare(a, b) < 0; }
inline bool operator>(const ZZ& a, const ZZ& b) { return compare(a, b) > 0; }
inline bool operator<=(const ZZ& a, const ZZ& b) { return compare(a, b) <= 0; }
inline bool operator>=(const ZZ& a, const ZZ& b) { return compare(a, b) >= 0; }
inline ZZ operator+(ZZ a, const ZZ& b) { return a += b; }
inline ZZ operator-(ZZ a, const ZZ& b) { return a -= b; }
inline ZZ operator*(ZZ a, const ZZ& b) { return a *= b; }
inline ZZ operator/(ZZ a, const ZZ& b) { return a /= b; }
inline ZZ operator%(ZZ a, const ZZ& b) { return a %= b; }
inline ZZ operator<<(ZZ a, long k) { return a.shl(k); }
inline ZZ operator>>(ZZ a, long k) { return a.shr(k); }
inline ZZ to_ZZ(long v) { return ZZ(v); }
inline int sign(const ZZ& a) { return a.is_zero() ? 0 : (a.neg ? -1 : 1); }
inline long rem(const ZZ& a, long b) {   // NTL rem(ZZ,long): in [0,b) for b > 0
  ZZ q, r; ZZ::DivRem(q, r, a, ZZ(b)); return r.to_long();
}
inline double log(const ZZ& a) { if (a.bits() <= 53) return __builtin_log((double)(a.mag.empty() ? 0 : a.mag[0])); return (a.bits() - 1) * 0.6931471805599453 + __builtin_log((double)a.mag.back() / (double)(1ull << ((a.bits() - 1) % 64))); }

// ---- ZZX: dense integer polynomial (NTL's ZZX surface used by the mirrored classes)
class ZZX {
 public:
  std::vector<ZZ> rep;
  void normalize() { while (!rep.empty() && rep.back().is_zero()) rep.pop_back(); }
  void SetMaxLength(long) {}
  void SetLength(long n) { rep.resize(n); }
};
inline long deg(const ZZX& a) { return (long)a.rep.size() - 1; }
inline ZZ coeff(const ZZX& a, long i) { return (i >= 0 && i < (long)a.rep.size()) ? a.rep[i] : ZZ(); }
inline void SetCoeff(ZZX& a, long i, const ZZ& v) { if (i >= (long)a.rep.size()) a.rep.resize(i + 1); a.rep[i] = v; a.normalize(); }
inline void SetCoeff(ZZX& a, long i, long v) { SetCoeff(a, i, ZZ(v)); }
inline void clear(ZZX& a) { a.rep.clear(); }
inline ZZX& operator+=(ZZX& a, const ZZX& b) { if (a.rep.size() < b.rep.size()) a.rep.resize(b.rep.size()); for (size_t i = 0; i < b.rep.size(); ++i) a.rep[i] += b.rep[i]; a.normalize(); return a; }
inline ZZX& operator-=(ZZX& a, const ZZX& b) { if (a.rep.size() < b.rep.size()) a.rep.resize(b.rep.size()); for (size_t i = 0; i < b.rep.size(); ++i) a.rep[i] -= b.rep[i]; a.normalize(); return a; }
inline ZZX& operator*=(ZZX& a, const ZZ& s) { for (auto& c : a.rep) c *= s; a.normalize(); return a; }
inline ZZX operator*(ZZX a, const ZZ& s) { return a *= s; }
inline ZZX operator*(const ZZ& s, ZZX a) { return a *= s; }
inline bool operator==(const ZZX& a, const ZZX& b) { return a.rep == b.rep; }
inline ZZX mul(const ZZX& a, const ZZX& b) {
  ZZX r; if (a.rep.empty() || b.rep.empty()) return r;
  r.rep.assign(a.rep.size() + b.rep.size() - 1, ZZ());
  for (size_t i = 0; i < a.rep.size(); ++i) if (!a.rep[i].is_zero()) for (size_t j = 0; j < b.rep.size(); ++j) r.rep[i + j] += a.rep[i] * b.rep[j];
  r.normalize(); return r;
}
// remainder modulo a monic integer polynomial
inline void rem(ZZX& r, const ZZX& a, const ZZX& f) {
  ZZX t = a; const long df = deg(f);
  for (long i = deg(t); i >= df; --i) { ZZ c = coeff(t, i); if (c.is_zero()) continue; for (long j = 0; j <= df; ++j) t.rep[i - df + j] -= c * f.rep[j]; }
  if ((long)t.rep.size() > df) t.rep.resize(df);
  t.normalize(); r = t;
}

}  // namespace fhesi
