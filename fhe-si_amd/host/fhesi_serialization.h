// fhesi_serialization.h -- the reference's binary wire format (Serialization.h:11-85, Serialization.cpp:3-119,
// FHEcontext::ExportSIContext / ImportSIContext FHEContext.cpp:45-81, key Export / Import FHE-SI.cpp:72-78,137-143,270-276) on the
// mirrored classes, byte for byte: contexts (m, logQ, p, generator, decompSize, the primes WITH their roots -- which pin every
// DoubleCRT row value), keys and ciphertexts written by a real fhe-si build import here and vice versa.
// Raw little-endian struct writes of an LP64 build: unsigned / uint32_t / int32_t 4 bytes, long 8 bytes, bool 1 byte.
// Streams are std::ostream / std::istream (the reference's ofstream / ifstream convert implicitly).
#pragma once
#include <cstdint>
#include <istream>
#include <ostream>
#include <string>

#include "fhesi_matrix.h"

namespace fhesi {

template <typename T> inline void ExportRaw(std::ostream& out, const T& v) { out.write(reinterpret_cast<const char*>(&v), sizeof(T)); }
template <typename T> inline void ImportRaw(std::istream& in, T& v) { in.read(reinterpret_cast<char*>(&v), sizeof(T)); if (!in) Error("Import: unexpected end of stream"); }
// Sizes read from a stream drive allocations, so every importer bounds them by what the active context can hold (a corrupt or
// hostile file must fail with a message, not with a multi-gigabyte resize): integers up to the chain product times 2^64, polynomials
// up to degree m, rows of phi(m) words, at most 64 primes, at most 2^20 vector elements.
inline uint32_t ImportCount(std::istream& in, uint64_t max, const char* what) {
  uint32_t n; ImportRaw(in, n);
  if (n > max) Error((std::string("Import: ") + what + " count in the stream exceeds what the context allows").c_str());
  return n;
}
// The bounds come from the context the stream is imported INTO: the key / ciphertext importers below name it (ImportBounds), plain
// Import calls fall back to activeContext -- on which new DoubleCRT elements are built anyway (Serialization.h:50-58 -> DoubleCRT.cpp:261-266).
inline const FHEcontext*& ImportBoundsContext() { static thread_local const FHEcontext* c = nullptr; return c; }
struct ImportBounds { const FHEcontext* prev; explicit ImportBounds(const FHEcontext& c) : prev(ImportBoundsContext()) { ImportBoundsContext() = &c; } ~ImportBounds() { ImportBoundsContext() = prev; } };
inline const FHEcontext* ImportTargetContext() { return ImportBoundsContext() ? ImportBoundsContext() : activeContext; }
inline uint64_t ImportMaxBytes() { const FHEcontext* c = ImportTargetContext(); return c ? (uint64_t)c->numPrimes() * 8 + 2 * (uint64_t)c->logQ / 8 + 64 : (1u << 16); }
inline uint64_t ImportMaxDegree() { const FHEcontext* c = ImportTargetContext(); return c ? (uint64_t)c->zMstar.M() : (1u << 20); }

// plain-old-data overloads of Serialization.h:29-37
inline void Export(std::ostream& out, uint32_t v) { ExportRaw(out, v); }
inline void Export(std::ostream& out, int32_t v) { ExportRaw(out, v); }
inline void Export(std::ostream& out, long v) { ExportRaw(out, v); }
inline void Import(std::istream& in, uint32_t& v) { ImportRaw(in, v); }
inline void Import(std::istream& in, int32_t& v) { ImportRaw(in, v); }
inline void Import(std::istream& in, long& v) { ImportRaw(in, v); }

// ZZ: NumBytes, sign flag, magnitude little endian (Serialization.cpp:3-27)
inline void Export(std::ostream& out, const ZZ& val) {
  const uint32_t nBytes = (uint32_t)((val.bits() + 7) / 8);
  ExportRaw(out, nBytes);
  const bool neg = val < ZZ();
  ExportRaw(out, neg);
  for (uint32_t i = 0; i < nBytes; ++i) { const unsigned char b = (unsigned char)(val.mag[i / 8] >> (8 * (i % 8))); out.put((char)b); }
}
inline void Import(std::istream& in, ZZ& val) {
  const uint32_t nBytes = ImportCount(in, ImportMaxBytes(), "ZZ byte");
  bool neg; ImportRaw(in, neg);
  val = ZZ();
  val.mag.assign((nBytes + 7) / 8, 0);
  for (uint32_t i = 0; i < nBytes; ++i) { const int b = in.get(); if (b < 0) Error("Import: unexpected end of stream"); val.mag[i / 8] |= (uint64_t)(unsigned char)b << (8 * (i % 8)); }
  val.trim();
  if (neg && !val.is_zero()) val.neg = true;
}
// ZZX: degree (-1 = zero), coefficients (Serialization.cpp:29-54)
inline void Export(std::ostream& out, const ZZX& poly) {
  const int32_t degree = (int32_t)deg(poly);
  ExportRaw(out, degree);
  for (int32_t i = 0; i <= degree; ++i) Export(out, poly.rep[i]);
}
inline void Import(std::istream& in, ZZX& poly) {
  clear(poly);
  int32_t degree; ImportRaw(in, degree);
  if (degree < 0) return;
  if ((uint64_t)degree > ImportMaxDegree()) Error("Import: polynomial degree in the stream exceeds m");
  poly.rep.resize(degree + 1);
  for (int32_t i = 0; i <= degree; ++i) Import(in, poly.rep[i]);
  poly.normalize();
}
// vec_long (Serialization.cpp:83-99)
inline void Export(std::ostream& out, const vec_long& v) { ExportRaw(out, (uint32_t)v.size()); for (long x : v) ExportRaw(out, x); }
inline void Import(std::istream& in, vec_long& v) { const uint32_t n = ImportCount(in, ImportMaxDegree(), "row element"); v.resize(n); for (auto& x : v) ImportRaw(in, x); }

// vector<T> (Serialization.h:41-58)
template <typename T> void Export(std::ostream& out, const std::vector<T>& v);
template <typename T> void Import(std::istream& in, std::vector<T>& v);

// DoubleCRT: index-set size, then (prime index, row) ascending (Serialization.cpp:56-81); the rows come from / go to HBM
inline void Export(std::ostream& out, const DoubleCRT& d) {
  const auto map = d.getMap();
  ExportRaw(out, (uint32_t)map.size());
  for (const auto& kv : map) { ExportRaw(out, (long)kv.first); Export(out, kv.second); }
}
inline void Import(std::istream& in, DoubleCRT& d) {
  std::map<long, vec_long> map;
  const uint32_t size = ImportCount(in, 64, "DoubleCRT row");
  for (uint32_t i = 0; i < size; ++i) { long key; ImportRaw(in, key); if (key < 0 || key >= 64) Error("Import: prime index out of range"); Import(in, map[key]); }
  d.setMap(map);
}
inline void Export(std::ostream& out, const CiphertextPart& part) { Export(out, part.poly); }      // Serialization.cpp:101-107
inline void Import(std::istream& in, CiphertextPart& part) { Import(in, part.poly); }
inline void Export(std::ostream& out, const Ciphertext& ctxt) { Ciphertext copy = ctxt; copy.ScaleDown(); Export(out, copy.parts.host()); }   // :109-114
inline void Import(std::istream& in, Ciphertext& ctxt) { ctxt.Clear(); Import(in, ctxt.parts.host()); }                                        // :116-119

template <typename T> void Export(std::ostream& out, const std::vector<T>& v) { ExportRaw(out, (uint32_t)v.size()); for (const auto& x : v) Export(out, x); }
template <typename T> void Import(std::istream& in, std::vector<T>& v) {
  const uint32_t n = ImportCount(in, 1u << 20, "vector element");
  v.resize(n);
  for (auto& x : v) Import(in, x);
}
// Matrix<T> (Serialization.h:60-85): row count, column count, entries row major
template <typename T> void Export(std::ostream& out, const Matrix<T>& m) {
  ExportRaw(out, (unsigned)m.NumRows()); ExportRaw(out, (unsigned)m.NumCols());
  for (unsigned i = 0; i < m.NumRows(); ++i) for (unsigned j = 0; j < m.NumCols(); ++j) Export(out, m(i, j));
}
template <typename T> void Import(std::istream& in, Matrix<T>& m) {
  const uint32_t r = ImportCount(in, 1u << 16, "matrix row"), c = ImportCount(in, 1u << 16, "matrix column");
  m.Resize(r, c);
  for (unsigned i = 0; i < r; ++i) for (unsigned j = 0; j < c; ++j) Import(in, m(i, j));
}

// FHEcontext::ExportSIContext / ImportSIContext (FHEContext.cpp:45-81)
inline void ExportSIContext(const FHEcontext& c, std::ostream& out) {
  ExportRaw(out, (unsigned)c.zMstar.M()); ExportRaw(out, (unsigned)c.logQ); Export(out, c.ModulusP()); ExportRaw(out, (unsigned)c.Generator()); ExportRaw(out, (unsigned)c.decompSize);
  ExportRaw(out, (uint32_t)c.numPrimes());
  for (long i = 0; i < c.numPrimes(); ++i) { ExportRaw(out, (long)c.ithModulus((unsigned)i).getQ()); ExportRaw(out, (long)c.ithModulus((unsigned)i).getRoot()); }
}
// the reference re-initialises an existing object; the mirror's context binds its chain to the device on first use, so a fresh
// object is built from the stream instead
inline std::unique_ptr<FHEcontext> ImportSIContext(std::istream& in, int device = 0) {
  unsigned m, logQ, generator, decompSize; ZZ p;
  ImportRaw(in, m); ImportRaw(in, logQ); Import(in, p); ImportRaw(in, generator); ImportRaw(in, decompSize);
  if (p.bits() > 32 || p < ZZ(2L)) Error("ImportSIContext: the plaintext modulus must fit the 32-bit `unsigned p` of FHEcontext's constructor (FHEContext.h:105)");
  if (m < 2 || m > (1u << 20) || logQ < 1 || logQ > (1u << 14) || decompSize < 1 || decompSize > 7) Error("ImportSIContext: parameters out of range");
  std::unique_ptr<FHEcontext> c(new FHEcontext(m, logQ, (unsigned)p.to_long(), generator, decompSize, device));
  const uint32_t size = ImportCount(in, 64, "prime");
  for (uint32_t i = 0; i < size; ++i) { long q, root; ImportRaw(in, q); ImportRaw(in, root); c->AddPrime(q, false, root); }
  return c;
}

// keys (FHE-SI.cpp:72-78, 137-143, 270-276)
inline void Export(std::ostream& out, const FHESISecKey& k) { Export(out, k.GetRepresentation()); }
inline void Export(std::ostream& out, const FHESIPubKey& k) { Export(out, k.GetRepresentation()); }
inline void Export(std::ostream& out, const KeySwitchSI& k) { Export(out, k.GetRepresentation()); }
// (new DoubleCRT elements are built on activeContext, as in the reference: Serialization.h:50-58 -> DoubleCRT.cpp:261-266)
inline void Import(std::istream& in, FHESISecKey& k) { ImportBounds b(k.GetContext()); std::vector<DoubleCRT> rep; Import(in, rep); k.UpdateRepresentation(rep); }
inline void Import(std::istream& in, FHESIPubKey& k) { ImportBounds b(k.GetContext()); std::vector<DoubleCRT> rep; Import(in, rep); k.UpdateRepresentation(rep); }
inline void Import(std::istream& in, KeySwitchSI& k) { ImportBounds b(k.GetContext()); std::vector<std::vector<DoubleCRT>> rep; Import(in, rep); k.UpdateRepresentation(rep); }

}  // namespace fhesi
