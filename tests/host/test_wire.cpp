// test_wire.cpp -- the reference's binary wire format (fhesi_serialization.h) on the mirrored classes.
//
//   test_wire logQ p generator seed outdir
//
// Same key / plaintext / ciphertext sequence as test_addmul's runTest (Test_AddMul.cpp:15-60) from the documented PRNG, then
//   outdir/context.bin  FHEcontext::ExportSIContext            outdir/sk.bin   FHESISecKey::Export
//   outdir/pk.bin       FHESIPubKey::Export                    outdir/ksk.bin  KeySwitchSI::Export (s^2 -> s matrix)
//   outdir/c1.bin       Export(Ciphertext) of a fresh ciphertext
//   outdir/prod.bin     Export(Ciphertext) of the scaled-up product c1 * c2 (Serialization.cpp:109-114 scales it down first)
// The parity test compares these files byte for byte with the Python model's rendering of the same objects.  The program then
// imports everything into a second context built by ImportSIContext and checks that the imported keys decrypt the imported
// ciphertexts and that the imported key-switch matrix relinearises the product.  Exit code 0 on success.
#include <fstream>
#include <iostream>
#include <sstream>

#include "../../fhe-si_amd/host/fhesi_serialization.h"
#include "host_helpers_literal.h"

using namespace fhesi;
namespace fhesi { FHEcontext* activeContext = nullptr; }

static std::vector<long> mul_mod_phi(const std::vector<long>& a, const std::vector<long>& b, const FHEcontext& c, long p) {
  ZZX x, y; for (size_t i = 0; i < a.size(); ++i) SetCoeff(x, (long)i, a[i]); for (size_t i = 0; i < b.size(); ++i) SetCoeff(y, (long)i, b[i]);
  ZZX r = mul(x, y); rem(r, r, c.zMstar.PhimX());
  std::vector<long> out(c.zMstar.phiM(), 0); for (long i = 0; i <= deg(r); ++i) out[i] = rem(r.rep[i], p);
  return out;
}

int main(int argc, char* argv[]) {
  if (argc < 6) { std::cout << "usage: test_wire logQ p generator seed outdir" << std::endl; return 1; }
  const unsigned logQ = atoi(argv[1]), p = atoi(argv[2]), g = atoi(argv[3]);
  const long long seed = atoll(argv[4]);
  const std::string dir = argv[5];
  auto path = [&](const char* f) { return dir + "/" + f; };
  std::vector<long> m1, m2;
  {
    FHEcontext context(p - 1, logQ, p, g, 3);
    activeContext = &context;
    context.SetUpSIContext();
    SetSeed((uint64_t)seed);
    FHESISecKey secretKey(context);
    FHESIPubKey publicKey(secretKey);
    const long phim = context.zMstar.phiM();
    Plaintext ptxt1, ptxt2; ptxt1.message.resize(phim); ptxt2.message.resize(phim);
    for (long i = 0; i < phim; ++i) ptxt1.message[i] = RandomBnd((long)p);
    for (long i = 0; i < phim; ++i) ptxt2.message[i] = RandomBnd((long)p);
    m1 = ptxt1.message; m2 = ptxt2.message;
    Ciphertext c1(context), c2(context);
    publicKey.Encrypt(c1, ptxt1); publicKey.Encrypt(c2, ptxt2);
    KeySwitchSI keySwitch(secretKey);
    Ciphertext prod = c1; prod *= c2;
    { std::ofstream f(path("context.bin"), std::ios::binary); ExportSIContext(context, f); }
    { std::ofstream f(path("sk.bin"), std::ios::binary); Export(f, secretKey); }
    { std::ofstream f(path("pk.bin"), std::ios::binary); Export(f, publicKey); }
    { std::ofstream f(path("ksk.bin"), std::ios::binary); Export(f, keySwitch); }
    { std::ofstream f(path("c1.bin"), std::ios::binary); Export(f, c1); }
    { std::ofstream f(path("c2.bin"), std::ios::binary); Export(f, c2); }
    { std::ofstream f(path("prod.bin"), std::ios::binary); Export(f, prod); }
    activeContext = nullptr;
  }
  // ---- import side: nothing survives from above except the files
  std::ifstream fc(path("context.bin"), std::ios::binary);
  std::unique_ptr<FHEcontext> ctx2 = ImportSIContext(fc);
  activeContext = ctx2.get();
  SetSeed(12345);
  FHESISecKey sk2(*ctx2);                 // freshly drawn keys, replaced by the imported representation
  FHESIPubKey pk2(sk2);
  KeySwitchSI ks2(sk2);
  { std::ifstream f(path("sk.bin"), std::ios::binary); Import(f, sk2); }
  { std::ifstream f(path("pk.bin"), std::ios::binary); Import(f, pk2); }
  { std::ifstream f(path("ksk.bin"), std::ios::binary); Import(f, ks2); }
  Ciphertext c1(*ctx2), c2(*ctx2), prod(*ctx2);
  { std::ifstream f(path("c1.bin"), std::ios::binary); Import(f, c1); }
  { std::ifstream f(path("c2.bin"), std::ios::binary); Import(f, c2); }
  { std::ifstream f(path("prod.bin"), std::ios::binary); Import(f, prod); }
  int failures = 0;
  Plaintext res;
  sk2.Decrypt(res, c1);
  if (res.message != m1) { std::cout << "imported key does not decrypt imported ciphertext 1" << std::endl; ++failures; }
  sk2.Decrypt(res, c2);
  if (res.message != m2) { std::cout << "imported key does not decrypt imported ciphertext 2" << std::endl; ++failures; }
  // a fresh encryption under the imported public key
  Plaintext pt; pt.message = m2; Ciphertext fresh(*ctx2); pk2.Encrypt(fresh, pt);
  sk2.Decrypt(res, fresh);
  if (res.message != m2) { std::cout << "imported public key does not encrypt for the imported secret key" << std::endl; ++failures; }
  // batched Encrypt / Decrypt on the device give what the per-object methods give from the same PRNG state
  {
    std::vector<Plaintext> pts(3); pts[0].message = m1; pts[1].message = m2; pts[2].message = m1;
    SetSeed(777); std::vector<Ciphertext> one(3, Ciphertext(*ctx2)); for (int i = 0; i < 3; ++i) pk2.EncryptObjects(one[i], pts[i]);
    SetSeed(777); std::vector<Ciphertext> many; pk2.EncryptBatch(many, pts);
    bool same = many.size() == 3; for (int i = 0; same && i < 3; ++i) same = one[i][0] == many[i][0] && one[i][1] == many[i][1];
    if (!same) { std::cout << "EncryptBatch differs from Encrypt" << std::endl; ++failures; }
    std::vector<Plaintext> dec; sk2.DecryptBatch(dec, many);
    if (dec.size() != 3 || dec[0].message != m1 || dec[1].message != m2 || dec[2].message != m1) { std::cout << "DecryptBatch does not invert EncryptBatch" << std::endl; ++failures; }
  }
  // KeySwitchSI::Init as one device call (fhesi_keyswitch_init_batch) equals the reference's object-at-a-time loop from the same PRNG state
  {
    SetSeed(4242); KeySwitchSI kb(sk2);
    SetSeed(4242); KeySwitchSI ko(sk2, KeySwitchSI::ObjectAtATime());
    const auto &A = kb.GetRepresentation(), &B = ko.GetRepresentation();
    bool same = A.size() == 2 && B.size() == 2 && A[0].size() == B[0].size() && A[1].size() == B[1].size();
    for (int r = 0; same && r < 2; ++r) for (size_t c = 0; same && c < A[r].size(); ++c) same = A[r][c] == B[r][c];
    if (!same) { std::cout << "batched KeySwitchSI::Init differs from the object-at-a-time loop" << std::endl; ++failures; }
  }
  // ApplyKeySwitch through the fused device call (the form that runs) equals the reference's object-at-a-time body, on a scaled-up product
  // and on an unscaled three-part ciphertext; MulRelinBatch gives the same ciphertexts again
  {
    Ciphertext x = c1; x *= c2; Ciphertext y = c1; y.MulObjects(c2);          // (operator*= is one device call; MulObjects the reference's loop)
    ks2.ApplyKeySwitch(x); ks2.ApplyKeySwitchObjects(y);
    if (!(x.size() == 2 && y.size() == 2 && x[0] == y[0] && x[1] == y[1])) { std::cout << "ApplyKeySwitch (device) differs from the object-at-a-time body on a scaled-up ciphertext" << std::endl; ++failures; }
    Ciphertext u = prod, v = prod;
    ks2.ApplyKeySwitch(u); ks2.ApplyKeySwitchObjects(v);
    if (!(u.size() == 2 && v.size() == 2 && u[0] == v[0] && u[1] == v[1])) { std::cout << "ApplyKeySwitch (device) differs from the object-at-a-time body on an unscaled ciphertext" << std::endl; ++failures; }
    std::vector<Ciphertext> va{c1, c2, c1}, vb{c2, c2, c1};
    ks2.MulRelinBatch(va, vb);
    Ciphertext z = c2; z *= c2; ks2.ApplyKeySwitch(z);
    if (!(va[0][0] == x[0] && va[0][1] == x[1] && va[1][0] == z[0] && va[1][1] == z[1])) { std::cout << "MulRelinBatch differs from operator*= + ApplyKeySwitch" << std::endl; ++failures; }
  }
  // randomness drawn on the device (csrc/philox.h): seeded encryptions decrypt to their messages, do not depend on how a batch is split, and
  // a seeded key-switch matrix relinearises their product
  {
    std::vector<Plaintext> pts(3); pts[0].message = m1; pts[1].message = m2; pts[2].message = m1;
    std::vector<Ciphertext> all, tail;
    pk2.EncryptBatchSeeded(all, pts, 99, 10);
    std::vector<Plaintext> last(pts.begin() + 2, pts.end());
    pk2.EncryptBatchSeeded(tail, last, 99, 12);
    if (!(tail[0][0] == all[2][0] && tail[0][1] == all[2][1])) { std::cout << "seeded encryption depends on the batch split" << std::endl; ++failures; }
    if (all[0][0] == all[2][0]) { std::cout << "seeded encryptions of one message under different indices coincide" << std::endl; ++failures; }
    std::vector<Plaintext> dec; sk2.DecryptBatch(dec, all);
    if (dec.size() != 3 || dec[0].message != m1 || dec[1].message != m2 || dec[2].message != m1) { std::cout << "seeded encryptions do not decrypt" << std::endl; ++failures; }
    SeedSequence seq(4711, 815, 1000);                      // one counter for every object drawn from this stream
    KeySwitchSI kseed(sk2, seq);
    if (seq.used() != 1000 + 3 * ctx2->ndigits) { std::cout << "SeedSequence did not advance by the matrix's columns" << std::endl; ++failures; }
    Ciphertext pr = all[0]; pr *= all[1]; kseed.ApplyKeySwitch(pr);
    Plaintext r; sk2.Decrypt(r, pr);
    if (r.message != mul_mod_phi(m1, m2, *ctx2, (long)p)) { std::cout << "seeded key-switch matrix does not relinearise" << std::endl; ++failures; }
  }
  // the imported product has 3 parts (it was scaled down on export): relinearise it with the imported matrix
  if (prod.parts.size() != 3) { std::cout << "imported product has " << prod.parts.size() << " parts" << std::endl; ++failures; }
  ks2.ApplyKeySwitch(prod);
  sk2.Decrypt(res, prod);
  if (res.message != mul_mod_phi(m1, m2, *ctx2, (long)p)) { std::cout << "imported key-switch matrix does not relinearise the imported product" << std::endl; ++failures; }
  // re-export equals the file
  { std::ostringstream os; Export(os, ks2); std::ifstream f(path("ksk.bin"), std::ios::binary); std::string orig((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (os.str() != orig) { std::cout << "re-export of the key-switch matrix differs" << std::endl; ++failures; } }
  // odds and ends of the class surface (IndexSet.h:94-123, FHEContext.cpp:118-141, FHEContext.h:152-189, NumbTh.cpp:361-375, Ciphertext.cpp:220-224)
  {
    IndexSet s1(0, 5), s2(4, 9), s3(6, 9);
    IndexSet r = s1; r.retain(s2);
    if (!(r == IndexSet(4, 5)) || s1.disjointFrom(s2) || !s1.disjointFrom(s3) || !disjoint(s1, s3)) { std::cout << "IndexSet::retain / disjointFrom wrong" << std::endl; ++failures; }
    FHEcontext byNumber(ctx2->zMstar.M(), logQ, p, g);
    const double lg = AddPrimesByNumber(byNumber, 3, 1000);
    const long twoM = 2 * (long)byNumber.zMstar.M();
    bool ok = byNumber.numPrimes() == 3 && byNumber.ithPrime(0) > 1000 && byNumber.ithPrime(0) < byNumber.ithPrime(1) && byNumber.ithPrime(1) < byNumber.ithPrime(2);
    for (unsigned i = 0; i < 3 && ok; ++i) ok = byNumber.ithPrime(i) % twoM == 1 && ProbPrime((uint64_t)byNumber.ithPrime(i));
    ok = ok && std::fabs(lg - byNumber.logOfProduct(byNumber.ctxtPrimes)) < 1e-9 && byNumber.isZeroDivisor(ZZ(byNumber.ithPrime(1)) * ZZ(7L)) && !byNumber.isZeroDivisor(ZZ(7L));
    if (!ok) { std::cout << "AddPrimesByNumber / logOfProduct / isZeroDivisor wrong" << std::endl; ++failures; }
    { ZZX zz; for (long i = 0; i < 6; ++i) SetCoeff(zz, i, ZZ((long)(i * 37 - 90)));         // Util.cpp:33-43 against Reduce for a power of two, and an odd modulus by hand
      ZZX a1 = zz, a2 = zz; ReduceCoefficientsSlow(a1, ZZ(64L)); ReduceCoefficients(a2, 6);
      ZZX a3 = zz; ReduceCoefficientsSlow(a3, 7u, true);
      bool okR = true; for (long i = 0; i < 6; ++i) { okR = okR && (coeff(a1, i) == coeff(a2, i) || (coeff(a1, i) == ZZ(32L) && coeff(a2, i) == ZZ(-32L))) && coeff(a3, i) == ZZ((long)((((i * 37 - 90) % 7) + 7) % 7)); }
      std::vector<long> v1{2, 3}, v2{5, 7, 11}, tp; TensorProduct(tp, v1, v2);
      okR = okR && tp == std::vector<long>{10, 14, 22, 15, 21, 33} && ComputeLog(1u) == 0 && ComputeLog(1024ul) == 10 && ComputeLog(1025ul) == 10;
      if (!okR) { std::cout << "ReduceCoefficientsSlow / TensorProduct / ComputeLog wrong" << std::endl; ++failures; } }
    {   // NumbTh.h:43-66,76-79,202
      auto coeffs = [](const ZZX& f) { std::vector<long> v; for (auto& c : f.rep) v.push_back(c.to_long()); return v; };
      bool okN = Cyclotomic((int)ctx2->zMstar.M()) == ctx2->zMstar.PhimX() && coeffs(Cyclotomic(16)) == std::vector<long>{1, 0, 0, 0, 0, 0, 0, 0, 1}
                 && coeffs(Cyclotomic(15)) == std::vector<long>{1, -1, 0, 1, -1, 1, 0, -1, 1} && deg(Cyclotomic(105)) == 48 && largestCoeff(Cyclotomic(105)) == ZZ(2L);
      okN = okN && phi_N(8422) == 4210 && phi_N(105) == 48 && mobius(30) == -1 && mobius(12) == 0 && mobius(35) == 1 && ord(48, 2) == 4 && ord(48, 5) == 0;
      const int gr = primroot(23, 22); std::set<long> seen; long x = 1; for (int i = 0; i < 22; ++i) { x = x * gr % 23; seen.insert(x); }
      okN = okN && seen.size() == 22;
      std::vector<long> fs; factorize(fs, 8422); okN = okN && fs == std::vector<long>{2, 4211};
      ZZX in, o1, o2, o3; for (long i = 0; i < 5; ++i) SetCoeff(in, i, ZZ((long)(i * 5 - 9)));        // -9 -4 1 6 11
      PolyRed(o1, in, 7); PolyRed(o2, in, 7, true); PolyRed(o3, in, 2);
      okN = okN && coeffs(o1) == std::vector<long>{-2, 3, 1, -1, -3} && coeffs(o2) == std::vector<long>{5, 3, 1, 6, 4} && coeffs(o3) == std::vector<long>{-1, 0, 1, 0, 1};
      std::vector<long> av{3, 9, -2, 9}; okN = okN && argmax(av) == 1 && argmin(av) == 2;
      if (!okN) { std::cout << "Cyclotomic / phi_N / mobius / ord / primroot / factorize / PolyRed / argmax wrong" << std::endl; ++failures; }
    }
    DoubleCRT small(*ctx2); small.sampleSmall(); ZZX sp; small.toPoly(sp);
    bool tern = true; long nz = 0; for (auto& cf : sp.rep) { tern = tern && cf.bits() <= 1; if (!cf.is_zero()) ++nz; }
    if (!tern) { std::cout << "sampleSmall left a coefficient outside {-1, 0, 1}" << std::endl; ++failures; }
    // SetTensorRepresentation: the rows of a product handed to another Ciphertext object relinearise to the same ciphertext
    const bool was = LazyCiphertexts(); LazyCiphertexts() = false;
    Ciphertext x(*ctx2), y(*ctx2); Plaintext px, py; px.message = m1; py.message = m2; pk2.EncryptObjects(x, px); pk2.EncryptObjects(y, py);
    Ciphertext direct = x; direct.MulObjects(y); ks2.ApplyKeySwitch(direct);
    // (the tensor product formed by hand on DoubleCRT objects, Ciphertext.cpp:169-186)
    std::vector<DoubleCRT> c1, c2, rows(3, DoubleCRT(*ctx2));
    for (unsigned i = 0; i < 2; ++i) { c1.push_back(DoubleCRT(x[i].poly * ctx2->ModulusP(), *ctx2)); c2.push_back(DoubleCRT(y[i].poly, *ctx2)); }
    for (unsigned i = 0; i < 2; ++i) for (unsigned j = 0; j < 2; ++j) { DoubleCRT t = c1[i]; t *= c2[j]; rows[i + j] += t; }
    Ciphertext handed(*ctx2); handed.SetTensorRepresentation(rows);
    bool okT = handed.isScaledUp() && handed.size() == 3 && rows.empty();
    ks2.ApplyKeySwitch(handed);
    okT = okT && handed[0] == direct[0] && handed[1] == direct[1];
    if (!okT) { std::cout << "SetTensorRepresentation: rows handed over do not relinearise to the product" << std::endl; ++failures; }
    LazyCiphertexts() = was;
  }
  std::cout << (failures ? "wire roundtrip FAILED" : "wire roundtrip ok") << std::endl;
  activeContext = nullptr;
  return failures;
}
