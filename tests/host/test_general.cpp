// test_general.cpp -- counterpart of the reference's Test_General driver (Test_General.cpp:16-101) on the mirrored classes: p = 2027,
// g = 3, logQ = 120 (m = p - 1 = 2026 = 2 x 1013), the same operation sequence on four ciphertexts
//   c1 *= c2 + key switch;  c0 += const1;  c2 *= const2;  c3 >>= g^rot + automorphism key switch;  c1 *= -1;  c3 *= c2 + key switch;  c0 += -c3
// and the same predicate (every ciphertext decrypts to the plaintext-side result).  The reference's Plaintext packs slots
// (PlaintextSpace, out of scope); here the plaintext side is the same ring arithmetic on the message polynomials modulo (Phi_m, p), with
// `>>=` as the automorphism X -> X^(g^rot) that the ciphertext operation applies.  Besides the predicate, the device forms of
// Ciphertext::operator+=(ZZX) and operator*=(ZZX) are compared bit for bit with the host forms that follow Ciphertext.cpp:29-36,147-156
// literally.  CLI: test_general [seed] [p g logQ].  Exit code = number of failed checks.
#include <cstring>
#include <iostream>

#include "../../fhe-si_amd/host/fhesi_host.h"

using namespace fhesi;
namespace fhesi { FHEcontext* activeContext = nullptr; }

typedef std::vector<long> Msg;
static ZZX to_poly(const Msg& a) { ZZX x; for (size_t i = 0; i < a.size(); ++i) SetCoeff(x, (long)i, a[i]); return x; }
static Msg from_poly(const ZZX& r, const FHEcontext& c, long p) { Msg out(c.zMstar.phiM(), 0); for (long i = 0; i <= deg(r); ++i) out[i] = rem(r.rep[i], p); return out; }
static Msg mul_mod(const Msg& a, const Msg& b, const FHEcontext& c, long p) { ZZX r = mul(to_poly(a), to_poly(b)); rem(r, r, c.zMstar.PhimX()); return from_poly(r, c, p); }
static Msg add_mod(const Msg& a, const Msg& b, long p) { Msg r(a.size()); for (size_t i = 0; i < a.size(); ++i) r[i] = (a[i] + b[i]) % p; return r; }
static Msg sub_mod(const Msg& a, const Msg& b, long p) { Msg r(a.size()); for (size_t i = 0; i < a.size(); ++i) r[i] = ((a[i] - b[i]) % p + p) % p; return r; }
static Msg neg_mod(const Msg& a, long p) { Msg r(a.size()); for (size_t i = 0; i < a.size(); ++i) r[i] = (p - a[i]) % p; return r; }
// a(X^k) modulo Phi_m: exponents modulo m (Phi_m divides X^m - 1), then the remainder
static Msg automorph_mod(const Msg& a, long k, const FHEcontext& c, long p) {
  const long m = (long)c.zMstar.M();
  ZZX r; r.rep.assign(m, ZZ());
  for (size_t i = 0; i < a.size(); ++i) r.rep[(long)((i * (unsigned long)k) % (unsigned long)m)] += ZZ(a[i]);
  r.normalize(); rem(r, r, c.zMstar.PhimX());
  return from_poly(r, c, p);
}
static Msg random_msg(long n, long p) { Msg m(n); for (long i = 0; i < n; ++i) m[i] = RandomBnd(p); return m; }

int main(int argc, char* argv[]) {
  long long seed = argc > 1 ? atoll(argv[1]) : 1;
  unsigned p = argc > 4 ? atoi(argv[2]) : 2027, g = argc > 4 ? atoi(argv[3]) : 3, logQ = argc > 4 ? atoi(argv[4]) : 120;       // Test_General.cpp:22-24
  SetSeed((uint64_t)seed);
  FHEcontext context(p - 1, logQ, p, g);
  activeContext = &context;
  context.SetUpSIContext();
  FHESISecKey secretKey(context);
  FHESIPubKey publicKey(secretKey);
  KeySwitchSI keySwitch(secretKey);
  const long phim = context.zMstar.phiM(), m = (long)context.zMstar.M();
  const long rotAmt = RandomBnd(phim);                         // (the reference draws rand() % numSlots; any power of the generator serves)
  long rotDeg = 1;
  for (long i = 0; i < rotAmt; ++i) rotDeg = rotDeg * (long)context.Generator() % m;
  KeySwitchSI automorphKeySwitch(secretKey, (unsigned)rotDeg);
  std::cout << "m=" << m << " phi(m)=" << phim << " logQ=" << logQ << " primes=" << context.numPrimes() << " rotation exponent " << rotDeg << std::endl;

  Msg p0 = random_msg(phim, p), p1 = random_msg(phim, p), p2 = random_msg(phim, p), p3 = random_msg(phim, p), const1 = random_msg(phim, p), const2 = random_msg(phim, p);
  Plaintext P0, P1, P2, P3; P0.message = p0; P1.message = p1; P2.message = p2; P3.message = p3;
  Ciphertext c0(context), c1(context), c2(context), c3(context);
  publicKey.Encrypt(c0, P0); publicKey.Encrypt(c1, P1); publicKey.Encrypt(c2, P2); publicKey.Encrypt(c3, P3);

  // plaintext side (Test_General.cpp:61-67)
  p1 = mul_mod(p1, p2, context, p);
  p0 = add_mod(p0, const1, p);
  p2 = mul_mod(p2, const2, context, p);
  p3 = automorph_mod(p3, rotDeg, context, p);
  p1 = neg_mod(p1, p);
  p3 = mul_mod(p3, p2, context, p);
  p0 = sub_mod(p0, p3, p);

  int failed = 0;
  // ciphertext side (Test_General.cpp:69-87)
  c1 *= c2;
  keySwitch.ApplyKeySwitch(c1);
  {   // c0 += const1: device form against the host form of Ciphertext.cpp:147-156
    Ciphertext host = c0;
    ZZX sc = to_poly(const1);
    for (auto& c : sc.rep) { c <<= (long)context.logQ; c /= context.ModulusP(); }
    host[0] += sc; ReduceCoefficients(host[0].poly, context.logQ);
    c0 += const1;
    if (!(c0[0] == host[0] && c0[1] == host[1])) { std::cout << "operator+=(ZZX): device and host forms differ" << std::endl; ++failed; }
  }
  {   // c2 *= const2: device form against CiphertextPart::operator*=(ZZX), Ciphertext.cpp:29-36
    Ciphertext host = c2;
    for (unsigned i = 0; i < host.size(); ++i) host[i] *= to_poly(const2);
    c2 *= const2;
    if (!(c2[0] == host[0] && c2[1] == host[1])) { std::cout << "operator*=(ZZX): device and host forms differ" << std::endl; ++failed; }
  }
  c3 >>= rotDeg;
  automorphKeySwitch.ApplyKeySwitch(c3);
  c1 *= -1L;
  c3 *= c2;
  keySwitch.ApplyKeySwitch(c3);
  Ciphertext tmp(c3);
  tmp *= -1L;
  c0 += tmp;
  {   // the scaled-up branches (Ciphertext.cpp:157-159, 250-254): (c1 * c2 + const) and (c1 * c2) * const before the scale-down
    Ciphertext s1(context), s2(context); Plaintext Q1, Q2; Q1.message = random_msg(phim, p); Q2.message = random_msg(phim, p);
    publicKey.Encrypt(s1, Q1); publicKey.Encrypt(s2, Q2);
    Ciphertext prod = s1; prod *= s2;
    Ciphertext prodc = prod; prodc *= const2;                  // tProd[i] *= DoubleCRT(const2)
    keySwitch.ApplyKeySwitch(prodc);
    Plaintext R; secretKey.Decrypt(R, prodc);
    if (R.message != mul_mod(mul_mod(Q1.message, Q2.message, context, p), const2, context, p)) { std::cout << "scaled-up operator*=(ZZX) failed" << std::endl; ++failed; }
  }
  Plaintext pp0, pp1, pp2, pp3;
  secretKey.Decrypt(pp0, c0); secretKey.Decrypt(pp1, c1); secretKey.Decrypt(pp2, c2); secretKey.Decrypt(pp3, c3);
  if (pp0.message != p0) { std::cout << "oops 0" << std::endl; ++failed; }
  if (pp1.message != p1) { std::cout << "oops 1" << std::endl; ++failed; }
  if (pp2.message != p2) { std::cout << "oops 2" << std::endl; ++failed; }
  if (pp3.message != p3) { std::cout << "oops 3" << std::endl; ++failed; }
  std::cout << "All tests finished." << (failed ? "" : " Test SUCCEEDED") << std::endl;
  return failed;
}
