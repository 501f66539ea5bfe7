// test_addmul.cpp -- counterpart of the reference's Test_AddMul driver (Test_AddMul.cpp:11-171) on the mirrored classes:
// same CLI (`logQ p generator [seed]`), same operation sequence (add; 7-fold add; mul + key switch; square + key switch;
// 9-fold add + key switch, mul, key switch) and the same success predicate (decrypted results equal the plaintext
// products modulo Phi_m).  Every DoubleCRT operation runs on the GPU through the C ABI.  Exit code = number of failed tests.
// `--dump` (with a seed) prints the secret key, both ciphertexts and the mul+keyswitch result as JSON for the parity test
// against tests/golden/ciphertext.json (same documented PRNG stream as the Python model).
#include <cstring>
#include <ctime>
#include <iostream>

#include "../../fhe-si_amd/host/fhesi_host.h"

using namespace fhesi;
namespace fhesi { FHEcontext* activeContext = nullptr; }

static std::vector<long> mul_mod_phi(const std::vector<long>& a, const std::vector<long>& b, const FHEcontext& c, long p) {
  const long m = (long)c.zMstar.M(), n = (long)c.zMstar.phiM();
  if (!(m & (m - 1))) {                 // Phi_m = X^n + 1: plain negacyclic convolution on machine words (the generic path below is cubic in bigints at n = 2^14)
    std::vector<long> out(n, 0);
    for (long i = 0; i < n; ++i) { const long ai = a[i] % p; if (!ai) continue; for (long j = 0; j < n; ++j) { const long k = i + j, t = ai * (b[j] % p) % p; if (k < n) out[k] = (out[k] + t) % p; else out[k - n] = (out[k - n] + p - t) % p; } }
    return out;
  }
  ZZX x, y; for (size_t i = 0; i < a.size(); ++i) SetCoeff(x, (long)i, a[i]); for (size_t i = 0; i < b.size(); ++i) SetCoeff(y, (long)i, b[i]);
  ZZX r = mul(x, y); rem(r, r, c.zMstar.PhimX());
  std::vector<long> out(c.zMstar.phiM(), 0); for (long i = 0; i <= deg(r); ++i) out[i] = rem(r.rep[i], p);
  return out;
}
static void dump_poly(const char* name, const ZZX& p, long n, bool last = false) {
  std::cout << "\"" << name << "\":["; for (long i = 0; i < n; ++i) std::cout << (i ? "," : "") << "\"" << coeff(p, i).str() << "\""; std::cout << "]" << (last ? "" : ",");
}

static bool g_time = false;
static double now_s() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static bool runTest(bool disp, long long seed, unsigned p, FHEcontext& context, bool dump) {
  SetSeed((uint64_t)seed);                                            // Test_AddMul.cpp:15-16
  FHESISecKey secretKey(context);
  FHESIPubKey publicKey(secretKey);
  long phim = context.zMstar.phiM();
  Plaintext ptxt1, ptxt2; ptxt1.message.resize(phim); ptxt2.message.resize(phim);
  for (long i = 0; i < phim; ++i) ptxt1.message[i] = RandomBnd((long)p);
  for (long i = 0; i < phim; ++i) ptxt2.message[i] = RandomBnd((long)p);
  std::vector<long> sum(phim), sumMult(phim);
  for (long i = 0; i < phim; ++i) { sum[i] = (ptxt1.message[i] + ptxt2.message[i]) % p; sumMult[i] = ptxt2.message[i] * 7 % p; }
  std::vector<long> prod = mul_mod_phi(ptxt1.message, ptxt2.message, context, p), prod2 = mul_mod_phi(prod, prod, context, p);
  std::vector<long> sumQuad = mul_mod_phi(prod2, prod2, context, p); for (auto& v : sumQuad) v = v * 9 % p;

  Ciphertext ctxt1(context), ctxt2(context);
  publicKey.Encrypt(ctxt1, ptxt1); publicKey.Encrypt(ctxt2, ptxt2);
  Ciphertext cSum = ctxt1; cSum += ctxt2;
  Ciphertext cSumMult = ctxt2; for (int i = 1; i < 7; ++i) cSumMult += ctxt2;
  double t_mul = now_s();
  Ciphertext cProd = ctxt1; cProd *= ctxt2;
  t_mul = now_s() - t_mul;                                            // (recorded, not run, unless FHESI_EAGER=1: fhesi_engine.h)
  Plaintext resSum, resSumMult, resProd, resProd2, resSumQuad;
  secretKey.Decrypt(resSum, cSum); secretKey.Decrypt(resSumMult, cSumMult);
  double t_kg = now_s();
  KeySwitchSI keySwitch(secretKey);
  t_kg = now_s() - t_kg;
  double t_ks = now_s();
  keySwitch.ApplyKeySwitch(cProd);
  if (g_time) SyncCiphertexts(context);                               // the recorded product + key switch run here
  t_ks = now_s() - t_ks;
  if (g_time) std::cout << "surface timing (one object at a time, first use): operator*= " << t_mul * 1e3 << " ms, ApplyKeySwitch " << t_ks * 1e3 << " ms, KeySwitchSI(sk) " << t_kg * 1e3 << " ms" << std::endl;
  secretKey.Decrypt(resProd, cProd);
  if (dump) {
    ZZX t; secretKey.GetRepresentation()[1].toPoly(t);
    std::cout << "{"; dump_poly("t", t, phim); dump_poly("c1_0", ctxt1[0].poly, phim); dump_poly("c1_1", ctxt1[1].poly, phim);
    { auto& K = keySwitch.GetRepresentation(); for (int r = 0; r < 2; ++r) for (int col = 0; col < 2; ++col) { auto mp = K[r][col].getMap(); std::cout << "\"ksm_" << r << "_" << col << "\":["; bool first = true; for (long v : mp[0]) { std::cout << (first ? "" : ",") << "\"" << v << "\""; first = false; } std::cout << "],"; } }
    dump_poly("c2_0", ctxt2[0].poly, phim); dump_poly("c2_1", ctxt2[1].poly, phim); dump_poly("res_0", cProd[0].poly, phim); dump_poly("res_1", cProd[1].poly, phim, true);
    std::cout << "}" << std::endl;
  }
  double t2 = now_s();
  cProd *= cProd;
  Ciphertext tmp = cProd, cSumQuad = cProd;
  keySwitch.ApplyKeySwitch(cProd);
  if (g_time) SyncCiphertexts(context);                               // a batch of ONE: the result is asked for after every object
  t2 = now_s() - t2;
  if (g_time) std::cout << "surface timing (second use): operator*= + copy + ApplyKeySwitch " << t2 * 1e3 << " ms = " << 1.0 / t2 << " ciphertext-mults/s through the class surface" << std::endl;
  secretKey.Decrypt(resProd2, cProd);
  for (int i = 0; i < 8; ++i) cSumQuad += tmp;
  keySwitch.ApplyKeySwitch(cSumQuad); cSumQuad *= cProd; keySwitch.ApplyKeySwitch(cSumQuad);
  secretKey.Decrypt(resSumQuad, cSumQuad);
  bool success = resSum.message == sum && resSumMult.message == sumMult && resProd.message == prod && resProd2.message == prod2 && resSumQuad.message == sumQuad;
  if (!dump && (disp || !success)) {
    std::cout << "Seed: " << seed << std::endl << std::endl;
    if (resSum.message != sum) std::cout << "Add failed." << std::endl;
    if (resSumMult.message != sumMult) std::cout << "Adding multiple times failed." << std::endl;
    if (resProd.message != prod) std::cout << "Multiply failed." << std::endl;
    if (resProd2.message != prod2) std::cout << "Squaring failed." << std::endl;
    if (resSumQuad.message != sumQuad) std::cout << "Sum and quad failed." << std::endl;
    std::cout << "Test " << (success ? "SUCCEEDED" : "FAILED") << std::endl;
  }
  return success;
}

int main(int argc, char* argv[]) {
  bool dump = false; int ntests = 20, sp_nbits = 60; long m_override = 0;
  std::vector<char*> args;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--dump")) dump = true; else if (!strncmp(argv[i], "--tests=", 8)) ntests = atoi(argv[i] + 8);
    else if (!strncmp(argv[i], "--m=", 4)) m_override = atol(argv[i] + 4);               // a ring other than the reference driver's m = p - 1 (e.g. 32768: the metric ring)
    else if (!strncmp(argv[i], "--sp-nbits=", 11)) sp_nbits = atoi(argv[i] + 11);        // NTL_SP_NBITS of the mirrored NTL build (FHEContext.cpp:92)
    else if (!strcmp(argv[i], "--time")) g_time = true;
    else args.push_back(argv[i]);
  }
  if (args.size() < 3) { std::cout << "usage: test_addmul logQ p generator [seed] [--dump] [--tests=N] [--m=M] [--sp-nbits=B] [--time]" << std::endl; return 1; }
  unsigned logQ = atoi(args[0]), p = atoi(args[1]), g = atoi(args[2]);
  if (!dump) std::cout << "==================================================" << std::endl << "Running add/multiply tests using Brakerski system." << std::endl << "==================================================" << std::endl;
  FHEcontext context(m_override ? (unsigned)m_override : p - 1, logQ, p, g, 3);
  activeContext = &context;
  context.spNbits = sp_nbits;
  context.SetUpSIContext();
  if (!dump) std::cout << "Finished setting up context: m=" << context.zMstar.M() << " phi(m)=" << context.zMstar.phiM() << " logQ=" << logQ << " primes=" << context.numPrimes()
                       << " first prime bits=" << ZZ(context.ithPrime(0)).bits() << " ndigits=" << context.ndigits << std::endl;
  if (args.size() >= 4) {
    long long seed = atoll(args[3]);
    bool res = runTest(true, seed, p, context, dump);
    if (!res) { std::cout << "Failed test with seed " << seed << std::endl; return 1; }
    return 0;
  }
  int failed = 0;
  for (int iter = 0; iter < ntests; ++iter) if (!runTest(false, 1000 + iter, p, context, false)) ++failed;
  if (!failed) std::cout << "All tests SUCCEEDED!" << std::endl; else std::cout << failed << " of " << ntests << " failed." << std::endl;
  return failed;
}
