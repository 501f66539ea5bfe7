// test_lazy.cpp -- TEST HARNESS for the recording form of the mirror's Ciphertext (fhe-si_amd/host/fhesi_engine.h): code written one
// object at a time against the reference's surface (Ciphertext.h:44-97, FHE-SI.h KeySwitchSI::ApplyKeySwitch) must give, with the
// operations recorded and evaluated in batches on ciphertexts that stay in HBM, exactly the ciphertexts it gives when every statement runs
// at once (LazyCiphertexts() = false: the bodies that follow Ciphertext.cpp / FHE-SI.cpp:241-260 statement by statement).
//
//   test_lazy [m logQ p g [seed]] [--devices=0,1,..]   the checks below; exit code = number of failed checks
//   test_lazy --time N [m logQ p g] [--devices=..]     N multiplications + key switches written per object, recorded vs at once (rates)
//   test_lazy --fuzz N [m logQ p g [seed]] [--devices=..]   N random statements over a pool of ciphertexts, recorded and at once side by side
// --devices: the recorded operations run on this group of GPUs (EnableCiphertextGroup; the first = the context's; a repeated device makes a
// loopback group on one GPU) -- every check must come out the same.
#include <chrono>
#include <cstring>
#include <iostream>
#include <sstream>

#include "../../fhe-si_amd/host/fhesi_serialization.h"

using namespace fhesi;
namespace fhesi { FHEcontext* activeContext = nullptr; }

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int failures = 0;
static void expect(bool ok, const char* what) { std::cout << (ok ? "  ok   " : "  FAIL ") << what << std::endl; if (!ok) ++failures; }

// a host-only copy (the device image is dropped by the writable access), so that the at-once bodies start from the same bits
static Ciphertext host_copy(const Ciphertext& c) { Ciphertext r = c; r.parts.host(); return r; }
static bool same(Ciphertext& a, Ciphertext& b) { if (a.isScaledUp() || b.isScaledUp() || a.size() != b.size()) return false; for (unsigned i = 0; i < a.size(); ++i) if (!(a[i] == b[i])) return false; return true; }
static std::vector<Plaintext> random_plaintexts(long count, long n, long p) { std::vector<Plaintext> v(count); for (auto& x : v) { x.message.resize(n); for (auto& c : x.message) c = RandomBnd(p); } return v; }

struct Eager { bool was; Eager() : was(LazyCiphertexts()) { LazyCiphertexts() = false; } ~Eager() { LazyCiphertexts() = was; } };   // statements run at once inside the scope

int main(int argc, char* argv[]) {
  long timeN = 0, fuzzN = 0; std::vector<char*> args; std::vector<int> devices;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--time") && i + 1 < argc) timeN = atol(argv[++i]);
    else if (!strcmp(argv[i], "--fuzz") && i + 1 < argc) fuzzN = atol(argv[++i]);
    else if (!strncmp(argv[i], "--devices=", 10)) { for (char* t = strtok(argv[i] + 10, ","); t; t = strtok(nullptr, ",")) devices.push_back(atoi(t)); }
    else args.push_back(argv[i]);
  }
  const unsigned m = args.size() >= 4 ? atoi(args[0]) : 64, logQ = args.size() >= 4 ? atoi(args[1]) : 100, p = args.size() >= 4 ? atoi(args[2]) : 23, g = args.size() >= 4 ? atoi(args[3]) : 7;
  const long long seed = args.size() >= 5 ? atoll(args[4]) : 1;
  SetSeed((uint64_t)seed);
  FHEcontext context(m, logQ, p, g);
  activeContext = &context;
  context.SetUpSIContext();
  FHESISecKey secretKey(context);
  FHESIPubKey publicKey(secretKey);
  KeySwitchSI keySwitch(secretKey);
  const long n = context.zMstar.phiM();
  CtEngine& eng = ct_engine(context);
  if (!LazyCiphertexts()) { std::cout << "recording is off (FHESI_EAGER): this program compares the recorded form with it, nothing to do" << std::endl << "Test SUCCEEDED" << std::endl; return 0; }
  if (!devices.empty()) EnableCiphertextGroup(context, devices);
  std::cout << "m=" << m << " phi(m)=" << n << " logQ=" << logQ << " primes=" << context.numPrimes() << " recording " << (LazyCiphertexts() ? "on" : "off") << ", " << eng.group_size() << " GPU rank(s)" << std::endl;

  if (fuzzN) {
    // Random statements of the kinds the reference's drivers write, on a pool of ciphertexts that is kept twice: L (recorded, device values)
    // and E (every statement at once, host values).  Noise is irrelevant here -- the two pools must hold the same BITS whenever they are
    // compared -- so products are chained far beyond what would still decrypt.  Exercises the arena (slots freed and reused, growth), the
    // sharing of equal operations, the levelling of deep and wide graphs, evaluation triggered by reads, by the threshold and by copies.
    const int K = 10;
    std::vector<unsigned> ks; { unsigned k = context.Generator(); for (int i = 0; i < 2; ++i) { ks.push_back(k); k = (unsigned)(((unsigned long)k * k) % m); } }
    std::vector<KeySwitchSI> autoKeys; for (unsigned kk : ks) autoKeys.push_back(KeySwitchSI(secretKey, kk));
    std::vector<Plaintext> pts = random_plaintexts(K, n, p);
    std::vector<Ciphertext> L, E;
    publicKey.EncryptBatchSeeded(L, pts, 21, 0);
    for (auto& c : L) E.push_back(host_copy(c));
    SplitMix64 rng((uint64_t)seed * 977 + 5);
    auto pick = [&](int mod) { return (int)(rng.next() % (uint64_t)mod); };
    long compared = 0, mism = 0;
    auto compare_all = [&]() { for (int i = 0; i < K; ++i) { ++compared; Ciphertext a = L[i], b = E[i]; if (!same(a, b)) { ++mism; std::cout << "  mismatch in pool entry " << i << std::endl; } } };
    for (long step = 0; step < fuzzN; ++step) {
      const int op = pick(12), i = pick(K), j = pick(K), a2 = pick(K), b2 = pick(K);
      auto both = [&](auto stmt) { stmt(L); { Eager at_once; stmt(E); } };
      switch (op) {
        case 0: case 1: both([&](std::vector<Ciphertext>& P) { Ciphertext c = P[i]; c *= P[j]; keySwitch.ApplyKeySwitch(c); P[i] = c; }); break;
        case 2: both([&](std::vector<Ciphertext>& P) { P[i] += P[j]; }); break;
        case 3: { const long l = (long)pick(5) - 2; both([&](std::vector<Ciphertext>& P) { P[i] *= l; }); break; }
        case 4: { const int w = pick((int)ks.size()); both([&](std::vector<Ciphertext>& P) { Ciphertext t = P[i]; t >>= (long)ks[w]; autoKeys[w].ApplyKeySwitch(t); P[j] += t; }); break; }
        case 5: { std::vector<long> cst(n); for (auto& v : cst) v = (long)(rng.next() % p); both([&](std::vector<Ciphertext>& P) { P[i] += cst; }); break; }
        case 6: both([&](std::vector<Ciphertext>& P) { Ciphertext t = P[i]; t *= P[j]; Ciphertext u = P[a2]; u *= P[b2]; t += u; Ciphertext sq = P[j]; sq *= sq; t += sq; keySwitch.ApplyKeySwitch(t); P[a2] = t; }); break;
        case 7: both([&](std::vector<Ciphertext>& P) { P[i] = P[j]; }); break;
        case 8: { Plaintext d1, d2; secretKey.Decrypt(d1, L[i]); { Eager at_once; secretKey.Decrypt(d2, E[i]); } ++compared; if (d1.message != d2.message) { ++mism; std::cout << "  decryptions differ at step " << step << std::endl; } break; }
        case 9: { std::vector<Plaintext> one = random_plaintexts(1, n, p); std::vector<Ciphertext> fresh; publicKey.EncryptBatchSeeded(fresh, one, 77, (uint64_t)step); L[i] = fresh[0]; E[i] = host_copy(fresh[0]); break; }
        case 10: eng.flushAt = (pick(2) ? 3 : 8192); break;
        case 11: { std::vector<long> poly(n, 0); poly[0] = 1 + pick(3); poly[pick((int)n)] += 1; both([&](std::vector<Ciphertext>& P) { P[i] *= poly; }); break; }
      }
      if (step % 64 == 63) compare_all();
    }
    compare_all();
    std::cout << "fuzz: " << fuzzN << " statements, " << compared << " comparisons, " << mism << " mismatches; engine: " << eng.stats.recorded << " recorded, " << eng.stats.shared << " shared, "
              << eng.stats.flushes << " evaluations, " << eng.stats.calls << " device calls" << std::endl;
    std::cout << (mism ? "Test FAILED" : "Test SUCCEEDED") << std::endl;
    return mism ? 1 : 0;
  }

  if (timeN) {
    // the per-object statements at the rate a caller of the class surface sees: operands encrypted on the device, results decrypted in one batch
    std::vector<Plaintext> pa = random_plaintexts(timeN, n, p), pb = random_plaintexts(timeN, n, p), dec;
    std::vector<Ciphertext> a, b;
    for (int rep = 0; rep < 3; ++rep) {
      publicKey.EncryptBatchSeeded(a, pa, 11, 0); publicKey.EncryptBatchSeeded(b, pb, 11, (uint64_t)timeN);
      SyncCiphertexts(context);
      const double t0 = now();
      for (long i = 0; i < timeN; ++i) { a[i] *= b[i]; keySwitch.ApplyKeySwitch(a[i]); }
      const double tr = now() - t0;
      SyncCiphertexts(context);
      const double t1 = now() - t0;
      std::cout << "recorded: " << timeN << " x (operator*=, ApplyKeySwitch) in " << t1 << " s (" << tr << " s recording) = " << timeN / t1 << " per second; device calls so far " << eng.stats.calls << std::endl;
    }
    {
      // the same statements with every result asked for before the next object is touched (a batch of one per object: what the reference's
      // object-at-a-time semantics cost when results are read between statements, Test_AddMul.cpp:59-86)
      std::vector<Ciphertext> ea, eb;
      const long oneN = std::min<long>(timeN, 256);
      std::vector<Plaintext> qa(pa.begin(), pa.begin() + oneN), qb(pb.begin(), pb.begin() + oneN);
      publicKey.EncryptBatchSeeded(ea, qa, 12, 0); publicKey.EncryptBatchSeeded(eb, qb, 12, (uint64_t)timeN);
      SyncCiphertexts(context);
      for (int rep = 0; rep < 2; ++rep) {
        if (rep) { publicKey.EncryptBatchSeeded(ea, qa, 13, 0); SyncCiphertexts(context); }
        const double t0 = now();
        for (long i = 0; i < oneN; ++i) { ea[i] *= eb[i]; keySwitch.ApplyKeySwitch(ea[i]); SyncCiphertexts(context); }
        const double t1 = now() - t0;
        std::cout << "result asked after every object: " << oneN << " x (operator*=, ApplyKeySwitch, evaluate) in " << t1 << " s = " << oneN / t1 << " per second" << std::endl;
      }
    }
    secretKey.DecryptBatch(dec, a);
    bool ok = true;
    {   // the plaintext-side product of the first and the last pair
      for (long i : {0L, timeN - 1}) {
        ZZX x, y; for (long j = 0; j < n; ++j) { SetCoeff(x, j, pa[i].message[j]); SetCoeff(y, j, pb[i].message[j]); }
        ZZX r = mul(x, y); rem(r, r, context.zMstar.PhimX());
        for (long j = 0; j < n; ++j) ok = ok && rem(coeff(r, j), (long)p) == dec[i].message[j];
      }
    }
    std::cout << "decrypts to the products: " << (ok ? "yes" : "NO") << std::endl;
    const long eagerN = std::min<long>(timeN, 16);
    {
      Eager at_once;
      std::vector<Ciphertext> ea, eb;
      pa.resize(eagerN); pb.resize(eagerN);
      publicKey.EncryptBatchSeeded(ea, pa, 11, 0); publicKey.EncryptBatchSeeded(eb, pb, 11, (uint64_t)timeN);
      const double t0 = now();
      for (long i = 0; i < eagerN; ++i) { ea[i] *= eb[i]; keySwitch.ApplyKeySwitch(ea[i]); }
      const double t1 = now() - t0;
      std::cout << "at once: " << eagerN << " x (operator*=, ApplyKeySwitch) in " << t1 << " s = " << eagerN / t1 << " per second" << std::endl;
    }
    return ok ? 0 : 1;
  }

  const long N = 12;
  std::vector<Plaintext> pa = random_plaintexts(N, n, p), pb = random_plaintexts(N, n, p);
  std::vector<Ciphertext> a, b;
  publicKey.EncryptBatchSeeded(a, pa, 5, 0); publicKey.EncryptBatchSeeded(b, pb, 5, (uint64_t)N);
  expect(a[0].parts.resident() && !a[0].isScaledUp() && a[0].size() == 2, "EncryptBatchSeeded leaves the ciphertexts in HBM");
  std::vector<Ciphertext> ha, hb; for (long i = 0; i < N; ++i) { ha.push_back(host_copy(a[i])); hb.push_back(host_copy(b[i])); }
  {
    Eager at_once; std::vector<Ciphertext> ea, eb;
    publicKey.EncryptBatchSeeded(ea, pa, 5, 0);
    bool ok = true; for (long i = 0; i < N; ++i) ok = ok && same(ea[i], ha[i]);
    expect(ok, "... with the bits of the host form");
  }

  // (1) the loop every driver of the reference writes: c *= d; ApplyKeySwitch(c)
  {
    std::vector<Ciphertext> c = a;
    const long calls0 = eng.stats.calls, fl0 = eng.stats.flushes;
    for (long i = 0; i < N; ++i) { c[i] *= b[i]; keySwitch.ApplyKeySwitch(c[i]); }
    expect(eng.stats.calls == calls0 && eng.stats.flushes == fl0, "N multiplications + key switches are recorded, nothing runs");
    expect(c[0].size() == 2 && !c[0].isScaledUp(), "a recorded key switch reports an unscaled 2-part ciphertext");
    std::vector<Plaintext> dec; secretKey.DecryptBatch(dec, c);
    expect(eng.stats.calls == calls0 + 1 && eng.stats.flushes == fl0 + 1, "... and run as ONE device call when the results are decrypted");
    bool ok = true;
    {
      Eager at_once;
      for (long i = 0; i < N; ++i) { Ciphertext e = ha[i]; e *= hb[i]; keySwitch.ApplyKeySwitch(e); ok = ok && same(e, c[i]); Plaintext d1; secretKey.Decrypt(d1, e); ok = ok && d1.message == dec[i].message; }
    }
    expect(ok, "bit-identical to the statements run at once; DecryptBatch on device values = Decrypt on host values");
    Plaintext one; secretKey.Decrypt(one, a[3]);
    expect(one.message == pa[3].message, "the operands are unchanged (values are immutable, copies share them)");
  }

  // (2) a row of a matrix product: sum of products while scaled up, one key switch (Matrix.cpp:57-79 + Regression.h:131-134); a *= a
  {
    Ciphertext acc = a[0]; acc *= b[0];
    for (long k = 1; k < 5; ++k) { Ciphertext t = a[k]; t *= b[k]; acc += t; }
    { Ciphertext sq = a[5]; sq *= sq; acc += sq; }
    keySwitch.ApplyKeySwitch(acc);
    Ciphertext ref(context);
    {
      Eager at_once;
      ref = ha[0]; ref *= hb[0];
      for (long k = 1; k < 5; ++k) { Ciphertext t = ha[k]; t *= hb[k]; ref += t; }
      { Ciphertext sq = ha[5]; sq *= sq; ref += sq; }
      keySwitch.ApplyKeySwitch(ref);
    }
    expect(same(acc, ref), "sum of six products + key switch");
  }

  // (3) unscaled algebra on device values: +=, *= long, += constant, *= polynomial, chained through two levels
  {
    std::vector<long> cst(n), poly(n, 0); for (auto& v : cst) v = RandomBnd((long)p); poly[0] = 3; poly[1] = 1; if (n > 5) poly[5] = p - 1;
    auto flow = [&](std::vector<Ciphertext>& x, std::vector<Ciphertext>& y) {
      Ciphertext r = x[0]; r *= y[0]; keySwitch.ApplyKeySwitch(r);
      Ciphertext s = x[1]; s *= -1L; r += s;
      r += cst;
      Ciphertext t = r; t *= y[2]; keySwitch.ApplyKeySwitch(t);
      t *= poly; t += x[3]; t *= 7L;
      return t;
    };
    Ciphertext lz = flow(a, b), ref(context);
    { Eager at_once; ref = flow(ha, hb); }
    expect(same(lz, ref), "key switch, += (-1 * c), += constant, second product, *= polynomial, += c, *= 7");
  }

  // (4) automorphism + its key switch (Regression::SumBatchedData, Regression.h:166-178), recorded and at once
  {
    std::vector<unsigned> ks; unsigned k = context.Generator(); for (int i = 0; i < 3; ++i) { ks.push_back(k); k = (unsigned)(((unsigned long)k * k) % m); }
    std::vector<KeySwitchSI> autoKeys; for (unsigned kk : ks) autoKeys.push_back(KeySwitchSI(secretKey, kk));
    auto sumBatched = [&](Ciphertext& batched) { for (size_t i = 0; i < ks.size(); ++i) { Ciphertext tmp = batched; tmp >>= (long)ks[i]; autoKeys[i].ApplyKeySwitch(tmp); batched += tmp; } };
    std::vector<Ciphertext> lz(a.begin(), a.begin() + 4), ref(ha.begin(), ha.begin() + 4);
    const long calls0 = eng.stats.calls;
    for (auto& c : lz) sumBatched(c);
    std::vector<Plaintext> dec; secretKey.DecryptBatch(dec, lz);
    expect(eng.stats.calls - calls0 == 2 * (long)ks.size(), "four SumBatchedData chains: one automorphism key switch + one addition per step for all four");
    { Eager at_once; for (auto& c : ref) sumBatched(c); }
    bool ok = true; for (size_t i = 0; i < lz.size(); ++i) ok = ok && same(lz[i], ref[i]);
    expect(ok, "SumBatchedData bit-identical to the statements run at once");
    // an automorphism that is looked at before its key switch, and the key switch applied to the evaluated value
    Ciphertext x = a[6], y = ha[6];
    x >>= (long)ks[0]; Plaintext d1; secretKey.Decrypt(d1, x); autoKeys[0].ApplyKeySwitch(x);
    { Eager at_once; y >>= (long)ks[0]; autoKeys[0].ApplyKeySwitch(y); }
    expect(same(x, y), "automorphism evaluated on its own, then key-switched");
  }

  // (5) somebody looks at the rows of a recorded product: ScaleDown, a sum with a multiplied-out product, *= long while scaled up, export
  {
    Ciphertext s = a[7]; s *= b[7];
    Ciphertext t = a[8]; t *= b[8]; t *= 3L;                  // *= long on a scaled-up ciphertext multiplies it out (Ciphertext.cpp:238-241)
    s += t;
    Ciphertext u = s; u.ScaleDown();
    keySwitch.ApplyKeySwitch(s);
    Ciphertext rs(context), ru(context);
    { Eager at_once; rs = ha[7]; rs *= hb[7]; Ciphertext rt = ha[8]; rt *= hb[8]; rt *= 3L; rs += rt; ru = rs; ru.ScaleDown(); keySwitch.ApplyKeySwitch(rs); }
    expect(same(s, rs) && same(u, ru), "recorded product multiplied out on demand: *= long, += tProd, ScaleDown (3 parts), key switch");
    std::stringstream wire; Ciphertext prod = a[9]; prod *= b[9]; keySwitch.ApplyKeySwitch(prod);
    Export(wire, prod); Ciphertext back(context); Import(wire, back);
    Ciphertext rp(context); { Eager at_once; rp = ha[9]; rp *= hb[9]; keySwitch.ApplyKeySwitch(rp); }
    expect(same(back, rp), "Export of a recorded result evaluates it; Import gives the same ciphertext");
  }

  // (6) lifetimes: the key-switching object dies before the recorded operation runs; a small evaluation threshold
  {
    Ciphertext c = a[10]; c *= b[10];
    { KeySwitchSI shortLived(keySwitch); shortLived.ApplyKeySwitch(c); }
    Ciphertext r(context); { Eager at_once; r = ha[10]; r *= hb[10]; keySwitch.ApplyKeySwitch(r); }
    expect(same(c, r), "a recorded key switch keeps its matrix alive");
    const long keep = eng.flushAt; eng.flushAt = 3;
    std::vector<Ciphertext> cs = a; for (long i = 0; i < N; ++i) { cs[i] *= b[(i + 1) % N]; keySwitch.ApplyKeySwitch(cs[i]); }
    eng.flushAt = keep;
    bool ok = true;
    { Eager at_once; for (long i = 0; i < N; ++i) { Ciphertext e = ha[i]; e *= hb[(i + 1) % N]; keySwitch.ApplyKeySwitch(e); ok = ok && same(e, cs[i]); } }
    expect(ok, "evaluation triggered by the recording threshold gives the same ciphertexts");
  }
  // (7) the group is changed while ciphertexts exist: the arena is replicated as it is, results stay the same
  {
    std::vector<Ciphertext> cs(a.begin(), a.begin() + 6);
    EnableCiphertextGroup(context, devices.empty() ? std::vector<int>{context.deviceIndex(), context.deviceIndex()} : std::vector<int>());
    for (long i = 0; i < 6; ++i) { cs[i] *= b[i]; keySwitch.ApplyKeySwitch(cs[i]); cs[i] += a[(i + 1) % 6]; }
    bool ok = true;
    { Eager at_once; for (long i = 0; i < 6; ++i) { Ciphertext e = ha[i]; e *= hb[i]; keySwitch.ApplyKeySwitch(e); e += ha[(i + 1) % 6]; ok = ok && same(e, cs[i]); } }
    expect(ok, devices.empty() ? "switched to a loopback group of two ranks mid-life: same ciphertexts" : "switched back to one GPU mid-life: same ciphertexts");
    EnableCiphertextGroup(context, devices);
  }
  std::cout << "engine: " << eng.stats.recorded << " operations recorded, " << eng.stats.flushes << " evaluations, " << eng.stats.calls << " device calls" << std::endl;
  std::cout << (failures ? "Test FAILED" : "Test SUCCEEDED") << std::endl;
  return failures;
}
