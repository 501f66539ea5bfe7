// test_regression.cpp -- counterpart of the reference's Test_Regression driver (Test_Regression.cpp:10-131) on the mirrored
// classes, with coefficient-form plaintexts (slot packing is outside the hot-path scope, see fhe-si_amd/host/fhesi_matrix.h).
//
//   test_regression p generator dim nrows [seed] [--batched-only] [--check=ring|slots|none] [--m=M] [--logQ=B] [--devices=0,1,...] [--overlap=C]
//
// Context as in Test_Regression.cpp:97-125: m = p-1, logQ from the same noise formula, SetUpSIContext(xi).  The data matrix
// (nrows x dim) and the labels are random polynomials over Z_p; Regression::Regress is evaluated three ways
//   (1) object at a time through LMatrix<Ciphertext> (matrix_literal.h) -- the reference's control flow, its statements recorded and evaluated
//       in batches by the mirror's Ciphertext (fhesi_engine.h); with --at-once also with every statement run at once (FHESI_EAGER),
//   (2) in waves on the device (Regression::RegressBatched),
//   (3) in the plaintext ring Z_p[X]/Phi_m with the same LMatrix<T> template,
// and the run succeeds when (1) and (2) give bit-identical ciphertexts and both decrypt to (3).
// --check=slots replaces (3) for sizes where ring arithmetic on the host is too slow (the reference's own d = 8, p = 8423
// configuration): p = 1 mod m there, so Z_p[X]/Phi_m splits into phi(m) copies of Z_p (the plaintext slots of
// PlaintextSpace.cpp) and the decrypted theta / det are compared, at a few slots, with a scalar regression over Z_p on the
// slot values of the inputs.  --m / --logQ override the reference's m = p-1 and noise formula for throughput replays.
// Exit code 0 on success.
#include <chrono>
#include <cstring>
#include <iostream>
#include <string>

#include "matrix_literal.h"
#include "ring_elem.h"

using namespace fhesi;
namespace fhesi { FHEcontext* activeContext = nullptr; }

// scalar of Z_p for the slot-wise check
struct ModP {
  static long p;
  long v = 0;
  ModP() {}
  explicit ModP(long x) : v(((x % p) + p) % p) {}
  ModP& operator+=(const ModP& o) { v = (v + o.v) % p; return *this; }
  ModP& operator*=(const ModP& o) { v = v * o.v % p; return *this; }
  ModP& operator*=(long l) { v = v * (((l % p) + p) % p) % p; return *this; }
};
long ModP::p = 2;

static long eval_at(const std::vector<long>& c, long x, long p) { long r = 0; for (size_t i = c.size(); i-- > 0;) r = (r * x + c[i]) % p; return r; }

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char* argv[]) {
  bool batchedOnly = false, atOnce = false; std::string check = "ring"; unsigned mOverride = 0, logQOverride = 0; int repeat = 1;
  std::vector<int> literalDevices;         // --literal-devices=0,1,...: the recorded literal control flow runs on this group of GPUs (EnableCiphertextGroup)
  std::vector<int> devices;                // --devices=0,1,...: also run the wave evaluator sharded over these GPUs (first = the context's)
  int overlapChunks = 1;                   // --overlap=C: with --devices, every wave in C chunks, the exchange of a chunk overlapped with the next chunk's compute
  std::vector<char*> args;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--batched-only")) batchedOnly = true;
    else if (!strcmp(argv[i], "--at-once")) atOnce = true;
    else if (!strncmp(argv[i], "--check=", 8)) check = argv[i] + 8;
    else if (!strncmp(argv[i], "--repeat=", 9)) repeat = atoi(argv[i] + 9);
    else if (!strncmp(argv[i], "--literal-devices=", 18)) { for (char* t = strtok(argv[i] + 18, ","); t; t = strtok(nullptr, ",")) literalDevices.push_back(atoi(t)); }
    else if (!strncmp(argv[i], "--overlap=", 10)) overlapChunks = atoi(argv[i] + 10);
    else if (!strncmp(argv[i], "--devices=", 10)) { for (char* t = strtok(argv[i] + 10, ","); t; t = strtok(nullptr, ",")) devices.push_back(atoi(t)); }
    else if (!strncmp(argv[i], "--m=", 4)) mOverride = atoi(argv[i] + 4);
    else if (!strncmp(argv[i], "--logQ=", 7)) logQOverride = atoi(argv[i] + 7);
    else args.push_back(argv[i]);
  }
  if (args.size() < 4) { std::cout << "usage: test_regression p generator dim nrows [seed] [--batched-only] [--check=ring|slots|none] [--m=M] [--logQ=B] [--repeat=N]" << std::endl; return 1; }
  const unsigned p = atoi(args[0]), g = atoi(args[1]), dim = atoi(args[2]), nrows = atoi(args[3]);
  const long long seed = args.size() >= 5 ? atoll(args[4]) : 1;
  // Test_Regression.cpp:97-108
  const unsigned n = (p - 1) / 2 - 1, xi = std::max(nrows, dim);
  const double lgQ = 4.5 * std::log((double)n) + std::max(1, (int)dim - 1) * (std::log(1280.0) + 2 * std::log((double)n) + std::log((double)xi));
  const unsigned logQ = logQOverride ? logQOverride : (unsigned)std::ceil(lgQ / std::log(2.0) + 24.7);
  const unsigned m = mOverride ? mOverride : p - 1;
  FHEcontext context(m, logQ, p, g, 3);
  activeContext = &context;
  context.SetUpSIContext(xi);
  context.handle();
  RingElem::init(&context);
  ModP::p = p;
  const long phim = context.zMstar.phiM();
  std::cout << "regression: p=" << p << " m=" << m << " phi(m)=" << phim << " logQ=" << logQ << " primes=" << context.numPrimes() << " ndigits=" << context.ndigits
            << " dim=" << dim << " rows=" << nrows << " seed=" << seed << std::endl;

  SetSeed((uint64_t)seed);
  Regression regress(context);
  std::cout << "automorphism keys: " << regress.AutomorphismExponents().size() << " (k =";
  for (unsigned k : regress.AutomorphismExponents()) std::cout << " " << k;
  std::cout << ")" << std::endl;

  std::vector<std::vector<Plaintext>> ptxtData(nrows, std::vector<Plaintext>(dim));
  std::vector<Plaintext> ptxtLabels(nrows);
  LMatrix<RingElem> plainX(nrows, dim);
  std::vector<RingElem> plainY(nrows);
  for (unsigned i = 0; i < nrows; ++i) {
    for (unsigned j = 0; j < dim; ++j) { ptxtData[i][j].message.resize(phim); for (auto& v : ptxtData[i][j].message) v = RandomBnd((long)p); plainX(i, j).c = ptxtData[i][j].message; }
    ptxtLabels[i].message.resize(phim); for (auto& v : ptxtLabels[i].message) v = RandomBnd((long)p);
    plainY[i].c = ptxtLabels[i].message;
  }
  regress.AddData(ptxtData, ptxtLabels);

  // (3) plaintext ring, same template and the same sequence as Regression::Regress
  std::vector<RingElem> thetaP; RingElem detP;
  if (check == "ring") {
    LMatrix<RingElem> A = plainX; A.Transpose();
    LMatrix<RingElem> last = A * plainY;
    A.MultByTranspose();
    auto sumBatched = [&](RingElem& e) { for (unsigned k : regress.AutomorphismExponents()) { RingElem t = e; t >>= (long)k; e += t; } };
    last.MapAll(sumBatched); A.MapAll(sumBatched);
    if (dim == 1) { detP = A(0, 0); thetaP.assign(1, last(0, 0)); }
    else { A.Invert(detP); A *= last; thetaP.resize(dim); for (unsigned i = 0; i < dim; ++i) thetaP[i] = A(i, 0); }
  }
  // (3') slot-wise: f -> f(zeta^e) is a ring homomorphism onto Z_p for every unit e, and SumBatchedData's product of
  // (1 + sigma_k), k = g, g^2, g^4, ..., is the sum of sigma_{g^j} over j < 2^r
  std::vector<long> slotE; std::vector<std::vector<ModP>> thetaS; std::vector<ModP> detS; long zeta = 0;
  if (check == "slots") {
    if ((p - 1) % m) { std::cout << "--check=slots needs p = 1 mod m" << std::endl; return 1; }
    std::vector<unsigned long> facts; { unsigned long t = m; for (unsigned long f = 2; f * f <= t; ++f) if (t % f == 0) { facts.push_back(f); while (t % f == 0) t /= f; } if (t > 1) facts.push_back(t); }
    for (long a = 2; !zeta; ++a) { long z = (long)PowerMod(a, (p - 1) / m, p); bool ok = true; for (auto f : facts) if (PowerMod(z, m / f, p) == 1) ok = false; if (ok) zeta = z; }
    const size_t r = regress.AutomorphismExponents().size();
    for (long e : {1L, (long)g, (long)(m - 1)}) {
      slotE.push_back(e);
      LMatrix<ModP> A(dim, dim); std::vector<ModP> lastv(dim);
      long ge = 1;                       // g^j mod m
      for (unsigned long j = 0; j < (1ul << r); ++j, ge = (long)(((unsigned long)ge * g) % m)) {
        const long x = (long)PowerMod(zeta, ((unsigned long)e * ge) % m, p);
        std::vector<std::vector<long>> xv(nrows, std::vector<long>(dim)); std::vector<long> yv(nrows);
        for (unsigned i = 0; i < nrows; ++i) { for (unsigned jj = 0; jj < dim; ++jj) xv[i][jj] = eval_at(plainX(i, jj).c, x, p); yv[i] = eval_at(plainY[i].c, x, p); }
        for (unsigned a = 0; a < dim; ++a) {
          for (unsigned b = 0; b < dim; ++b) { long acc = 0; for (unsigned i = 0; i < nrows; ++i) acc = (acc + xv[i][a] * xv[i][b]) % p; A(a, b) += ModP(acc); }
          long acc = 0; for (unsigned i = 0; i < nrows; ++i) acc = (acc + xv[i][a] * yv[i]) % p; lastv[a] += ModP(acc);
        }
      }
      ModP dS; std::vector<ModP> tS(dim);
      if (dim == 1) { dS = A(0, 0); tS[0] = lastv[0]; }
      else { LMatrix<ModP> last(dim, 1); for (unsigned a = 0; a < dim; ++a) last(a, 0) = lastv[a]; A.Invert(dS); A *= last; for (unsigned a = 0; a < dim; ++a) tS[a] = A(a, 0); }
      detS.push_back(dS); thetaS.push_back(tS);
    }
  }

  int failures = 0;
  auto check_fn = [&](const char* what, const std::vector<Ciphertext>& theta, const Ciphertext& det) {
    if (check == "none") return;
    Plaintext tmp;
    regress.GetSecretKey().Decrypt(tmp, det);
    bool ok = true;
    if (check == "ring") {
      ok = tmp.message == detP.c;
      for (unsigned i = 0; i < theta.size(); ++i) { regress.GetSecretKey().Decrypt(tmp, theta[i]); ok = ok && tmp.message == thetaP[i].c; }
    } else {
      for (size_t s = 0; s < slotE.size(); ++s) ok = ok && eval_at(tmp.message, (long)PowerMod(zeta, slotE[s], p), p) == detS[s].v;
      for (unsigned i = 0; i < theta.size(); ++i) {
        regress.GetSecretKey().Decrypt(tmp, theta[i]);
        for (size_t s = 0; s < slotE.size(); ++s) ok = ok && eval_at(tmp.message, (long)PowerMod(zeta, slotE[s], p), p) == thetaS[s][i].v;
      }
    }
    std::cout << what << ": decrypts to the plaintext regression: " << (ok ? "yes" : "NO") << std::endl;
    if (!ok) ++failures;
  };

  std::vector<Ciphertext> thetaB; Ciphertext detB(context);
  double t0 = 0, tB = 0;
  for (int it = 0; it < repeat; ++it) {        // (the first call also sizes the device workspaces)
    t0 = now();
    regress.RegressBatched(thetaB, detB);
    tB = now() - t0;
    if (it + 1 < repeat) std::cout << "batched (call " << it + 1 << "): " << tB << " s" << std::endl;
  }
  std::cout << "batched: " << tB << " s, " << regress.stats.waves << " waves, " << regress.stats.products << " products, " << regress.stats.key_switches
            << " key switches, " << regress.stats.automorph_key_switches << " automorphism key switches" << std::endl;
  check_fn("batched", thetaB, detB);

  if (!devices.empty()) {
    // the same waves sharded over several GPUs (keys RCCL-broadcast from rank 0, wave outputs exchanged): bit-identical results
    std::vector<Ciphertext> thetaG; Ciphertext detG(context);
    double tG = 0;
    for (int it = 0; it < std::max(repeat, 1); ++it) { t0 = now(); regress.RegressBatchedMultiGpu(devices, thetaG, detG); tG = now() - t0; }
    std::cout << "batched on " << devices.size() << " rank(s): " << tG << " s" << std::endl;
    bool same = thetaG.size() == thetaB.size() && detG[0] == detB[0] && detG[1] == detB[1];
    for (unsigned i = 0; same && i < thetaG.size(); ++i) same = thetaG[i][0] == thetaB[i][0] && thetaG[i][1] == thetaB[i][1];
    std::cout << "multi-rank ciphertexts bit-identical to one GPU: " << (same ? "yes" : "NO") << std::endl;
    if (!same) ++failures;
    if (overlapChunks > 1) {
      // ... and with every wave cut into chunks whose exchange overlaps the next chunk's compute (fhesi_comm_exchange_begin / _end)
      std::vector<Ciphertext> thetaO; Ciphertext detO(context);
      double tO = 0;
      for (int it = 0; it < std::max(repeat, 1); ++it) { t0 = now(); regress.RegressBatchedMultiGpu(devices, thetaO, detO, overlapChunks); tO = now() - t0; }
      std::cout << "batched on " << devices.size() << " rank(s), exchange overlapped (" << overlapChunks << " chunks per wave): " << tO << " s" << std::endl;
      long chunked = 0;
      for (auto& line : regress.LastSchedule()) { std::cout << "  schedule: " << line << std::endl; if (line.find("chunks:") != std::string::npos) ++chunked; }
      std::cout << "waves run in chunks: " << chunked << " of " << regress.LastSchedule().size() << std::endl;
      bool sameO = thetaO.size() == thetaB.size() && detO[0] == detB[0] && detO[1] == detB[1];
      for (unsigned i = 0; sameO && i < thetaO.size(); ++i) sameO = thetaO[i][0] == thetaB[i][0] && thetaO[i][1] == thetaB[i][1];
      std::cout << "overlapped-exchange ciphertexts bit-identical to one GPU: " << (sameO ? "yes" : "NO") << std::endl;
      if (!sameO) ++failures;
    }
  }
  if (!batchedOnly) {
    std::vector<Ciphertext> thetaA; Ciphertext detA(context);
    CtEngine& eng = ct_engine(context);
    if (!literalDevices.empty()) { EnableCiphertextGroup(context, literalDevices); std::cout << "recorded operations run on " << eng.group_size() << " GPU rank(s)" << std::endl; }
    const long calls0 = eng.stats.calls, rec0 = eng.stats.recorded;
    t0 = now();
    RegressLiteral(regress, thetaA, detA);
    SyncCiphertexts(context);
    double tA = now() - t0;
    std::cout << "object at a time" << (LazyCiphertexts() ? " (recorded: " + std::to_string(eng.stats.recorded - rec0) + " operations, " + std::to_string(eng.stats.calls - calls0) + " device calls)" : std::string(" (at once)"))
              << ": " << tA << " s" << std::endl;
    check_fn("object at a time", thetaA, detA);
    bool same = thetaA.size() == thetaB.size() && detA[0] == detB[0] && detA[1] == detB[1];
    for (unsigned i = 0; same && i < thetaA.size(); ++i) same = thetaA[i][0] == thetaB[i][0] && thetaA[i][1] == thetaB[i][1];
    std::cout << "ciphertexts of both evaluators bit-identical: " << (same ? "yes" : "NO") << std::endl;
    if (!same) ++failures;
    if (atOnce && LazyCiphertexts()) {
      LazyCiphertexts() = false;
      std::vector<Ciphertext> thetaE; Ciphertext detE(context);
      t0 = now();
      RegressLiteral(regress, thetaE, detE);
      std::cout << "object at a time (at once): " << now() - t0 << " s" << std::endl;
      LazyCiphertexts() = true;
      bool sameE = thetaE.size() == thetaA.size() && detE[0] == detA[0] && detE[1] == detA[1];
      for (unsigned i = 0; sameE && i < thetaE.size(); ++i) sameE = thetaE[i][0] == thetaA[i][0] && thetaE[i][1] == thetaA[i][1];
      std::cout << "recorded and at-once ciphertexts bit-identical: " << (sameE ? "yes" : "NO") << std::endl;
      if (!sameE) ++failures;
    }
  }
  std::cout << (failures ? "Test FAILED" : "Test SUCCEEDED") << std::endl;
  return failures;
}
