// test_statistics.cpp -- counterpart of the reference's Test_Statistics driver (Test_Statistics.cpp:66-244) on the mirrored classes, with
// coefficient-form plaintexts (slot packing is outside the hot-path scope, see statistics_literal.h).
//
//   test_statistics p generator dim nblocks [seed] [--m=M] [--logQ=B]
//
// Context as in Test_Statistics.cpp:199-241: m = p - 1, logQ = ceil((6.5 ln n + ln xi) / ln 2 + 36.1) with n = (p-1)/2 - 1, xi = max(blocks,
// dim), SetUpSIContext(xi).  The data blocks (nblocks x dim) and the block sizes are random polynomials over Z_p in place of the packed
// columns; Statistics::ComputeCovariance runs (a) with the Ciphertext operations recorded and evaluated in batches, (b) with every statement
// run at once (LazyCiphertexts() = false), and (c) in the plaintext ring Z_p[X]/Phi_m with the same LMatrix<T> template and the same sequence
// (mean = SumBatched(sum of the column), cov = SumBatched(X^T X) * n - mu mu^T, n^2); the run succeeds when (a) and (b) give bit-identical
// ciphertexts and decrypt to (c).  ComputeNthMoment(2) is checked the same way.  Exit code = number of failed checks.
#include <chrono>
#include <cstring>
#include <iostream>

#include "statistics_literal.h"
#include "ring_elem.h"

namespace fhesi { FHEcontext* activeContext = nullptr; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char* argv[]) {
  unsigned mOverride = 0, logQOverride = 0; std::vector<char*> args;
  for (int i = 1; i < argc; ++i) {
    if (!strncmp(argv[i], "--m=", 4)) mOverride = atoi(argv[i] + 4);
    else if (!strncmp(argv[i], "--logQ=", 7)) logQOverride = atoi(argv[i] + 7);
    else args.push_back(argv[i]);
  }
  if (args.size() < 4) { std::cout << "usage: test_statistics p generator dim nblocks [seed] [--m=M] [--logQ=B]" << std::endl; return 1; }
  const unsigned p = atoi(args[0]), g = atoi(args[1]), dim = atoi(args[2]), nBlocks = atoi(args[3]);
  const long long seed = args.size() >= 5 ? atoll(args[4]) : 1;
  const unsigned n = (p - 1) / 2 - 1, xi = std::max(nBlocks, dim);                                    // Test_Statistics.cpp:210-217
  const unsigned logQ = logQOverride ? logQOverride : (unsigned)std::ceil((6.5 * std::log((double)n) + std::log((double)xi)) / std::log(2.0) + 36.1);
  FHEcontext context(mOverride ? mOverride : p - 1, logQ, p, g, 3);
  activeContext = &context;
  context.SetUpSIContext(xi);
  context.handle();
  RingElem::init(&context);
  const long phim = context.zMstar.phiM();
  std::cout << "statistics: p=" << p << " m=" << context.zMstar.M() << " phi(m)=" << phim << " logQ=" << logQ << " primes=" << context.numPrimes() << " dim=" << dim << " blocks=" << nBlocks << " seed=" << seed << std::endl;
  SetSeed((uint64_t)seed);
  Statistics stats(context);

  Matrix<Plaintext> blocks; std::vector<Plaintext> blockSizes(nBlocks);
  LMatrix<RingElem> X(nBlocks, dim); std::vector<RingElem> sizes(nBlocks);
  for (unsigned i = 0; i < nBlocks; ++i) {
    std::vector<Plaintext> row(dim);
    for (unsigned j = 0; j < dim; ++j) { row[j].message.resize(phim); for (auto& v : row[j].message) v = RandomBnd((long)p); X(i, j).c = row[j].message; }
    blocks.AddRow(row);
    blockSizes[i].message.resize(phim); for (auto& v : blockSizes[i].message) v = RandomBnd((long)p);
    sizes[i].c = blockSizes[i].message;
  }
  stats.AddData(blocks, blockSizes);

  // (c) the plaintext ring, the same sequence as Statistics.h:48-133
  auto sumBatched = [&](RingElem& e) { for (unsigned k : stats.AutomorphismExponents()) { RingElem t = e; t >>= (long)k; e += t; } };
  RingElem nP = sizes[0]; for (unsigned i = 1; i < nBlocks; ++i) nP += sizes[i];
  std::vector<RingElem> muP(dim), m2P(dim);
  for (unsigned j = 0; j < dim; ++j) {
    muP[j] = X(0, j); m2P[j] = X(0, j); m2P[j] *= X(0, j);
    for (unsigned i = 1; i < nBlocks; ++i) { muP[j] += X(i, j); RingElem t = X(i, j); t *= X(i, j); m2P[j] += t; }
    sumBatched(muP[j]); sumBatched(m2P[j]);
  }
  LMatrix<RingElem> covP = X; covP.Transpose(); covP.MultByTranspose();
  for (unsigned i = 0; i < dim; ++i) for (unsigned j = 0; j < dim; ++j) { sumBatched(covP(i, j)); covP(i, j) *= nP; RingElem t = muP[i]; t *= muP[j]; t *= -1; covP(i, j) += t; }
  RingElem n2P = nP; n2P *= nP;

  int failures = 0;
  auto run = [&](bool recorded, LMatrix<Ciphertext>& cov, std::vector<Ciphertext>& mu, std::vector<Ciphertext>& m2, Ciphertext& encN, Ciphertext& encN2) {
    LazyCiphertexts() = recorded;
    CtEngine& eng = ct_engine(context);
    const long calls0 = eng.stats.calls, rec0 = eng.stats.recorded;
    const double t0 = now();
    stats.ComputeCovariance(cov, mu, encN, encN2);
    Ciphertext denom(context);
    stats.ComputeNthMoment(m2, denom, 2);
    SyncCiphertexts(context);
    std::cout << (recorded ? "recorded" : "at once") << ": ComputeCovariance + second moments in " << now() - t0 << " s";
    if (recorded) std::cout << " (" << eng.stats.recorded - rec0 << " operations recorded, " << eng.stats.calls - calls0 << " device calls)";
    std::cout << std::endl;
    bool ok = true; Plaintext tmp;
    FHESISecKey& sk = stats.GetSecretKey();
    for (unsigned j = 0; j < dim; ++j) { sk.Decrypt(tmp, mu[j]); ok = ok && tmp.message == muP[j].c; sk.Decrypt(tmp, m2[j]); ok = ok && tmp.message == m2P[j].c; }
    sk.Decrypt(tmp, encN); ok = ok && tmp.message == nP.c;
    sk.Decrypt(tmp, encN2); ok = ok && tmp.message == n2P.c;
    for (unsigned i = 0; i < dim; ++i) for (unsigned j = 0; j < dim; ++j) { sk.Decrypt(tmp, cov(i, j)); ok = ok && tmp.message == covP(i, j).c; }
    std::cout << (recorded ? "recorded" : "at once") << ": mean, second moments, N, N^2 and covariance decrypt to the plaintext statistics: " << (ok ? "yes" : "NO") << std::endl;
    if (!ok) ++failures;
    LazyCiphertexts() = true;
  };
  Ciphertext proto(context); LMatrix<Ciphertext> covA(proto), covB(proto); std::vector<Ciphertext> muA, muB, m2A, m2B; Ciphertext nA(context), nB(context), n2A(context), n2B(context);
  run(true, covA, muA, m2A, nA, n2A);
  run(false, covB, muB, m2B, nB, n2B);
  auto eq = [](Ciphertext& a, Ciphertext& b) { return a.size() == 2 && b.size() == 2 && a[0] == b[0] && a[1] == b[1]; };
  bool same = eq(nA, nB) && eq(n2A, n2B);
  for (unsigned j = 0; j < dim; ++j) same = same && eq(muA[j], muB[j]) && eq(m2A[j], m2B[j]);
  for (unsigned i = 0; i < dim; ++i) for (unsigned j = 0; j < dim; ++j) same = same && eq(covA(i, j), covB(i, j));
  std::cout << "recorded and at-once ciphertexts bit-identical: " << (same ? "yes" : "NO") << std::endl;
  if (!same) ++failures;
  std::cout << (failures ? "Test FAILED" : "Test SUCCEEDED") << std::endl;
  return failures;
}
