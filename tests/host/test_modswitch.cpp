// test_modswitch.cpp -- DoubleCRT::addPrimesAndScale / scaleDownToSet (DoubleCRT.cpp:162-208, 518-558) on the mirrored
// class: prints the resulting rows as JSON; tests/test_gpu_host_mirror.py compares them with the Python model.
#include <iostream>
#include "../../fhe-si_amd/host/fhesi_host.h"
using namespace fhesi;
namespace fhesi { FHEcontext* activeContext = nullptr; }
static void dump(const char* name, const DoubleCRT& d, bool last = false) {
  auto m = d.getMap(); std::cout << "\"" << name << "\":{"; bool f1 = true;
  for (auto& kv : m) { std::cout << (f1 ? "" : ",") << "\"" << kv.first << "\":["; f1 = false; bool f2 = true; for (long v : kv.second) { std::cout << (f2 ? "" : ",") << "\"" << (unsigned long)v << "\""; f2 = false; } std::cout << "]"; }
  std::cout << "}" << (last ? "" : ",");
}
int main(int argc, char** argv) {
  unsigned m = argc > 1 ? atoi(argv[1]) : 64, logQ = 100, p = 23;
  FHEcontext context(m, logQ, p, 3, 3); activeContext = &context; context.SetUpSIContext();
  SetSeed(99);
  ZZX poly; SampleRandom(poly, context.modulusQ, context.zMstar.phiM());
  long L = context.numPrimes();
  DoubleCRT a(poly, context, IndexSet(0, 1));            // two primes, then grow by the rest with scaling
  a.addPrimesAndScale(IndexSet(2, L - 1));
  DoubleCRT b(poly, context);                            // all primes, then switch down to the first two
  b.scaleDownToSet(IndexSet(0, 1));
  std::cout << "{\"L\":" << L << ",\"poly\":["; for (unsigned i = 0; i < context.zMstar.phiM(); ++i) std::cout << (i ? "," : "") << "\"" << coeff(poly, i).str() << "\""; std::cout << "],";
  dump("grown", a); dump("scaled", b, true); std::cout << "}" << std::endl;
  return 0;
}
