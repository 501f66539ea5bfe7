// test_scrt.cpp -- the mirrored SingleCRT class (SingleCRT.h:41-175) against the mirrored DoubleCRT and big-integer arithmetic.
//   test_scrt m [logQ] [p]
// Checks (exit code = number of failures):
//   toPoly(SingleCRT(poly)) = poly centred modulo the product of the set's primes            (SingleCRT.cpp:239-251, 299-334)
//   DoubleCRT = SingleCRT equals DoubleCRT(poly); toSingleCRT is its inverse, also on a subset  (DoubleCRT.cpp:484-515)
//   += / -= SingleCRT, ZZX, ZZ (constant coefficient only), *= and /= ZZ, ++ / --              (SingleCRT.cpp:61-153, 279-296)
//   addPrimes / removePrimes, Op on mismatched index sets with and without matchIndexSets      (SingleCRT.cpp:68-80, 254-268)
#include <iostream>
#include "../../fhe-si_amd/host/fhesi_host.h"
using namespace fhesi;
namespace fhesi { FHEcontext* activeContext = nullptr; }

static ZZ centred(const ZZ& v, const ZZ& P) { ZZ r = v % P; if (r < ZZ(0L)) r += P; if (r > P / ZZ(2L)) r -= P; return r; }
static ZZX centred(const ZZX& a, const ZZ& P, long n) { ZZX r; r.rep.resize(n); for (long i = 0; i < n; ++i) r.rep[i] = centred(coeff(a, i), P); r.normalize(); return r; }
static int fails = 0;
static void expect(bool ok, const char* what) { if (!ok) { ++fails; std::cout << "FAILED: " << what << std::endl; } }

int main(int argc, char** argv) {
  const unsigned m = argc > 1 ? atoi(argv[1]) : 64, logQ = argc > 2 ? atoi(argv[2]) : 100, p = argc > 3 ? atoi(argv[3]) : 23;
  FHEcontext context(m, logQ, p, 7, 3);
  activeContext = &context;
  context.SetUpSIContext();
  const long n = context.zMstar.phiM(), L = context.numPrimes();
  SetSeed(11);
  const ZZ P = context.productOfPrimes();
  ZZX a, b; a.rep.resize(n); b.rep.resize(n);
  for (long i = 0; i < n; ++i) { a.rep[i] = RandomBnd(P) - P / ZZ(2L); b.rep[i] = RandomBnd(P) - P / ZZ(2L); }
  a.rep[0] = (P - ZZ(1L)) / ZZ(2L); a.rep[1] = ZZ(0L) - (P - ZZ(1L)) / ZZ(2L);
  a.normalize(); b.normalize();

  SingleCRT sa(a, context), sb(b, context);
  ZZX t;
  sa.toPoly(t); expect(t == centred(a, P, n), "toPoly(SingleCRT(a)) == a");
  IndexSet sub(0, 1); ZZ P01 = context.productOfPrimes(sub);
  sa.toPoly(t, sub); expect(t == centred(a, P01, n), "toPoly over a subset");
  // conversions
  DoubleCRT da(context); da = sa;
  expect(da == DoubleCRT(a, context), "DoubleCRT = SingleCRT equals DoubleCRT(poly)");
  SingleCRT back(context); da.toSingleCRT(back); expect(back == sa, "toSingleCRT inverts");
  SingleCRT part(context); da.toSingleCRT(part, sub); expect(part.getIndexSet() == sub, "toSingleCRT(subset): index set");
  part.toPoly(t); expect(t == centred(a, P01, n), "toSingleCRT(subset): value");
  SingleCRT viaAssign(context); viaAssign = da; expect(viaAssign == sa, "SingleCRT = DoubleCRT");
  // arithmetic
  { SingleCRT c = sa; c += sb; c.toPoly(t); ZZX w = a; w += b; expect(t == centred(w, P, n), "+= SingleCRT"); }
  { SingleCRT c = sa; c -= sb; c.toPoly(t); ZZX w = a; w -= b; expect(t == centred(w, P, n), "-= SingleCRT"); }
  { SingleCRT c = sa; c += b; c.toPoly(t); ZZX w = a; w += b; expect(t == centred(w, P, n), "+= ZZX"); }
  { SingleCRT c = sa; c -= b; c.toPoly(t); ZZX w = a; w -= b; expect(t == centred(w, P, n), "-= ZZX"); }
  const ZZ k = ZZ(123456789L) * ZZ(987654321L) * ZZ(1000003L);
  { SingleCRT c = sa; c += k; c.toPoly(t); ZZX w = a; w.rep.resize(n); w.rep[0] += k; expect(t == centred(w, P, n), "+= ZZ touches the constant coefficient only"); }
  { SingleCRT c = sa; c -= k; c.toPoly(t); ZZX w = a; w.rep.resize(n); w.rep[0] -= k; expect(t == centred(w, P, n), "-= ZZ"); }
  { SingleCRT c = sa; ++c; c--; expect(c == sa, "++ then --"); }
  { SingleCRT c = sa; c *= k; c.toPoly(t); ZZX w = a; for (auto& x : w.rep) x *= k; expect(t == centred(w, P, n), "*= ZZ"); SingleCRT d = c; d /= k; expect(d == sa, "/= ZZ inverts *= ZZ"); }
  { SingleCRT c(context); c = 5L; c.toPoly(t); expect(deg(t) == 0 && coeff(t, 0) == ZZ(5L), "= long"); c.setZero(); c.toPoly(t); expect(deg(t) == -1, "setZero"); }
  // index-set handling
  if (L >= 3) {
    IndexSet lo(0, 1), hi(2, L - 1);
    SingleCRT c(a, context, lo);
    c.toPoly(t); ZZX alo = t;                       // a centred modulo q0 q1
    c.addPrimes(hi); expect(c.getIndexSet() == IndexSet(0, L - 1), "addPrimes: index set");
    c.toPoly(t); expect(t == alo, "addPrimes keeps the (small) polynomial");
    c.removePrimes(hi); expect(c == SingleCRT(a, context, lo), "removePrimes");
    SingleCRT d(a, context, lo); d.Add(sb, false); expect(d.getIndexSet() == lo, "Add(matchIndexSets=false) keeps the index set");
    d.toPoly(t); ZZX w = a; w += b; expect(t == centred(w, context.productOfPrimes(lo), n), "Add(matchIndexSets=false): value");
    SingleCRT e(a, context, lo); e += sb; expect(e.getIndexSet() == IndexSet(0, L - 1), "+= grows to the union");
    e.toPoly(t); ZZX w2 = alo; w2 += b; expect(t == centred(w2, P, n), "+= on the union: value");
  }
  std::cout << (fails ? "Test FAILED" : "Test SUCCEEDED") << " (m=" << m << " phi(m)=" << n << " primes=" << L << ")" << std::endl;
  return fails;
}
