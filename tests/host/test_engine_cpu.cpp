// test_engine_cpu.cpp -- TEST HARNESS (no GPU): the host logic of the recording evaluator (fhe-si_amd/host/fhesi_engine.h) against a direct
// evaluation of the same random operation graphs, on the toy arithmetic of mock_abi.cpp.  Built with -fsanitize=address,undefined by
// tests/test_engine_host_logic.py.  What it covers: runs of the arena handed out, freed and reused while values die in any order; growth
// of the arena under live values; sharing of equal operations (and NOT sharing across different keys, scalars or operand orders);
// levelling of graphs whose depth and width vary; evaluation triggered by reads, by the threshold and by destruction order; key lifetime.
#include <iostream>
#include <random>

#include "../../fhe-si_amd/host/fhesi_host.h"

using namespace fhesi;
namespace fhesi { FHEcontext* activeContext = nullptr; }

typedef std::vector<uint64_t> Val;
struct MockKey { uint64_t tag; };                            // mirrors mock_abi.cpp's fhesi_ksk
static uint64_t tag_of(const DeviceKeyRef& k) { return reinterpret_cast<const MockKey*>(k->k)->tag; }

int main(int argc, char* argv[]) {
  const long steps = argc > 1 ? atol(argv[1]) : 3000; const unsigned seed = argc > 2 ? (unsigned)atol(argv[2]) : 1;
  FHEcontext context(16, 128, 23, 3);                        // phi = 8, nl = 2: 32 words per toy ciphertext
  activeContext = &context;
  context.AddPrime(97, false, 19);                           // any chain; the mock ignores it (97 = 1 mod 32, root unused)
  CtEngine& eng = ct_engine(context);
  const size_t W = (size_t)eng.words; const uint64_t P = 23;
  std::mt19937_64 rng(seed);
  auto make_key = [&](int nc) { fhesi_ksk* k = nullptr; ck(fhesi_ksk_create(context.handle(), nc, 6, &k)); return std::make_shared<DeviceKey>(k, nc, 6); };
  std::vector<DeviceKeyRef> keys3{make_key(3), make_key(3)}, keys2{make_key(2), make_key(2)};
  struct Entry { CtRef v; Val want; };
  std::vector<Entry> pool;
  auto fresh = [&]() { Val x(W); for (auto& w : x) w = rng(); Entry e{eng.upload(x.data()), x}; return e; };
  for (int i = 0; i < 6; ++i) pool.push_back(fresh());
  long checked = 0, bad = 0;
  {   // an operation recorded again on the same values is the value recorded first; a different key, scalar or operand order is not
    const CtRef x = pool[0].v, y = pool[1].v;
    CtTerms t1{std::make_pair(x, y)}, t2{std::make_pair(y, x)};
    const CtRef a = eng.ks_sum(t1, keys3[0]), b = eng.ks_sum(t1, keys3[0]), c = eng.ks_sum(t1, keys3[1]), d = eng.ks_sum(t2, keys3[0]);
    const CtRef e = eng.scale(x, 2), f = eng.scale(x, 2), g = eng.scale(x, 3), h = eng.add(x, y), i = eng.add(y, x), j = eng.add(x, y);
    const bool ok = a == b && a != c && a != d && e == f && e != g && h == j && h != i;
    ++checked; if (!ok) { ++bad; std::cout << "  sharing of equal operations is wrong" << std::endl; }
    eng.shareEqual = false;
    const CtRef k = eng.scale(x, 2);
    ++checked; if (k == e) { ++bad; std::cout << "  shareEqual = false still shares" << std::endl; }
    eng.shareEqual = true;
  }
  auto check = [&](Entry& e) { Val got(W); eng.download(e.v, got.data()); ++checked; if (got != e.want) { ++bad; std::cout << "  mismatch (value id " << e.v->id << ", kind " << (int)e.v->kind << ")" << std::endl; } };
  for (long s = 0; s < steps; ++s) {
    const int op = (int)(rng() % 14);
    auto pick = [&]() -> Entry& { return pool[rng() % pool.size()]; };
    switch (op) {
      case 0: case 1: case 2: {                              // key switch of a sum of 1..5 products
        const int nt = 1 + (int)(rng() % 5); const DeviceKeyRef& key = keys3[rng() % 2];
        CtTerms terms; Val acc(W, 0);
        for (int t = 0; t < nt; ++t) { Entry &a = pick(), &b = pick(); terms.push_back(std::make_pair(a.v, b.v)); for (size_t w = 0; w < W; ++w) acc[w] += (a.want[w] * P + 1) * (b.want[w] ^ (uint64_t)w); }
        for (auto& w : acc) w = w * 3 + tag_of(key);
        pool.push_back(Entry{eng.ks_sum(terms, key), acc}); break;
      }
      case 3: { Entry &a = pick(), &b = pick(); Val r = a.want; for (size_t w = 0; w < W; ++w) r[w] += b.want[w] * 2; pool.push_back(Entry{eng.add(a.v, b.v), r}); break; }
      case 4: { Entry& a = pick(); const long l = (long)(rng() % 7) - 3; Val r = a.want; for (auto& w : r) w = w * (uint64_t)l + 11; pool.push_back(Entry{eng.scale(a.v, l), r}); break; }
      case 5: { Entry& a = pick(); const long k = 1 + (long)(rng() % 5); const DeviceKeyRef& key = keys2[rng() % 2]; Val r(W); for (size_t w = 0; w < W; ++w) r[w] = a.want[(w + (size_t)k) % W] * 5 + tag_of(key) + (uint64_t)k;
                pool.push_back(Entry{eng.auto_ks(a.v, k, key), r}); break; }
      case 6: { Entry& a = pick(); const long k = 1 + (long)(rng() % 5); Val r(W); for (size_t w = 0; w < W; ++w) r[w] = a.want[(w + (size_t)k) % W] + 7; pool.push_back(Entry{eng.automorph(a.v, k), r}); break; }
      case 7: case 8: if (pool.size() > 4) { pool.erase(pool.begin() + (long)(rng() % pool.size())); } break;          // values die in any order (slots return to the arena)
      case 9: check(pick()); break;                                                                                    // a read evaluates everything recorded
      case 10: eng.flushAt = (rng() % 3 == 0) ? 4 : 8192; break;
      case 11: pool.push_back(fresh()); break;
      case 12: { Entry& a = pick(); const long sl = eng.clone_slot(a.v); pool.push_back(Entry{eng.wrap(sl), a.want}); break; }     // a copy made in place (Ciphertext += constant takes this route)
      case 13: if (rng() % 8 == 0) { keys3[rng() % 2] = make_key(3); } break;                                          // a key replaced while operations that use it are recorded
    }
    if (pool.size() > 40) pool.erase(pool.begin(), pool.begin() + 20);
  }
  for (auto& e : pool) check(e);
  std::cout << "engine host logic: " << steps << " steps, " << checked << " values compared, " << bad << " mismatches; " << eng.stats.recorded << " recorded, " << eng.stats.shared << " shared, "
            << eng.stats.flushes << " evaluations, " << eng.stats.calls << " calls" << std::endl;
  pool.clear();
  std::cout << (bad ? "Test FAILED" : "Test SUCCEEDED") << std::endl;
  return bad ? 1 : 0;
}
