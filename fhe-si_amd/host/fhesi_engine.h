// fhesi_engine.h -- device-resident, lazily evaluated ciphertext values behind the mirror's Ciphertext (included by fhesi_host.h
// after FHEcontext; not a stand-alone header).
//
// Why: code written against the reference's class surface works on ONE Ciphertext object at a time -- `tmp = A(i, k); tmp *= B(k, j);
// sum += tmp; ... keySwitch.ApplyKeySwitch(sum)` (Matrix.cpp:57-98,150-263, Regression.h:102-149,166-178).  Run literally, every statement
// is a round trip through host big integers (4 MiB per ciphertext at the metric ring) and a device call on a batch of one.  Here an
// unscaled two-part ciphertext is a VALUE (CtValue): a slot of an arena in HBM, or a recorded operation on other values that has not run
// yet.  Ciphertext's operators record; nothing runs until a value is looked at (decryption, parts[i].poly, export, ...), and then every
// recorded operation is evaluated level by level, all operations of a level and kind as ONE batched C-ABI call:
//     KS_SUM   out = ApplyKeySwitch(sum_t a_t *= b_t)        -> fhesi_ct_mul_sum_relin_dev  (Ciphertext.cpp:167-192 + :135-142 + FHE-SI.cpp:241-260)
//     AUTO_KS  out = ApplyKeySwitch_k(a >>= k)               -> fhesi_ct_automorph_key_switch_dev (Regression.h:170-172)
//     ADD / SCALE / AUTO  unscaled +=, *= long, >>= k        -> fhesi_ct_add_dev / fhesi_ct_mul_long_dev / fhesi_ct_automorph_dev
// Values are immutable, so copies of a Ciphertext share them, and an operation recorded twice on the same inputs is recorded once
// (the Laplace expansion of Matrix.cpp:227-263 recomputes equal minors: d = 8 writes 5.5 * 10^5 partial determinants, 12870 are distinct).  The results are the bits the object-at-a-time bodies give (every batched
// call is checked against them: tests/host/test_lazy.cpp, test_regression.cpp); FHESI_EAGER=1 (or LazyCiphertexts() = false) turns the
// recording off and runs every statement at once, as before.  Errors of a recorded operation surface when it runs, not when it is
// recorded.  Single-threaded, like the reference's classes.
#pragma once

namespace fhesi {

inline bool& LazyCiphertexts() { static bool on = std::getenv("FHESI_EAGER") == nullptr; return on; }

// a key-switching matrix as ONE object in HBM, shared by the KeySwitchSI it mirrors and by the recorded operations that will use it
struct DeviceKey {
  fhesi_ksk* k = nullptr;
  const long id;                                            // never reused (an address can be)
  static long next_id() { static long n = 0; return ++n; }
  explicit DeviceKey(fhesi_ksk* kk) : k(kk), id(next_id()) {}
  ~DeviceKey() { if (k) fhesi_ksk_free(k); }
  DeviceKey(const DeviceKey&) = delete;
};
typedef std::shared_ptr<DeviceKey> DeviceKeyRef;

class CtEngine;
struct CtValue;
typedef std::shared_ptr<CtValue> CtRef;
typedef std::vector<std::pair<CtRef, CtRef>> CtTerms;      // a scaled-up ciphertext that has not been multiplied out: sum of a_t x b_t

struct CtValue {
  enum Kind : uint8_t { DEVICE, KS_SUM, AUTO_KS, ADD, SCALE, AUTO };
  std::shared_ptr<CtEngine> eng;
  Kind kind = DEVICE;
  long id = 0;                    // unique per engine; never reused
  long slot = -1;                 // arena index once the value exists in HBM
  int depth = -1;                 // scratch of CtEngine::flush
  CtTerms terms;                  // KS_SUM
  CtRef a, b;                     // ADD(a, b); SCALE(a, s); AUTO(a, k = s); AUTO_KS(a, k = s)
  long s = 0;
  DeviceKeyRef key;               // KS_SUM, AUTO_KS
  bool pending() const { return slot < 0; }
  ~CtValue();
};

class CtEngine : public std::enable_shared_from_this<CtEngine> {
  const FHEcontext& context;
  uint64_t* base = nullptr;
  long cap = 0;
  std::map<long, long> freeRuns;                       // start -> length, coalesced
  std::vector<std::weak_ptr<CtValue>> recorded;        // every value that was pending when created
  bool dead = false;
  typedef std::vector<long> Sig;                       // (kind, scalar, key, ids of the inputs): equal signatures = equal values
  std::map<Sig, std::weak_ptr<CtValue>> memo;
  long nextId = 0;

  void add_free(long start, long len) {
    auto it = freeRuns.lower_bound(start);
    if (it != freeRuns.begin()) { auto pv = std::prev(it); if (pv->first + pv->second == start) { start = pv->first; len += pv->second; freeRuns.erase(pv); } }
    if (it != freeRuns.end() && start + len == it->first) { len += it->second; freeRuns.erase(it); }
    freeRuns[start] = len;
  }
  void grow(long count) {
    long tail = 0;                                      // a free run that ends at the top is extended instead of left behind
    if (!freeRuns.empty()) { auto last = std::prev(freeRuns.end()); if (last->first + last->second == cap) tail = last->second; }
    const long ncap = std::max<long>(std::max(cap * 2, cap + count - tail), 32);
    void* nb; ck(fhesi_dev_alloc(h, (size_t)ncap * words * 8, &nb));
    if (base) { ck(fhesi_dev_copy(h, nb, base, (size_t)cap * words * 8)); ck(fhesi_ctx_sync(h)); ck(fhesi_dev_free(h, base)); }
    base = (uint64_t*)nb;
    add_free(cap, ncap - cap);
    cap = ncap;
  }
  CtRef make(CtValue::Kind kind) {
    CtRef v = std::make_shared<CtValue>(); v->eng = shared_from_this(); v->kind = kind; v->id = ++nextId;
    if (kind != CtValue::DEVICE) { recorded.push_back(v); ++stats.recorded; }
    return v;
  }
  CtRef known(const Sig& sig) {
    if (!shareEqual) return nullptr;
    auto it = memo.find(sig);
    if (it == memo.end()) return nullptr;
    if (CtRef v = it->second.lock()) { ++stats.shared; return v; }
    memo.erase(it);
    return nullptr;
  }
  CtRef remember(const Sig& sig, CtRef v) { if (shareEqual) memo[sig] = v; maybe_flush(); return v; }
  static Sig sig_of(CtValue::Kind kind, long s, const DeviceKey* key, const CtValue* a, const CtValue* b) { return Sig{(long)kind, s, key ? key->id : 0, a ? a->id : 0, b ? b->id : 0}; }
  static void done(CtValue* v, long slot) { v->slot = slot; v->terms.clear(); v->a.reset(); v->b.reset(); v->key.reset(); }
  std::vector<int32_t> slots_of(const std::vector<CtValue*>& vs, bool second) const { std::vector<int32_t> r; for (auto v : vs) r.push_back((int32_t)(second ? v->b->slot : v->a->slot)); return r; }

 public:
  fhesi_ctx* const h;
  const long n;
  const int nl;
  const long words;                                    // uint64 per ciphertext: [2][phi(m)][nl]
  struct Stats { long recorded = 0, shared = 0, flushes = 0, calls = 0, products = 0, key_switches = 0; } stats;
  bool shareEqual = true;                              // an operation recorded again on the same values returns the value recorded first
  long flushAt = 8192;                                 // recorded operations that trigger an evaluation by themselves (bounds the graph held on the host)

  explicit CtEngine(const FHEcontext& c) : context(c), h(c.handle()), n(c.zMstar.phiM()), nl((int)((c.logQ + 63) / 64)), words(2 * (long)c.zMstar.phiM() * (long)((c.logQ + 63) / 64)) {}
  ~CtEngine() { shutdown(); }
  void shutdown() { dead = true; recorded.clear(); memo.clear(); if (base) { fhesi_dev_free(h, base); base = nullptr; } cap = 0; freeRuns.clear(); }
  const FHEcontext& ctx() const { return context; }
  uint64_t* ptr(long slot) const { return base + slot * words; }
  const uint64_t* pool() const { return base; }
  long alloc_run(long count) {                         // `count` consecutive slots (first fit); pointers taken earlier are invalid afterwards
    if (dead) Error("CtEngine: the context of this ciphertext is gone");
    for (;;) {
      for (auto it = freeRuns.begin(); it != freeRuns.end(); ++it)
        if (it->second >= count) { const long s = it->first, len = it->second; freeRuns.erase(it); if (len > count) freeRuns[s + count] = len - count; return s; }
      grow(count);
    }
  }
  void free_run(long start, long count) { if (!dead && count > 0) add_free(start, count); }
  void release(long slot) { free_run(slot, 1); }

  // ---- values that exist
  CtRef wrap(long slot) { CtRef v = make(CtValue::DEVICE); v->slot = slot; return v; }                 // takes ownership of a filled slot
  CtRef upload(const uint64_t* host) { const long s = alloc_run(1); ck(fhesi_dev_upload(h, ptr(s), host, (size_t)words * 8)); return wrap(s); }
  void download(const CtRef& v, uint64_t* host) { force(v); ck(fhesi_dev_download(h, host, ptr(v->slot), (size_t)words * 8)); }
  long clone_slot(const CtRef& v) { force(v); const long s = alloc_run(1); ck(fhesi_dev_copy(h, ptr(s), ptr(v->slot), (size_t)words * 8)); return s; }
  // ---- recorded operations
  CtRef ks_sum(CtTerms terms, DeviceKeyRef key) {
    Sig sig{(long)CtValue::KS_SUM, 0, key->id};
    for (auto& t : terms) { sig.push_back(t.first->id); sig.push_back(t.second->id); }
    if (CtRef v = known(sig)) return v;
    CtRef v = make(CtValue::KS_SUM); v->terms = std::move(terms); v->key = std::move(key); return remember(sig, v);
  }
  CtRef auto_ks(CtRef a, long k, DeviceKeyRef key) {
    const Sig sig = sig_of(CtValue::AUTO_KS, k, key.get(), a.get(), nullptr);
    if (CtRef v = known(sig)) return v;
    CtRef v = make(CtValue::AUTO_KS); v->a = std::move(a); v->s = k; v->key = std::move(key); return remember(sig, v);
  }
  CtRef add(CtRef a, CtRef b) {
    const Sig sig = sig_of(CtValue::ADD, 0, nullptr, a.get(), b.get());
    if (CtRef v = known(sig)) return v;
    CtRef v = make(CtValue::ADD); v->a = std::move(a); v->b = std::move(b); return remember(sig, v);
  }
  CtRef scale(CtRef a, long l) {
    const Sig sig = sig_of(CtValue::SCALE, l, nullptr, a.get(), nullptr);
    if (CtRef v = known(sig)) return v;
    CtRef v = make(CtValue::SCALE); v->a = std::move(a); v->s = l; return remember(sig, v);
  }
  CtRef automorph(CtRef a, long k) {
    const Sig sig = sig_of(CtValue::AUTO, k, nullptr, a.get(), nullptr);
    if (CtRef v = known(sig)) return v;
    CtRef v = make(CtValue::AUTO); v->a = std::move(a); v->s = k; return remember(sig, v);
  }
  void maybe_flush() { if ((long)recorded.size() >= flushAt) flush(); }
  void force(const CtRef& v) { if (v->pending()) flush(); if (v->pending()) Error("CtEngine: a recorded operation was not evaluated"); }

  // ---- evaluation of everything recorded, level by level
  void flush() {
    std::vector<CtRef> todo;
    for (auto& w : recorded) if (CtRef v = w.lock()) if (v->pending()) todo.push_back(v);
    recorded.clear();
    for (auto it = memo.begin(); it != memo.end();) { if (it->second.expired()) it = memo.erase(it); else ++it; }
    if (todo.empty()) return;
    ++stats.flushes;
    for (auto& v : todo) v->depth = -1;
    // depth = 1 + the deepest pending input; an explicit stack, graphs of long chains must not overflow the call stack
    auto inputs = [](CtValue* v, std::vector<CtValue*>& out) { out.clear(); for (auto& t : v->terms) { out.push_back(t.first.get()); out.push_back(t.second.get()); } if (v->a) out.push_back(v->a.get()); if (v->b) out.push_back(v->b.get()); };
    std::vector<CtValue*> stack, in;
    int maxd = 0;
    for (auto& root : todo) {
      if (root->depth >= 0) continue;
      stack.push_back(root.get());
      while (!stack.empty()) {
        CtValue* v = stack.back();
        if (v->depth >= 0) { stack.pop_back(); continue; }
        inputs(v, in);
        int d = 0; bool ready = true;
        for (CtValue* i : in) { if (!i->pending()) continue; if (i->depth < 0) { stack.push_back(i); ready = false; } else d = std::max(d, i->depth); }
        if (!ready) continue;
        v->depth = d + 1; maxd = std::max(maxd, v->depth); stack.pop_back();
      }
    }
    std::vector<std::vector<CtValue*>> level(maxd + 1);
    for (auto& v : todo) level[v->depth].push_back(v.get());
    for (int d = 1; d <= maxd; ++d) run_level(level[d]);
  }

 private:
  void run_level(const std::vector<CtValue*>& vs) {
    const int32_t logQ = (int32_t)context.logQ, decomp = (int32_t)context.decompSize; const uint64_t p = (uint64_t)context.ModulusP().to_long();
    std::map<DeviceKey*, std::vector<CtValue*>> sums;
    std::map<std::pair<DeviceKey*, long>, std::vector<CtValue*>> autoKs;
    std::map<long, std::vector<CtValue*>> scales, autos;
    std::vector<CtValue*> adds;
    for (CtValue* v : vs) switch (v->kind) {
      case CtValue::KS_SUM: sums[v->key.get()].push_back(v); break;
      case CtValue::AUTO_KS: autoKs[std::make_pair(v->key.get(), v->s)].push_back(v); break;
      case CtValue::ADD: adds.push_back(v); break;
      case CtValue::SCALE: scales[v->s].push_back(v); break;
      case CtValue::AUTO: autos[v->s].push_back(v); break;
      default: Error("CtEngine: a device value among the recorded operations");
    }
    for (auto& kv : sums) {                            // one wave: out[g] = KeySwitch(sum of the group's products)
      const auto& g = kv.second; const long G = (long)g.size();
      std::vector<int32_t> a, b, seg{0};
      for (CtValue* v : g) { for (auto& t : v->terms) { a.push_back((int32_t)t.first->slot); b.push_back((int32_t)t.second->slot); } seg.push_back((int32_t)a.size()); }
      const long first = alloc_run(G);
      ck(fhesi_ct_mul_sum_relin_dev(h, kv.first->k, logQ, p, decomp, base, nl, a.data(), b.data(), seg.data(), G, ptr(first)));
      ++stats.calls; stats.products += (long)a.size(); stats.key_switches += G;
      for (long i = 0; i < G; ++i) done(g[i], first + i);
    }
    for (auto& kv : autoKs) {                          // (ctxt >>= k; ApplyKeySwitch) on the gathered inputs
      const auto& g = kv.second; const long G = (long)g.size();
      const std::vector<int32_t> idx = slots_of(g, false);
      const long tmp = alloc_run(G), out = alloc_run(G);
      ck(fhesi_ct_gather_dev(h, base, idx.data(), G, words, ptr(tmp)));
      ck(fhesi_ct_automorph_key_switch_dev(h, kv.first.first->k, logQ, decomp, (int64_t)kv.first.second, ptr(tmp), nl, G, ptr(out), nl));
      free_run(tmp, G);                                // (stream order: whoever reuses it is queued behind the call that reads it)
      ++stats.calls; stats.key_switches += G;
      for (long i = 0; i < G; ++i) done(g[i], out + i);
    }
    if (!adds.empty()) {
      const long G = (long)adds.size();
      const std::vector<int32_t> ia = slots_of(adds, false), ib = slots_of(adds, true);
      const long tmp = alloc_run(G), out = alloc_run(G);
      ck(fhesi_ct_gather_dev(h, base, ia.data(), G, words, ptr(out)));
      ck(fhesi_ct_gather_dev(h, base, ib.data(), G, words, ptr(tmp)));
      ck(fhesi_ct_add_dev(h, logQ, ptr(out), ptr(tmp), 2, nl, G));
      free_run(tmp, G);
      ++stats.calls;
      for (long i = 0; i < G; ++i) done(adds[i], out + i);
    }
    for (auto& kv : scales) {
      const auto& g = kv.second; const long G = (long)g.size();
      const std::vector<int32_t> idx = slots_of(g, false);
      const long out = alloc_run(G);
      ck(fhesi_ct_gather_dev(h, base, idx.data(), G, words, ptr(out)));
      ck(fhesi_ct_mul_long_dev(h, logQ, ptr(out), (int64_t)kv.first, 2, nl, G));
      ++stats.calls;
      for (long i = 0; i < G; ++i) done(g[i], out + i);
    }
    for (auto& kv : autos) {                           // an automorphism looked at before its key switch: coefficients as CiphertextPart::operator>>= leaves them, modulo 2^(64 nl)
      const auto& g = kv.second; const long G = (long)g.size();
      const std::vector<int32_t> idx = slots_of(g, false);
      const long tmp = alloc_run(G), out = alloc_run(G);
      ck(fhesi_ct_gather_dev(h, base, idx.data(), G, words, ptr(tmp)));
      ck(fhesi_ct_automorph_dev(h, (int64_t)kv.first, ptr(tmp), 2, nl, G, ptr(out), nl));
      free_run(tmp, G);
      ++stats.calls;
      for (long i = 0; i < G; ++i) done(g[i], out + i);
    }
  }
};

inline CtValue::~CtValue() { if (slot >= 0 && eng) eng->release(slot); }

// one engine per context, created on first use; the context's destructor shuts it down before the device context goes
inline std::map<const FHEcontext*, std::shared_ptr<CtEngine>>& ct_engines() { static std::map<const FHEcontext*, std::shared_ptr<CtEngine>> m; return m; }
inline CtEngine& ct_engine(const FHEcontext& c) {
  auto& m = ct_engines();
  auto it = m.find(&c);
  if (it == m.end()) it = m.emplace(&c, std::make_shared<CtEngine>(c)).first;
  return *it->second;
}
inline void drop_ct_engine(const FHEcontext* c) { auto& m = ct_engines(); auto it = m.find(c); if (it != m.end()) { it->second->shutdown(); m.erase(it); } }
// evaluate everything recorded for this context and wait for the device (timing harnesses; results need no explicit call)
inline void SyncCiphertexts(const FHEcontext& c) { ct_engine(c).flush(); ck(fhesi_ctx_sync(c.handle())); }

}  // namespace fhesi
