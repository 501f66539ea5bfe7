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
// (the Laplace expansion of Matrix.cpp:227-263 recomputes equal minors: d = 8 writes 5.5 * 10^5 partial determinants, 12870 are distinct).
// The results are the bits the object-at-a-time bodies give (every batched call is checked against them: tests/host/test_lazy.cpp,
// test_regression.cpp, test_statistics.cpp); FHESI_EAGER=1 (or LazyCiphertexts() = false) turns the recording off and runs every statement
// at once.  Errors of a recorded operation surface when it runs, not when it is recorded.  One host thread uses the classes of a context,
// as in the reference.
//
// Several GPUs: EnableCiphertextGroup(context, devices) gives every GPU of the list a replica of the arena (same slots,
// same contents) and of every key-switching matrix a recorded operation uses (one RCCL broadcast each, FHE-SI.cpp:206-208).  The groups of a
// key-switch level are then sharded over the GPUs (shard_bounds) and the level's outputs exchanged (fhesi_comm_exchange), one host thread per
// GPU; the cheap unscaled operations run on every GPU.  Every ciphertext operation is deterministic, so the results are the bits one GPU gives.
#pragma once
#include <thread>

namespace fhesi {

inline bool& LazyCiphertexts() { static bool on = std::getenv("FHESI_EAGER") == nullptr; return on; }

// a key-switching matrix as ONE object in HBM, shared by the KeySwitchSI it mirrors and by the recorded operations that will use it
struct DeviceKey {
  fhesi_ksk* k = nullptr;
  const long id;                                            // never reused (an address can be)
  int ncomp = 0, ndigits = 0;                               // the matrix's shape (source-key components, digits per component)
  std::vector<fhesi_ksk*> replicas;                         // [rank] on the GPUs of an engine group, rank 0 = k (CtEngine::key_on_ranks); freed by the engine at shutdown
  long groupGen = 0;
  static long next_id() { static long n = 0; return ++n; }
  DeviceKey(fhesi_ksk* kk, int nc, int nd) : k(kk), id(next_id()), ncomp(nc), ndigits(nd) {}
  void drop_replicas() { for (size_t r = 1; r < replicas.size(); ++r) if (replicas[r]) fhesi_ksk_free(replicas[r]); replicas.clear(); }
  ~DeviceKey() { drop_replicas(); if (k) fhesi_ksk_free(k); }
  DeviceKey(const DeviceKey&) = delete;
};
typedef std::shared_ptr<DeviceKey> DeviceKeyRef;

// contiguous, balanced shard [lo, hi) of `total` units for `rank` (the first total % world ranks get one more) -- the rule of
// fhe-si_amd/shard.py::shard_bounds, so the C++ and Python hosts split a wave identically
inline void shard_bounds(long total, int rank, int world, long& lo, long& hi) {
  const long base = total / world, extra = total % world;
  lo = rank * base + std::min<long>(rank, extra);
  hi = lo + base + (rank < extra ? 1 : 0);
}

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
  // the other GPUs of a group (rank r = peers[r - 1]; rank 0 is the context's own GPU): a device context, a replica of the arena, a communicator
  struct Peer { fhesi_ctx* h = nullptr; uint64_t* base = nullptr; fhesi_comm* comm = nullptr; };
  std::vector<Peer> peers;
  fhesi_comm* comm0 = nullptr;
  long groupGen = 0;
  std::vector<std::weak_ptr<DeviceKey>> replicated;    // keys with replicas on the peers (freed before the peers' contexts go)
  int world() const { return 1 + (int)peers.size(); }
  fhesi_ctx* h_of(int r) const { return r ? peers[r - 1].h : h; }
  uint64_t* base_of(int r) const { return r ? peers[r - 1].base : base; }
  fhesi_comm* comm_of(int r) const { return r ? peers[r - 1].comm : comm0; }
  template <class F> void on_all(F f) {                // f(rank) on one host thread per GPU
    std::vector<std::thread> th;
    for (int r = 1; r < world(); ++r) th.emplace_back([&f, r] { f(r); });
    f(0);
    for (auto& t : th) t.join();
  }
  // the key's matrix on every GPU of the group: replicas created and filled by ONE broadcast from rank 0 on first use
  void key_on_ranks(const DeviceKeyRef& key) {
    if (peers.empty() || (key->groupGen == groupGen && (int)key->replicas.size() == world())) return;
    key->drop_replicas();
    key->replicas.assign(world(), nullptr); key->replicas[0] = key->k;
    for (int r = 1; r < world(); ++r) ck(fhesi_ksk_create(h_of(r), key->ncomp, key->ndigits, &key->replicas[r]));
    on_all([&](int r) { ck(fhesi_ksk_broadcast(key->replicas[r], comm_of(r), 0)); });
    key->groupGen = groupGen;
    replicated.push_back(key);
  }
  fhesi_ksk* key_of(const DeviceKey* key, int r) const { return r ? key->replicas[r] : key->k; }
  // slots [first, first + count) produced in shards by the ranks -> everywhere
  void exchange(long first, long count) {
    if (peers.empty()) return;
    const int R = world();
    std::vector<int64_t> off(R + 1);
    for (int r = 0; r < R; ++r) { long lo, hi; shard_bounds(count, r, R, lo, hi); off[r] = (first + lo) * words; off[r + 1] = (first + hi) * words; }
    on_all([&](int r) { ck(fhesi_comm_exchange(h_of(r), comm_of(r), base_of(r), off.data())); });
  }

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
    on_all([&](int r) {
      uint64_t*& b = r ? peers[r - 1].base : base;
      void* nb; ck(fhesi_dev_alloc(h_of(r), (size_t)ncap * words * 8, &nb));
      if (b) { ck(fhesi_dev_copy(h_of(r), nb, b, (size_t)cap * words * 8)); ck(fhesi_ctx_sync(h_of(r))); ck(fhesi_dev_free(h_of(r), b)); }
      b = (uint64_t*)nb;
    });
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
  void shutdown() { dead = true; recorded.clear(); memo.clear(); drop_group(); if (base) { fhesi_dev_free(h, base); base = nullptr; } cap = 0; freeRuns.clear(); }
  // ---- several GPUs
  // devices[0] must be the context's own GPU; a repeated device makes a loopback group (fhesi_comm_init_all), which is how one GPU tests this
  void enable_group(const std::vector<int>& devices) {
    if (dead) Error("CtEngine: the context is gone");
    flush();
    drop_group();
    const int G = (int)devices.size();
    if (G < 2) return;
    if (devices[0] != context.deviceIndex()) Error("EnableCiphertextGroup: devices[0] must be the context's GPU");
    std::vector<int32_t> devs(devices.begin(), devices.end());
    std::vector<fhesi_comm*> comms(G, nullptr);
    ck(fhesi_comm_init_all(G, devs.data(), comms.data()));
    comm0 = comms[0];
    peers.resize(G - 1);
    for (int r = 1; r < G; ++r) { peers[r - 1].h = context.replica(devices[r]); peers[r - 1].comm = comms[r]; }
    ++groupGen;
    if (cap) {                                         // the arena as it is now, on every GPU
      for (int r = 1; r < G; ++r) { void* nb; ck(fhesi_dev_alloc(h_of(r), (size_t)cap * words * 8, &nb)); peers[r - 1].base = (uint64_t*)nb; }
      publish(0, cap);
    }
  }
  void drop_group() {
    if (peers.empty() && !comm0) return;
    for (auto& w : replicated) if (auto key = w.lock()) key->drop_replicas();
    replicated.clear();
    for (auto& pe : peers) { if (pe.base) fhesi_dev_free(pe.h, pe.base); if (pe.comm) fhesi_comm_destroy(pe.comm); }
    if (comm0) { fhesi_comm_destroy(comm0); comm0 = nullptr; }
    for (auto& pe : peers) if (pe.h) fhesi_ctx_destroy(pe.h);
    peers.clear();
  }
  int group_size() const { return world(); }
  // slots written on rank 0 only (an encryption batch, a copy modified in place, ...) -> every GPU of the group
  void publish(long first, long count) {
    if (peers.empty() || count <= 0) return;
    on_all([&](int r) { ck(fhesi_comm_broadcast_dev(h_of(r), comm_of(r), base_of(r) + first * words, (size_t)count * words * 8, 0)); });
  }
  void sync_all() { for (int r = 0; r < world(); ++r) ck(fhesi_ctx_sync(h_of(r))); }
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
  CtRef upload(const uint64_t* host) { const long s = alloc_run(1); on_all([&](int r) { ck(fhesi_dev_upload(h_of(r), base_of(r) + s * words, host, (size_t)words * 8)); }); return wrap(s); }
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
    for (size_t r = 0; r < peers.size(); ++r) ck(fhesi_ctx_copy_options(peers[r].h, h));      // options changed on the primary context since the group was made apply to every rank
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
    const int R = world();
    for (auto& kv : sums) {                            // one wave: out[g] = KeySwitch(sum of the group's products); the groups sharded over the GPUs
      const auto& g = kv.second; const long G = (long)g.size();
      std::vector<int32_t> a, b, seg{0};
      for (CtValue* v : g) { for (auto& t : v->terms) { a.push_back((int32_t)t.first->slot); b.push_back((int32_t)t.second->slot); } seg.push_back((int32_t)a.size()); }
      const long first = alloc_run(G);
      key_on_ranks(g[0]->key);
      on_all([&](int r) {
        long lo, hi; shard_bounds(G, r, R, lo, hi);
        if (hi <= lo) return;
        const int32_t t0 = seg[lo];
        std::vector<int32_t> sseg(seg.begin() + lo, seg.begin() + hi + 1);
        for (auto& x : sseg) x -= t0;
        ck(fhesi_ct_mul_sum_relin_dev(h_of(r), key_of(kv.first, r), logQ, p, decomp, base_of(r), nl, a.data() + t0, b.data() + t0, sseg.data(), hi - lo, base_of(r) + (first + lo) * words));
      });
      exchange(first, G);
      ++stats.calls; stats.products += (long)a.size(); stats.key_switches += G;
      for (long i = 0; i < G; ++i) done(g[i], first + i);
    }
    for (auto& kv : autoKs) {                          // (ctxt >>= k; ApplyKeySwitch) on the gathered inputs, sharded likewise
      const auto& g = kv.second; const long G = (long)g.size();
      const std::vector<int32_t> idx = slots_of(g, false);
      const long tmp = alloc_run(G), out = alloc_run(G);
      key_on_ranks(g[0]->key);
      on_all([&](int r) {
        long lo, hi; shard_bounds(G, r, R, lo, hi);
        if (hi <= lo) return;
        uint64_t* bs = base_of(r);
        ck(fhesi_ct_gather_dev(h_of(r), bs, idx.data() + lo, hi - lo, words, bs + (tmp + lo) * words));
        ck(fhesi_ct_automorph_key_switch_dev(h_of(r), key_of(kv.first.first, r), logQ, decomp, (int64_t)kv.first.second, bs + (tmp + lo) * words, nl, hi - lo, bs + (out + lo) * words, nl));
      });
      exchange(out, G);
      free_run(tmp, G);                                // (stream order: whoever reuses it is queued behind the call that reads it)
      ++stats.calls; stats.key_switches += G;
      for (long i = 0; i < G; ++i) done(g[i], out + i);
    }
    // the unscaled operations are cheap and local: every GPU of a group computes all of them (no exchange)
    if (!adds.empty()) {
      const long G = (long)adds.size();
      const std::vector<int32_t> ia = slots_of(adds, false), ib = slots_of(adds, true);
      const long tmp = alloc_run(G), out = alloc_run(G);
      on_all([&](int r) {
        uint64_t* bs = base_of(r);
        ck(fhesi_ct_gather_dev(h_of(r), bs, ia.data(), G, words, bs + out * words));
        ck(fhesi_ct_gather_dev(h_of(r), bs, ib.data(), G, words, bs + tmp * words));
        ck(fhesi_ct_add_dev(h_of(r), logQ, bs + out * words, bs + tmp * words, 2, nl, G));
      });
      free_run(tmp, G);
      ++stats.calls;
      for (long i = 0; i < G; ++i) done(adds[i], out + i);
    }
    for (auto& kv : scales) {
      const auto& g = kv.second; const long G = (long)g.size();
      const std::vector<int32_t> idx = slots_of(g, false);
      const long out = alloc_run(G);
      on_all([&](int r) {
        uint64_t* bs = base_of(r);
        ck(fhesi_ct_gather_dev(h_of(r), bs, idx.data(), G, words, bs + out * words));
        ck(fhesi_ct_mul_long_dev(h_of(r), logQ, bs + out * words, (int64_t)kv.first, 2, nl, G));
      });
      ++stats.calls;
      for (long i = 0; i < G; ++i) done(g[i], out + i);
    }
    for (auto& kv : autos) {                           // an automorphism looked at before its key switch: coefficients as CiphertextPart::operator>>= leaves them, modulo 2^(64 nl)
      const auto& g = kv.second; const long G = (long)g.size();
      const std::vector<int32_t> idx = slots_of(g, false);
      const long tmp = alloc_run(G), out = alloc_run(G);
      on_all([&](int r) {
        uint64_t* bs = base_of(r);
        ck(fhesi_ct_gather_dev(h_of(r), bs, idx.data(), G, words, bs + tmp * words));
        ck(fhesi_ct_automorph_dev(h_of(r), (int64_t)kv.first, bs + tmp * words, 2, nl, G, bs + out * words, nl));
      });
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
inline void SyncCiphertexts(const FHEcontext& c) { ct_engine(c).flush(); ct_engine(c).sync_all(); }
// the recorded operations of this context's ciphertexts run on the GPUs `devices` from now on (devices[0] = the context's own); {} or one
// device: back to one GPU.  Keys are broadcast on first use, key-switch levels sharded, their outputs exchanged (see the head of this file).
inline void EnableCiphertextGroup(const FHEcontext& c, const std::vector<int>& devices) { ct_engine(c).enable_group(devices); }

}  // namespace fhesi
