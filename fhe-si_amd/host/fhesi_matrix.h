// fhesi_matrix.h -- mirror of Matrix<T> (Matrix.h:16-83, Matrix.cpp) and of Regression (Regression.h:68-191) on the classes of
// fhesi_host.h, plus the batched evaluator that SURVEY.md 8(f) ranks next after the multiplication path:
//
//   * Matrix<T>            the CONTAINER only (Matrix.h:16-83: storage, shape, transpose flag, element access, AddRow / Concatenate / MapAll and
//                          the wire format of fhesi_serialization.h).  The arithmetic of Matrix.cpp:20-263 is NOT restated in the package: in an
//                          integration the reference's own Matrix.cpp compiles unmodified on the mirror's recording Ciphertext (INTEGRATION.md
//                          section 3.1); the test harness keeps a literal restatement as its checker (tests/host/matrix_literal.h).
//   * ProductWave / WaveExecutor  device-resident unscaled ciphertexts addressed by pool index; one ProductWave = many independent
//                          "sum of products, then key switch" groups submitted as ONE fhesi_ct_mul_sum_relin_dev call per GPU
//                          (SingleGpuExecutor), or sharded over the GPUs of a node with the keys RCCL-broadcast and the wave
//                          outputs exchanged (GroupExecutor: one host thread per GPU).
//   * Regression           keys, data and RegressBatched(): the expression DAG of Regression::Regress (Regression.h:102-149) evaluated level by
//                          level (inner products -> SumBatchedData -> minors of growing size -> determinant -> adj * last) in waves.  Every ciphertext operation is deterministic, so equal minors
//                          the Laplace recursion of Matrix.cpp:227-263 recomputes are evaluated once; results are bit-identical to the
//                          literal object-at-a-time control flow (tests/host/matrix_literal.h: RegressLiteral).
// Slot packing (PlaintextSpace.cpp) is outside the hot-path scope: plaintexts are coefficient vectors and only the slot COUNT
// (Regression.h:72-79) is mirrored.  GenerateNoise (Regression.h:180-191) needs EmbedInSlots and is therefore not applied;
// both evaluators return the unmasked theta / det.
#pragma once
#include <functional>
#include <map>
#include <memory>
#include <thread>

#include "fhesi_host.h"

namespace fhesi {

// ---------------------------------------------------------------- Matrix<T> (Matrix.h:16-83, Matrix.cpp)
// Container only: what the wave evaluator, Regression's data store and the wire format need.  (Matrix.cpp's arithmetic written against
// this container and the recording Ciphertext is test infrastructure: tests/host/matrix_literal.h.)
template <class T>
class Matrix {
 protected:
  T dummy;
  std::vector<std::vector<T>> mat;      // storage rows; `transpose` swaps the roles of the two indices (Matrix.cpp:144-152)
  bool transpose = false;
  T& ElemAt(unsigned r, unsigned c) { return transpose ? mat[c][r] : mat[r][c]; }
  const T& ElemAt(unsigned r, unsigned c) const { return transpose ? mat[c][r] : mat[r][c]; }
 public:
  Matrix() : dummy(T()) {}
  Matrix(const T& d) : dummy(d) {}
  Matrix(unsigned nRows, unsigned nCols, const T& d) : dummy(d) { Resize(nRows, nCols); }
  Matrix(unsigned nRows, unsigned nCols) : dummy(T()) { Resize(nRows, nCols); }
  unsigned NumRows() const { return mat.empty() ? 0 : (unsigned)(transpose ? mat[0].size() : mat.size()); }
  unsigned NumCols() const { return mat.empty() ? 0 : (unsigned)(transpose ? mat.size() : mat[0].size()); }
  void Resize(unsigned nRows, unsigned nCols) { Clear(); transpose = false; mat.assign(nRows, std::vector<T>(nCols, dummy)); }
  void Clear() { mat.clear(); }
  void Transpose() { transpose = !transpose; }
  void AddRow(std::vector<T>& row) { if (!transpose) mat.push_back(row); }                 // no support on a transposed matrix (Matrix.cpp:293-297)
  void Concatenate(Matrix<T>& other) { if (!transpose) mat.insert(mat.end(), other.mat.begin(), other.mat.end()); }
  void MapAll(std::function<void(T&)> func) { for (auto& r : mat) for (auto& e : r) func(e); }   // storage order (Matrix.cpp:306-312)
  T& operator()(unsigned r, unsigned c) { return ElemAt(r, c); }
  const T& operator()(unsigned r, unsigned c) const { return ElemAt(r, c); }
  std::vector<T>& operator[](unsigned r) { return mat[r]; }
  const std::vector<T>& operator[](unsigned r) const { return mat[r]; }
  const T& Dummy() const { return dummy; }
};

// ---------------------------------------------------------------- slot counts (PlaintextSpace.cpp:29-43): factors of Phi_m mod p
inline unsigned TotalSlots(unsigned m, unsigned long p, unsigned phim) { unsigned d = 1; unsigned long x = p % m; while (x != 1) { x = (x * (p % m)) % m; ++d; } return phim / d; }
inline unsigned UsableSlots(unsigned m, unsigned long p, unsigned phim) { unsigned u = 1, t = TotalSlots(m, p, phim); while (t > 1) { u <<= 1; t >>= 1; } return u; }

// ---------------------------------------------------------------- wave executors
// A wave = a set of independent groups  out[g] = KeySwitch(sum_t pool[a_t] * pool[b_t])  over pool indices (ProductWave).  The pool
// holds unscaled 2-part ciphertexts [capacity][2][phi(m)][nl] in HBM; entries are written once and never modified afterwards, so
// indices can be shared freely (a symmetric matrix stores one entry for (i,j) and (j,i)).  RegressBatched (below) is written once
// against WaveExecutor; SingleGpuExecutor runs it on the context's GPU, GroupExecutor shards every wave over several GPUs.
struct ProductWave {
  std::vector<int32_t> a, b, seg{0};
  void product(long ai, long bi) { a.push_back((int32_t)ai); b.push_back((int32_t)bi); }
  long end_group() { seg.push_back((int32_t)a.size()); return (long)seg.size() - 2; }     // returns the group's position in the wave
  long groups() const { return (long)seg.size() - 1; }
};

struct WaveExecutor {
  virtual ~WaveExecutor() {}
  virtual void reserve(long entries) = 0;
  virtual long add(const Ciphertext& ct) = 0;                          // new pool entry <- an unscaled 2-part ciphertext
  virtual void get(long idx, Ciphertext& ct) = 0;
  // runs the wave, then (sum_batched) SumBatchedData on its outputs; returns the pool index of group 0 (groups are consecutive)
  virtual long run(const ProductWave& w, bool sum_batched) = 0;
  virtual long negated(const std::vector<int32_t>& idx) = 0;          // new entries = -1 * the given ones (Ciphertext.cpp:232-237)
};

// One KeySwitchSI matrix as a fhesi_ksk on the device of `h`
class DeviceKeySwitch {
  DeviceKeyRef shared;               // rank 0: the device object of the mirrored KeySwitchSI itself (no second copy of 300 MB, no second set of derived tables)
  fhesi_ksk* k = nullptr;            // other ranks: an empty replica of the same shape, the target of fhesi_ksk_broadcast
 public:
  DeviceKeySwitch(const FHEcontext& c, const KeySwitchSI& ks) : shared(ks.DeviceMatrix()) { ck(fhesi_ctx_sync(c.handle())); }
  DeviceKeySwitch(fhesi_ctx* h, int32_t ncomp, int32_t ndigits) { ck(fhesi_ksk_create(h, ncomp, ndigits, &k)); }
  ~DeviceKeySwitch() { if (k) fhesi_ksk_free(k); }
  DeviceKeySwitch(const DeviceKeySwitch&) = delete;
  fhesi_ksk* handle() const { return shared ? shared->k : k; }
};

// the pool and the keys of ONE GPU, and the work of one rank on a wave
class RankState {
 public:
  const FHEcontext& context;
  fhesi_ctx* h;                                   // this rank's device context
  bool owns_ctx;
  std::unique_ptr<DeviceKeySwitch> main;
  std::vector<std::unique_ptr<DeviceKeySwitch>> autos;
  uint64_t* d = nullptr;
  long cap = 0;
  const int nl;
  const long n, words;                            // words per ciphertext
  RankState(const FHEcontext& c, fhesi_ctx* hh, bool owns) : context(c), h(hh), owns_ctx(owns), nl((int)((c.logQ + 63) / 64)), n(c.zMstar.phiM()), words(2 * n * nl) {}
  ~RankState() { if (d) fhesi_dev_free(h, d); main.reset(); autos.clear(); if (owns_ctx && h) fhesi_ctx_destroy(h); }
  RankState(const RankState&) = delete;
  uint64_t* ptr(long idx) const { return d + idx * words; }
  void reserve(long want, long used) {
    if (want <= cap) return;
    long ncap = std::max(want, cap * 2);
    void* nd; ck(fhesi_dev_alloc(h, (size_t)ncap * words * 8, &nd));
    if (d) { ck(fhesi_dev_copy(h, nd, d, (size_t)used * words * 8)); ck(fhesi_dev_free(h, d)); }
    d = (uint64_t*)nd; cap = ncap;
  }
  // groups [lo, hi) of the wave -> pool entries first + lo .. first + hi, then SumBatchedData on them (Regression.h:166-178)
  void run_shard(const ProductWave& w, long first, long lo, long hi, bool sum_batched, const std::vector<unsigned>& autoK) {
    if (hi <= lo) return;
    const int32_t t0 = w.seg[lo];
    std::vector<int32_t> seg(w.seg.begin() + lo, w.seg.begin() + hi + 1);
    for (auto& v : seg) v -= t0;
    ck(fhesi_ct_mul_sum_relin_dev(h, main->handle(), (int32_t)context.logQ, (uint64_t)context.ModulusP().to_long(), (int32_t)context.decompSize, d, nl,
                                  w.a.data() + t0, w.b.data() + t0, seg.data(), hi - lo, ptr(first + lo)));
    if (!sum_batched || autos.empty()) return;
    const long count = hi - lo;
    void* tmp; ck(fhesi_dev_alloc(h, (size_t)count * words * 8, &tmp));
    for (size_t i = 0; i < autos.size(); ++i) {
      ck(fhesi_ct_automorph_key_switch_dev(h, autos[i]->handle(), (int32_t)context.logQ, (int32_t)context.decompSize, (int64_t)autoK[i], ptr(first + lo), nl, count, (uint64_t*)tmp, nl));
      ck(fhesi_ct_add_dev(h, (int32_t)context.logQ, ptr(first + lo), (const uint64_t*)tmp, 2, nl, count));
    }
    ck(fhesi_dev_free(h, tmp));
  }
  void negate_into(const std::vector<int32_t>& idx, long first) {
    ck(fhesi_ct_gather_dev(h, d, idx.data(), (int64_t)idx.size(), words, ptr(first)));
    ck(fhesi_ct_mul_long_dev(h, (int32_t)context.logQ, ptr(first), -1, 2, nl, (int64_t)idx.size()));
  }
  void upload(const Ciphertext& ct, long idx) {
    if (LazyCiphertexts() && h == context.handle() && ct.parts.resident()) {      // already a value in HBM of this GPU (fhesi_engine.h): device to device
      CtEngine& e = ct_engine(context); CtRef v = ct.parts.value(); e.force(v);
      ck(fhesi_dev_copy(h, ptr(idx), e.ptr(v->slot), (size_t)words * 8));
      return;
    }
    std::vector<uint64_t> v(words);
    for (int part = 0; part < 2; ++part) for (long i = 0; i < n; ++i) coeff(ct.parts[part].poly, i).to_limbs(&v[(part * n + i) * nl], nl);
    ck(fhesi_dev_upload(h, ptr(idx), v.data(), (size_t)words * 8));
  }
  void download(long idx, Ciphertext& ct) const {
    if (LazyCiphertexts() && h == context.handle()) {                             // stays in HBM; the host form is made when somebody reads parts[i].poly
      CtEngine& e = ct_engine(context); const long s = e.alloc_run(1);
      ck(fhesi_dev_copy(h, e.ptr(s), ptr(idx), (size_t)words * 8));
      e.publish(s, 1);
      ct = Ciphertext(context); ct.set_device_value(e.wrap(s));
      return;
    }
    std::vector<uint64_t> v(words);
    ck(fhesi_dev_download(h, v.data(), ptr(idx), (size_t)words * 8));
    ct.Initialize(2, context);
    for (int part = 0; part < 2; ++part) { ZZX p; p.rep.resize(n); for (long i = 0; i < n; ++i) p.rep[i] = ZZ::from_limbs(&v[(part * n + i) * nl], nl); p.normalize(); ct[part].poly = p; }
  }
};

// (shard_bounds: fhesi_engine.h)

class SingleGpuExecutor : public WaveExecutor {
  RankState rs;
  const std::vector<unsigned>& autoK;
  long used = 0;
 public:
  SingleGpuExecutor(const FHEcontext& c, const KeySwitchSI& ks, const std::vector<KeySwitchSI>& autoKs, const std::vector<unsigned>& k) : rs(c, c.handle(), false), autoK(k) {
    rs.main.reset(new DeviceKeySwitch(c, ks));
    for (auto& a : autoKs) rs.autos.emplace_back(new DeviceKeySwitch(c, a));
  }
  void reset() { used = 0; }
  void reserve(long entries) override { rs.reserve(entries, used); }
  long add(const Ciphertext& ct) override {
    if (ct.isScaledUp() || ct.parts.size() != 2) Error("pool: expects an unscaled 2-part ciphertext");
    rs.reserve(used + 1, used); rs.upload(ct, used); return used++;
  }
  void get(long idx, Ciphertext& ct) override { rs.download(idx, ct); }
  long run(const ProductWave& w, bool sum_batched) override {
    if (!w.groups()) return used;
    rs.reserve(used + w.groups(), used);
    const long first = used; used += w.groups();
    rs.run_shard(w, first, 0, w.groups(), sum_batched, autoK);
    return first;
  }
  long negated(const std::vector<int32_t>& idx) override {
    if (idx.empty()) return used;
    rs.reserve(used + (long)idx.size(), used);
    const long first = used; used += (long)idx.size();
    rs.negate_into(idx, first);
    return first;
  }
};

// Several GPUs of one node, one host thread per GPU: every rank holds the context tables, a replica of every
// key-switch matrix (RCCL broadcast from rank 0, where the keys were generated: KeySwitchSI::keySwitchMatrix, FHE-SI.cpp:206-208) and
// a full copy of the pool.  The groups of a wave are sharded over the ranks by shard_bounds; the wave's outputs are then exchanged
// (fhesi_comm_exchange: one grouped RCCL broadcast per producing rank) because the next wave reads ciphertexts produced by every
// rank.  Every ciphertext operation is deterministic, so the pool contents -- and the results -- are bit-identical to one GPU.
// Exchange / compute overlap (option, SetOverlap): a wave's outputs are read by the NEXT wave only, so a wave is cut into chunks of groups,
// every chunk sharded over all ranks; the exchange of chunk k (fhesi_comm_exchange_begin: the communicator's own stream, behind an event of
// the compute stream) travels while chunk k + 1 is computed, and fhesi_comm_exchange_end closes the wave.  The group -> rank assignment
// differs from the unchunked one; the ciphertexts do not (every group is computed by exactly one rank, deterministically).
class GroupExecutor : public WaveExecutor {
  const FHEcontext& context;
  std::vector<std::unique_ptr<RankState>> ranks;
  std::vector<fhesi_comm*> comms;
  const std::vector<unsigned>& autoK;
  long used = 0;
  int overlap_chunks = 1;                          // chunks per wave (1 = compute the whole shard, then exchange: no overlap)
  std::vector<std::string> schedule_log;           // what ran, wave by wave (the overlap cannot be TIMED on ranks that share one GPU; it can be read)
  template <class F> void parallel(F f) {
    std::vector<std::thread> th;
    for (size_t r = 1; r < ranks.size(); ++r) th.emplace_back([&f, r] { f((int)r); });
    f(0);
    for (auto& t : th) t.join();
  }
  void exchange(long first, long count) {
    const int G = (int)ranks.size();
    std::vector<int64_t> off(G + 1);
    for (int r = 0; r < G; ++r) { long lo, hi; shard_bounds(count, r, G, lo, hi); off[r] = (first + lo) * ranks[0]->words; off[r + 1] = (first + hi) * ranks[0]->words; }
    parallel([&](int r) { ck(fhesi_comm_exchange(ranks[r]->h, comms[r], ranks[r]->d, off.data())); });
  }
 public:
  // devices[0] must be the context's own GPU (its keys live there); a repeated device makes a loopback group (see fhesi_comm_init_all)
  GroupExecutor(const FHEcontext& c, const std::vector<int>& devices, const KeySwitchSI& ks, const std::vector<KeySwitchSI>& autoKs, const std::vector<unsigned>& k)
      : context(c), autoK(k) {
    const int G = (int)devices.size();
    if (G < 1 || devices[0] != c.deviceIndex()) Error("GroupExecutor: devices[0] must be the context's GPU");
    std::vector<int32_t> devs(devices.begin(), devices.end());
    comms.assign(G, nullptr);
    ck(fhesi_comm_init_all(G, devs.data(), comms.data()));
    for (int r = 0; r < G; ++r) ranks.emplace_back(new RankState(c, r == 0 ? c.handle() : c.replica(devices[r]), r != 0));
    // keys: rank 0 from the mirrored objects, the others as empty replicas, then ONE broadcast per matrix over xGMI
    ranks[0]->main.reset(new DeviceKeySwitch(c, ks));
    for (auto& a : autoKs) ranks[0]->autos.emplace_back(new DeviceKeySwitch(c, a));
    for (int r = 1; r < G; ++r) {
      ranks[r]->main.reset(new DeviceKeySwitch(ranks[r]->h, 3, (int32_t)c.ndigits));
      for (size_t i = 0; i < autoKs.size(); ++i) ranks[r]->autos.emplace_back(new DeviceKeySwitch(ranks[r]->h, 2, (int32_t)c.ndigits));
    }
    parallel([&](int r) {
      ck(fhesi_ksk_broadcast(ranks[r]->main->handle(), comms[r], 0));
      for (auto& a : ranks[r]->autos) ck(fhesi_ksk_broadcast(a->handle(), comms[r], 0));
    });
  }
  ~GroupExecutor() { ranks.clear(); for (auto cm : comms) fhesi_comm_destroy(cm); }
  int world() const { return (int)ranks.size(); }
  void reset() { used = 0; schedule_log.clear(); }
  void SetOverlap(int chunks) { overlap_chunks = chunks < 1 ? 1 : chunks; }
  const std::vector<std::string>& Schedule() const { return schedule_log; }
  void reserve(long entries) override { for (auto& r : ranks) r->reserve(entries, used); }
  long add(const Ciphertext& ct) override {
    if (ct.isScaledUp() || ct.parts.size() != 2) Error("pool: expects an unscaled 2-part ciphertext");
    reserve(used + 1);
    for (auto& r : ranks) r->upload(ct, used);          // inputs are replicated (host -> every GPU)
    return used++;
  }
  void get(long idx, Ciphertext& ct) override { ranks[0]->download(idx, ct); }
  long run(const ProductWave& w, bool sum_batched) override {
    if (!w.groups()) return used;
    reserve(used + w.groups());
    const long first = used, count = w.groups(); used += count;
    const int G = (int)ranks.size();
    // chunks of at least one group per rank; a single chunk is the plain form (compute, then the blocking exchange)
    const int C = (int)std::max<long>(1, std::min<long>(overlap_chunks, count / G));
    if (C == 1) {
      parallel([&](int r) { long lo, hi; shard_bounds(count, r, G, lo, hi); ranks[r]->run_shard(w, first, lo, hi, sum_batched, autoK); });
      exchange(first, count);
      schedule_log.push_back("wave of " + std::to_string(count) + " groups: compute, then exchange");
      return first;
    }
    std::string log = "wave of " + std::to_string(count) + " groups in " + std::to_string(C) + " chunks:";
    parallel([&](int r) {
      std::vector<int64_t> off(G + 1);
      for (int c = 0; c < C; ++c) {
        long c0, c1; shard_bounds(count, c, C, c0, c1);                 // chunk c = groups [c0, c1), sharded over ALL ranks
        long lo, hi; shard_bounds(c1 - c0, r, G, lo, hi);
        ranks[r]->run_shard(w, first, c0 + lo, c0 + hi, sum_batched, autoK);
        for (int q = 0; q < G; ++q) { long ql, qh; shard_bounds(c1 - c0, q, G, ql, qh); off[q] = (first + c0 + ql) * ranks[0]->words; off[q + 1] = (first + c0 + qh) * ranks[0]->words; }
        ck(fhesi_comm_exchange_begin(ranks[r]->h, comms[r], ranks[r]->d, off.data()));      // travels while chunk c + 1 is computed
      }
      ck(fhesi_comm_exchange_end(ranks[r]->h, comms[r]));
    });
    for (int c = 0; c < C; ++c) { long c0, c1; shard_bounds(count, c, C, c0, c1); log += " [" + std::to_string(c0) + "," + std::to_string(c1) + ") compute -> exchange begun" + (c + 1 < C ? " (overlaps the next chunk's compute);" : ";"); }
    schedule_log.push_back(log + " exchange end");
    return first;
  }
  long negated(const std::vector<int32_t>& idx) override {      // cheap and local: every rank computes all of them
    if (idx.empty()) return used;
    reserve(used + (long)idx.size());
    const long first = used; used += (long)idx.size();
    parallel([&](int r) { ranks[r]->negate_into(idx, first); });
    return first;
  }
};

// ---------------------------------------------------------------- Regression (Regression.h:68-191)
class Regression {
  const FHEcontext& context;
  FHESISecKey secretKey;
  FHESIPubKey publicKey;
  KeySwitchSI keySwitch;
  std::vector<KeySwitchSI> autoKeySwitch;
  std::vector<unsigned> autoK;                                   // k = g, g^2, g^4, ... mod m (Regression.h:71-80)
  Matrix<Ciphertext> data;
  std::unique_ptr<SingleGpuExecutor> single;                     // device copies of the keys, created on the first batched call
  std::unique_ptr<GroupExecutor> group;

  void SumBatchedData(Ciphertext& batched) const {               // Regression.h:166-178
    for (size_t i = 0; i < autoKeySwitch.size(); ++i) {
      Ciphertext tmp = batched;
      tmp >>= (long)autoK[i];
      autoKeySwitch[i].ApplyKeySwitch(tmp);
      batched += tmp;
    }
  }

 public:
  std::vector<Ciphertext> labels;
  struct Stats { long products = 0, key_switches = 0, automorph_key_switches = 0, waves = 0; } stats;    // work submitted by the last RegressBatched

  Regression(const FHEcontext& c) : context(c), secretKey(c), publicKey(secretKey), keySwitch(secretKey), data(Ciphertext(c)) {
    unsigned k = c.Generator();
    unsigned nSlots = UsableSlots(c.zMstar.M(), (unsigned long)c.ModulusP().to_long(), c.zMstar.phiM());
    while (nSlots > 1) {
      autoKeySwitch.push_back(KeySwitchSI(secretKey, k));
      autoK.push_back(k);
      nSlots >>= 1;
      k = (unsigned)(((unsigned long)k * k) % c.zMstar.M());
    }
  }
  FHESIPubKey& GetPublicKey() { return publicKey; }
  FHESISecKey& GetSecretKey() { return secretKey; }
  const std::vector<unsigned>& AutomorphismExponents() const { return autoK; }

  void AddData(const std::vector<std::vector<Plaintext>>& ptxtData, const std::vector<Plaintext>& ptxtLabels) {   // Regression.h:83-95
    for (size_t i = 0; i < ptxtData.size(); ++i) {
      std::vector<Ciphertext> row(ptxtData[i].size(), Ciphertext(context));
      for (size_t j = 0; j < ptxtData[i].size(); ++j) publicKey.Encrypt(row[j], ptxtData[i][j]);
      Ciphertext lab(context);
      publicKey.Encrypt(lab, ptxtLabels[i]);
      data.AddRow(row);
      labels.push_back(lab);
    }
  }
  void Clear() { data.Clear(); labels.clear(); }

  // access for the object-at-a-time checker (tests/host/matrix_literal.h)
  const Matrix<Ciphertext>& Data() const { return data; }
  const KeySwitchSI& KeySwitch() const { return keySwitch; }
  const FHEcontext& Context() const { return context; }
  void SumBatchedDataObject(Ciphertext& ct) const { SumBatchedData(ct); }

  // The expressions of Regression::Regress (Regression.h:102-149) as explicit waves on the context's GPU
  void RegressBatched(std::vector<Ciphertext>& theta, Ciphertext& det) {
    if (!single) single.reset(new SingleGpuExecutor(context, keySwitch, autoKeySwitch, autoK));
    single->reset();
    RegressWaves(*single, theta, det);
  }
  // ... sharded over the GPUs `devices` of this node (devices[0] = the context's GPU); keys are broadcast on the first call
  // overlap_chunks > 1: every wave in that many chunks, the exchange of a chunk overlapped with the next chunk's compute (GroupExecutor)
  void RegressBatchedMultiGpu(const std::vector<int>& devices, std::vector<Ciphertext>& theta, Ciphertext& det, int overlap_chunks = 1) {
    if (!group || group->world() != (int)devices.size()) group.reset(new GroupExecutor(context, devices, keySwitch, autoKeySwitch, autoK));
    group->reset();
    group->SetOverlap(overlap_chunks);
    RegressWaves(*group, theta, det);
  }
  const std::vector<std::string>& LastSchedule() const { static const std::vector<std::string> none; return group ? group->Schedule() : none; }

  // The expression DAG of Regression::Regress level by level (inner products -> SumBatchedData -> minors of growing size ->
  // determinant -> adj * last), every level one wave on the executor
  void RegressWaves(WaveExecutor& ex, std::vector<Ciphertext>& theta, Ciphertext& det) {
    stats = Stats();
    const unsigned N = data.NumRows(), d = data.NumCols();
    if (!N || !d) Error("Regression: no data");
    ex.reserve((long)N * (d + 1) + 4L * d * d + 64);
    std::vector<std::vector<long>> X(N, std::vector<long>(d));
    std::vector<long> y(N);
    for (unsigned i = 0; i < N; ++i) { for (unsigned j = 0; j < d; ++j) X[i][j] = ex.add(data(i, j)); y[i] = ex.add(labels[i]); }
    // wave 1: last = X^T y (Matrix.cpp:81-98) and the upper triangle of X^T X (Matrix.cpp:150-174), then key switch
    ProductWave w1;
    for (unsigned j = 0; j < d; ++j) { for (unsigned i = 0; i < N; ++i) w1.product(X[i][j], y[i]); w1.end_group(); }
    std::vector<std::vector<long>> A(d, std::vector<long>(d, -1));
    for (unsigned i = 0; i < d; ++i) for (unsigned j = i; j < d; ++j) { for (unsigned k = 0; k < N; ++k) w1.product(X[k][i], X[k][j]); w1.end_group(); }
    const long first1 = ex.run(w1, true);                         // processFunc: ApplyKeySwitch + SumBatchedData (Regression.h:112-117)
    note(w1);
    stats.automorph_key_switches += w1.groups() * (long)autoKeySwitch.size();
    std::vector<long> last(d);
    for (unsigned j = 0; j < d; ++j) last[j] = first1 + j;
    { long g = d; for (unsigned i = 0; i < d; ++i) for (unsigned j = i; j < d; ++j) { A[i][j] = A[j][i] = first1 + g; ++g; } }
    if (d == 1) { ex.get(A[0][0], det); theta.assign(1, Ciphertext(context)); ex.get(last[0], theta[0]); return; }
    // negated copies of the matrix entries: the `tmp *= -1` of the expansion (Matrix.cpp:245) acts on the unscaled entry
    std::vector<int32_t> flat;
    for (unsigned i = 0; i < d; ++i) for (unsigned j = 0; j < d; ++j) flat.push_back((int32_t)A[i][j]);
    const long negA = ex.negated(flat);
    auto entry = [&](unsigned r, unsigned c, bool neg) { return neg ? negA + (long)r * d + c : A[r][c]; };
    // minors needed by Invert (Matrix.cpp:182-200): Determinant with row i and column j struck out, dim d-1, memoised on
    // (used rows, used columns); level s holds the partial determinants of size s
    typedef std::pair<unsigned, unsigned> Key;
    std::vector<std::map<Key, long>> level(d);
    std::function<void(unsigned, unsigned, unsigned)> need = [&](unsigned R, unsigned C, unsigned dim) {
      if (level[dim].count(Key(R, C))) return;
      level[dim][Key(R, C)] = -1;
      if (dim == 1) return;
      unsigned row = 0; while (R >> row & 1) ++row;
      for (unsigned col = 0; col < d; ++col) if (!(C >> col & 1)) need(R | 1u << row, C | 1u << col, dim - 1);
    };
    for (unsigned i = 0; i < d; ++i) for (unsigned j = 0; j < d; ++j) need(1u << i, 1u << j, d - 1);
    for (auto& kv : level[1]) {                                   // size 1: the entry itself, no reduce (Matrix.cpp:238-241)
      unsigned row = 0; while (kv.first.first >> row & 1) ++row;
      unsigned col = 0; while (kv.first.second >> col & 1) ++col;
      kv.second = A[row][col];
    }
    for (unsigned s = 2; s + 1 <= d; ++s) {
      ProductWave w;
      std::vector<Key> order;
      for (auto& kv : level[s]) {
        const unsigned R = kv.first.first, C = kv.first.second;
        unsigned row = 0; while (R >> row & 1) ++row;
        bool negative = false;
        for (unsigned col = 0; col < d; ++col) {
          if (C >> col & 1) continue;
          w.product(entry(row, col, negative), level[s - 1][Key(R | 1u << row, C | 1u << col)]);
          negative = !negative;
        }
        w.end_group();
        order.push_back(kv.first);
      }
      const long first = ex.run(w, false);
      note(w);
      for (size_t g = 0; g < order.size(); ++g) level[s][order[g]] = first + (long)g;
    }
    // adjugate: adj(j,i) = (-1)^(i+j) * minor(i,j)  (Matrix.cpp:192-199; the sign acts on the reduced, unscaled minor)
    std::vector<std::vector<long>> adj(d, std::vector<long>(d));
    std::vector<int32_t> toNeg;
    for (unsigned i = 0; i < d; ++i) for (unsigned j = 0; j < d; ++j) { adj[j][i] = level[d - 1][Key(1u << i, 1u << j)]; if ((i + j) % 2 == 1) toNeg.push_back((int32_t)adj[j][i]); }
    { const long firstNeg = ex.negated(toNeg); long g = 0; for (unsigned i = 0; i < d; ++i) for (unsigned j = 0; j < d; ++j) if ((i + j) % 2 == 1) adj[j][i] = firstNeg + g++; }
    // det = sum_i A(0,i) adj(i,0) (Matrix.cpp:202-212), and theta = adj * last (Matrix.cpp:57-79 + MapAll key switch, Regression.h:131-134)
    ProductWave wf;
    for (unsigned i = 0; i < d; ++i) wf.product(A[0][i], adj[i][0]);
    wf.end_group();
    for (unsigned i = 0; i < d; ++i) { for (unsigned k = 0; k < d; ++k) wf.product(adj[i][k], last[k]); wf.end_group(); }
    const long firstF = ex.run(wf, false);
    note(wf);
    ex.get(firstF, det);
    theta.assign(d, Ciphertext(context));
    for (unsigned i = 0; i < d; ++i) ex.get(firstF + 1 + i, theta[i]);
  }

 private:
  void note(const ProductWave& w) { stats.products += (long)w.a.size(); stats.key_switches += w.groups(); stats.waves += 1; }
};

}  // namespace fhesi
