// fhesi_keys.h -- part of the C++ mirror of the reference's class surface (see fhesi_host.h, which includes the parts in order; not a
// standalone header): FHE-SI.cpp: FHESISecKey, SeedSequence, FHESIPubKey, KeySwitchSI.
#pragma once

namespace fhesi {

// ---------------------------------------------------------------- FHE-SI.cpp: keys and key switching
class FHESISecKey {
  const FHEcontext& context;
  std::vector<DoubleCRT> sKeys;
 public:
  FHESISecKey(const FHEcontext& c) : context(c) { Init(c); }
  void Init(const FHEcontext& c) { sKeys.assign(2, DoubleCRT(c)); sKeys[0] = 1L; sKeys[1].sampleHWt(64); }   // FHE-SI.cpp:86-91
  const std::vector<DoubleCRT>& GetRepresentation() const { return sKeys; }
  void UpdateRepresentation(const std::vector<DoubleCRT>& r) { sKeys = r; }
  const FHEcontext& GetContext() const { return context; }
  size_t GetSize() const { return sKeys.size(); }
  // Decrypt for many unscaled 2-part ciphertexts in one device call (fhesi_decrypt_batch); same values as repeated Decrypt calls
  void DecryptBatch(std::vector<Plaintext>& ptxts, const std::vector<Ciphertext>& ctxts) const {
    const long n = context.zMstar.phiM(), count = (long)ctxts.size(); const int nl = (int)((context.logQ + 63) / 64);
    std::vector<int64_t> msg((size_t)count * n);
    if (LazyCiphertexts() && count) {
      // the ciphertexts as values in HBM (whatever was recorded for them runs now), gathered into one run of the arena
      CtEngine& e = ct_engine(context);
      std::vector<CtRef> vals; for (auto& c : ctxts) vals.push_back(c.device_value());
      e.flush();
      std::vector<int32_t> idx; for (auto& v : vals) { e.force(v); idx.push_back((int32_t)v->slot); }
      const long run = e.alloc_run(count);
      ck(fhesi_ct_gather_dev(e.h, e.pool(), idx.data(), count, e.words, e.ptr(run)));
      int rc = fhesi_decrypt_batch(e.h, sKeys[1].handle(), (int32_t)context.logQ, (uint64_t)context.ModulusP().to_long(), e.ptr(run), nl, count, msg.data());
      e.free_run(run, count);
      ck(rc);
    } else {
      std::vector<uint64_t> host((size_t)count * 2 * n * nl);
      for (long c = 0; c < count; ++c) for (int part = 0; part < 2; ++part) for (long j = 0; j < n; ++j) coeff(ctxts[c].GetPart((unsigned)part).poly, j).to_limbs(&host[((c * 2 + part) * n + j) * nl], nl);
      void* dev; ck(fhesi_dev_alloc(context.handle(), host.size() * 8, &dev)); ck(fhesi_dev_upload(context.handle(), dev, host.data(), host.size() * 8));
      ck(fhesi_decrypt_batch(context.handle(), sKeys[1].handle(), (int32_t)context.logQ, (uint64_t)context.ModulusP().to_long(), (const uint64_t*)dev, nl, count, msg.data()));
      ck(fhesi_dev_free(context.handle(), dev));
    }
    ptxts.assign(count, Plaintext());
    for (long c = 0; c < count; ++c) ptxts[c].message.assign(msg.begin() + c * n, msg.begin() + (c + 1) * n);
  }
  void Decrypt(Plaintext& ptxt, const Ciphertext& ctxt) const {   // FHE-SI.cpp:93-119
    if (LazyCiphertexts() && !ctxt.isScaledUp() && ctxt.parts.resident() && sKeys.size() == 2) {
      // the ciphertext lives in HBM: the same dot product with (1, t), rounding and reduction as ONE device call on it (fhesi_decrypt_batch)
      CtEngine& e = ct_engine(context); CtRef v = ctxt.parts.value(); e.force(v);
      std::vector<int64_t> msg((size_t)e.n);
      ck(fhesi_decrypt_batch(e.h, sKeys[1].handle(), (int32_t)context.logQ, (uint64_t)context.ModulusP().to_long(), e.ptr(v->slot), e.nl, 1, msg.data()));
      ptxt.message.assign(msg.begin(), msg.end());
      return;
    }
    std::vector<DoubleCRT> cp, sp;
    for (size_t i = 0; i < sKeys.size(); ++i) { cp.push_back(DoubleCRT(ctxt.GetPart((unsigned)i).poly, context)); sp.push_back(sKeys[i]); }
    DoubleCRT tmp(context); DotProduct(tmp, cp, sp);
    ZZX z; tmp.toPoly(z);
    ZZ p = context.ModulusP(), q = context.modulusQ, q2 = q * ZZ(2L);
    ptxt.message.assign(context.zMstar.phiM(), 0);
    for (long i = 0; i <= deg(z); ++i) { ZZ c = z.rep[i]; c *= ZZ(2L) * p; c += q; c /= q2; ptxt.message[i] = rem(c, p.to_long()); }
  }
};
// One stream of on-device randomness (csrc/philox.h): the secret seed, the public seed of the key polynomials, and ONE monotonically
// increasing object counter shared by every Encrypt and every KeySwitchSI that draws from it -- callers never pick indices, so a pair
// (seed, index) cannot be handed out twice.  Both seeds must be uniformly random and the secret one stays secret; Philox is not a CSPRNG
// (64-bit key): see include/fhesi_hip.h for what that is good for.
struct SeedSequence {
  const uint64_t seed, public_seed;
  SeedSequence(uint64_t secret, uint64_t pub, uint64_t first = 0) : seed(secret), public_seed(pub), next(first) { if (secret == pub) Error("SeedSequence: the public seed must differ from the secret seed"); }
  uint64_t take(uint64_t count) { return next.fetch_add(count); }      // first index of a fresh range of `count` objects
  uint64_t used() const { return next.load(); }
 private:
  std::atomic<uint64_t> next;
};

class FHESIPubKey {
  const FHEcontext& context;
  std::vector<DoubleCRT> publicKey;
 public:
  FHESIPubKey(const FHESISecKey& sk) : context(sk.GetContext()) { Init(sk); }
  const FHEcontext& GetContext() const { return context; }
  const std::vector<DoubleCRT>& GetRepresentation() const { return publicKey; }
  void UpdateRepresentation(const std::vector<DoubleCRT>& r) { publicKey = r; }
  void Init(const FHESISecKey& sk) {   // FHE-SI.cpp:42-63
    ZZX c0, c1; sampleGaussian(c0, context.zMstar.phiM(), context.stdev); SampleRandom(c1, context.modulusQ, context.zMstar.phiM());
    ZZX tmp; sk.GetRepresentation()[1].toPoly(tmp); tmp = mul(tmp, c1);
    c0 += tmp; rem(c0, c0, context.zMstar.PhimX()); c1 *= ZZ(-1L);
    ReduceCoefficients(c0, context.logQ); ReduceCoefficients(c1, context.logQ);
    publicKey.clear(); publicKey.push_back(DoubleCRT(c0, context)); publicKey.push_back(DoubleCRT(c1, context));
  }
  // Encrypt for many plaintexts in one device call (fhesi_encrypt_batch).  The randomness is drawn here, per plaintext, in the
  // order Encrypt draws it (r, noise of part 0, noise of part 1), so the ciphertexts equal those of repeated Encrypt calls.
  void EncryptBatch(std::vector<Ciphertext>& ctxts, const std::vector<Plaintext>& ptxts) const {
    const long n = context.zMstar.phiM(), count = (long)ptxts.size(); const int nl = (int)((context.logQ + 63) / 64);
    std::vector<int64_t> rnd((size_t)count * 3 * n), msg((size_t)count * n, 0);
    for (long c = 0; c < count; ++c) {
      for (long j = 0; j < n; ++j) rnd[(c * 3) * n + j] = RandomBnd(2L);
      for (int i = 0; i < 2; ++i) { ZZX e; sampleGaussian(e, n, context.stdev); for (long j = 0; j < n; ++j) rnd[(c * 3 + 1 + i) * n + j] = coeff(e, j).to_long(); }
      for (size_t k = 0; k < ptxts[c].message.size() && (long)k < n; ++k) msg[c * n + k] = ptxts[c].message[k];
    }
    if (LazyCiphertexts() && count) {            // the ciphertexts stay in HBM, as consecutive slots of the arena
      CtEngine& e = ct_engine(context); const long first = e.alloc_run(count);
      int rc = fhesi_encrypt_batch(e.h, publicKey[0].handle(), publicKey[1].handle(), (int32_t)context.logQ, (uint64_t)context.ModulusP().to_long(), rnd.data(), msg.data(), count, e.ptr(first), nl);
      if (rc) { e.free_run(first, count); ck(rc); }
      e.publish(first, count);
      ctxts.assign(count, Ciphertext(context));
      for (long c = 0; c < count; ++c) ctxts[c].set_device_value(e.wrap(first + c));
      return;
    }
    void* dev; ck(fhesi_dev_alloc(context.handle(), (size_t)count * 2 * n * nl * 8, &dev));
    ck(fhesi_encrypt_batch(context.handle(), publicKey[0].handle(), publicKey[1].handle(), (int32_t)context.logQ, (uint64_t)context.ModulusP().to_long(), rnd.data(), msg.data(),
                           count, (uint64_t*)dev, nl));
    std::vector<uint64_t> host((size_t)count * 2 * n * nl);
    ck(fhesi_dev_download(context.handle(), host.data(), dev, host.size() * 8)); ck(fhesi_dev_free(context.handle(), dev));
    ctxts.assign(count, Ciphertext(context));
    for (long c = 0; c < count; ++c) {
      ctxts[c].Initialize(2, context);
      for (int part = 0; part < 2; ++part) { ZZX poly; poly.rep.resize(n); for (long j = 0; j < n; ++j) poly.rep[j] = ZZ::from_limbs(&host[((c * 2 + part) * n + j) * nl], nl); poly.normalize(); ctxts[c][part].poly = poly; }
    }
  }
  // ... with r and the noise drawn ON THE DEVICE from the counter-based generator (fhesi_encrypt_batch_seeded, csrc/philox.h): plaintext i
  // uses the streams of object index first + i, so a batch can be split or repeated anywhere and give the same ciphertexts
  // (no default index: an (seed, index) pair used twice repeats r, e0, e1 -- the difference of the two ciphertexts is delta (m1 - m2) in the clear)
  void EncryptBatchSeeded(std::vector<Ciphertext>& ctxts, const std::vector<Plaintext>& ptxts, SeedSequence& seq) const { EncryptBatchSeeded(ctxts, ptxts, seq.seed, seq.take(ptxts.size())); }
  void EncryptBatchSeeded(std::vector<Ciphertext>& ctxts, const std::vector<Plaintext>& ptxts, uint64_t seed, uint64_t first_obj) const {
    const long n = context.zMstar.phiM(), count = (long)ptxts.size(); const int nl = (int)((context.logQ + 63) / 64);
    std::vector<int64_t> msg((size_t)count * n, 0);
    for (long c = 0; c < count; ++c) for (size_t k = 0; k < ptxts[c].message.size() && (long)k < n; ++k) msg[c * n + k] = ptxts[c].message[k];
    if (LazyCiphertexts() && count) {
      CtEngine& e = ct_engine(context); const long first = e.alloc_run(count);
      int rc = fhesi_encrypt_batch_seeded(e.h, publicKey[0].handle(), publicKey[1].handle(), (int32_t)context.logQ, (uint64_t)context.ModulusP().to_long(), seed, first_obj, msg.data(), count, e.ptr(first), nl);
      if (rc) { e.free_run(first, count); ck(rc); }
      e.publish(first, count);
      ctxts.assign(count, Ciphertext(context));
      for (long c = 0; c < count; ++c) ctxts[c].set_device_value(e.wrap(first + c));
      return;
    }
    void* dev; ck(fhesi_dev_alloc(context.handle(), (size_t)count * 2 * n * nl * 8, &dev));
    ck(fhesi_encrypt_batch_seeded(context.handle(), publicKey[0].handle(), publicKey[1].handle(), (int32_t)context.logQ, (uint64_t)context.ModulusP().to_long(), seed, first_obj, msg.data(), count, (uint64_t*)dev, nl));
    std::vector<uint64_t> host((size_t)count * 2 * n * nl);
    ck(fhesi_dev_download(context.handle(), host.data(), dev, host.size() * 8)); ck(fhesi_dev_free(context.handle(), dev));
    ctxts.assign(count, Ciphertext(context));
    for (long c = 0; c < count; ++c) { ctxts[c].Initialize(2, context); for (int part = 0; part < 2; ++part) limbs_to_poly(ctxts[c][part].poly, &host[((size_t)(c * 2 + part) * n) * nl], n, nl); }
  }
  // Encrypt (FHE-SI.cpp:10-36).  With recording on, the randomness is drawn here exactly as below and the arithmetic is the device call of
  // EncryptBatch on one plaintext; the ciphertext stays in HBM (the same bits: tests/host/test_wire.cpp compares EncryptBatch with EncryptObjects)
  void Encrypt(Ciphertext& ctxt, const Plaintext& ptxt) const {
    if (!LazyCiphertexts()) { EncryptObjects(ctxt, ptxt); return; }
    std::vector<Ciphertext> one;
    EncryptBatch(one, std::vector<Plaintext>(1, ptxt));
    ctxt = one[0];
  }
  void EncryptObjects(Ciphertext& ctxt, const Plaintext& ptxt) const {   // the reference's body, one DoubleCRT object at a time
    ctxt.Initialize(2, context);
    ZZX small; small.rep.assign(context.zMstar.phiM(), ZZ());
    for (auto& c : small.rep) c = ZZ(RandomBnd(2L));
    small.normalize();
    DoubleCRT r(small, context), e(context);
    std::vector<DoubleCRT> ct = publicKey;
    for (size_t i = 0; i < ct.size(); ++i) { e.sampleGaussian(); e *= context.ModulusP(); ct[i] *= r; ct[i] += e; ct[i].toPoly(ctxt[(unsigned)i].poly); }
    ZZ delta = context.modulusQ / context.ModulusP(); ZZX msg;
    for (size_t k = 0; k < ptxt.message.size(); ++k) SetCoeff(msg, (long)k, ZZ(ptxt.message[k]));
    ctxt[0] += delta * msg;
    for (size_t i = 0; i < ct.size(); ++i) ReduceCoefficients(ctxt[(unsigned)i].poly, context.logQ);
  }
};
class KeySwitchSI {
  const FHEcontext& context;
  std::vector<std::vector<DoubleCRT>> keySwitchMatrix;
  bool objectAtATime = false;          // checker mode: build the matrix with the reference's per-object loop (InitObjects)
  void InitAny(const FHESISecKey& src, const FHESISecKey& dst) { if (objectAtATime) InitObjects(src, dst); else Init(src, dst); }
 public:
  struct ObjectAtATime {};
  KeySwitchSI(const FHESISecKey& s, ObjectAtATime) : context(s.GetContext()), objectAtATime(true) { InitS2(s); }
  KeySwitchSI(const FHESISecKey& s) : context(s.GetContext()) { InitS2(s); }
  KeySwitchSI(const FHESISecKey& src, const FHESISecKey& dst) : context(src.GetContext()) { Init(src, dst); }
  const std::vector<std::vector<DoubleCRT>>& GetRepresentation() const { return keySwitchMatrix; }
  void UpdateRepresentation(const std::vector<std::vector<DoubleCRT>>& rep) { keySwitchMatrix = rep; drop_device_key(); }
  const FHEcontext& GetContext() const { return context; }
  // FHE-SI.cpp:153-209.  The randomness is drawn here in the reference's order (per column: the SampleRandom polynomial, then the
  // Gaussian error); the arithmetic of all columns -- 2 ncol L forward and ncol L inverse row transforms, products, CRT, the shifted
  // key term and the reduction modulo 2^logQ -- is ONE device call (fhesi_keyswitch_init_batch).  InitObjects below is the same
  // computation one DoubleCRT object at a time, as the reference writes it; both give identical matrices (tests/host/test_wire.cpp).
  void Init(const FHESISecKey& src, const FHESISecKey& dst) {
    const std::vector<DoubleCRT>& s = src.GetRepresentation();
    const size_t n = src.GetSize(); const long phim = context.zMstar.phiM(), L = context.numPrimes();
    const long ncol = (long)(context.ndigits * n); const int nl = (int)((context.logQ + 63) / 64);
    std::vector<uint64_t> a((size_t)ncol * phim * nl); std::vector<int64_t> err((size_t)ncol * phim);
    for (long ind = 0; ind < ncol; ++ind) {
      ZZX poly; SampleRandom(poly, context.modulusQ, phim);
      for (long k = 0; k < phim; ++k) coeff(poly, k).to_limbs(&a[((size_t)ind * phim + k) * nl], nl);
      ZZX e; sampleGaussian(e, phim, context.stdev);
      for (long k = 0; k < phim; ++k) err[(size_t)ind * phim + k] = coeff(e, k).to_long();
    }
    fhesi_ksk* k = nullptr;
    ck(fhesi_ksk_create(context.handle(), (int32_t)n, (int32_t)context.ndigits, &k));
    std::vector<const fhesi_dcrt*> hs; for (auto& d : s) hs.push_back(d.handle());
    int rc = fhesi_keyswitch_init_batch(k, hs.data(), (int32_t)n, dst.GetRepresentation()[1].handle(), (int32_t)context.logQ, (int32_t)context.decompSize, a.data(), nl, err.data());
    if (rc) { fhesi_ksk_free(k); ck(rc); }
    const uint64_t* rows = (const uint64_t*)fhesi_ksk_device_ptr(k); const size_t rowWords = (size_t)L * phim;
    keySwitchMatrix.assign(2, std::vector<DoubleCRT>());
    for (int r = 0; r < 2; ++r)
      for (long col = 0; col < ncol; ++col) {
        DoubleCRT d(context);
        ck(fhesi_dev_copy(context.handle(), fhesi_dcrt_device_ptr(d.handle()), rows + ((size_t)r * ncol + col) * rowWords, rowWords * 8));
        keySwitchMatrix[r].push_back(d);
      }
    devKey = std::make_shared<DeviceKey>(k, (int)n, (int)context.ndigits);   // the device object the matrix was generated in serves the fused calls as it is
  }
  // the same matrix with the column randomness drawn on the device (fhesi_keyswitch_init_batch_seeded): column c <-> object index first + c;
  // the public polynomials a from public_seed, the secret errors from seed.  KeySwitchSI(sk, seq) takes its index range from a SeedSequence.
  struct Seeded { uint64_t seed, public_seed, first; };
  KeySwitchSI(const FHESISecKey& s, SeedSequence& seq) : KeySwitchSI(s, Seeded{seq.seed, seq.public_seed, seq.take((uint64_t)(s.GetContext().ndigits * (s.GetRepresentation().size() * 2 - 1)))}) {}
  KeySwitchSI(const FHESISecKey& s, Seeded sd) : context(s.GetContext()) {
    std::vector<DoubleCRT> sKeys = s.GetRepresentation(), tKeys(sKeys.size() * 2 - 1, sKeys[1]);
    tKeys[0] = sKeys[0];
    for (size_t i = 2; i < tKeys.size(); ++i) tKeys[i] *= tKeys[i - 1];
    InitSeeded(tKeys, s, sd);
  }
  void InitSeeded(const std::vector<DoubleCRT>& s, const FHESISecKey& dst, Seeded sd) {
    const size_t n = s.size(); const long phim = context.zMstar.phiM(), L = context.numPrimes(); const long ncol = (long)(context.ndigits * n);
    fhesi_ksk* k = nullptr;
    ck(fhesi_ksk_create(context.handle(), (int32_t)n, (int32_t)context.ndigits, &k));
    std::vector<const fhesi_dcrt*> hs; for (auto& d : s) hs.push_back(d.handle());
    int rc = fhesi_keyswitch_init_batch_seeded(k, hs.data(), (int32_t)n, dst.GetRepresentation()[1].handle(), (int32_t)context.logQ, (int32_t)context.decompSize, sd.seed, sd.public_seed, sd.first);
    if (rc) { fhesi_ksk_free(k); ck(rc); }
    const uint64_t* rows = (const uint64_t*)fhesi_ksk_device_ptr(k); const size_t rowWords = (size_t)L * phim;
    keySwitchMatrix.assign(2, std::vector<DoubleCRT>());
    for (int r = 0; r < 2; ++r) for (long col = 0; col < ncol; ++col) {
      DoubleCRT d(context);
      ck(fhesi_dev_copy(context.handle(), fhesi_dcrt_device_ptr(d.handle()), rows + ((size_t)r * ncol + col) * rowWords, rowWords * 8));
      keySwitchMatrix[r].push_back(d);
    }
    devKey = std::make_shared<DeviceKey>(k, (int)n, (int)context.ndigits);
  }
  void InitObjects(const FHESISecKey& src, const FHESISecKey& dst) {   // the reference's loop, one object at a time
    std::vector<DoubleCRT> s = src.GetRepresentation(); std::vector<ZZX> sCoeff(s.size());
    for (size_t i = 0; i < s.size(); ++i) s[i].toPoly(sCoeff[i]);
    DoubleCRT t = dst.GetRepresentation()[1]; size_t n = src.GetSize();
    std::vector<DoubleCRT> A, b;
    for (size_t i = 0; i < n; ++i)
      for (unsigned j = 0; j < context.ndigits; ++j) {
        ZZX poly; SampleRandom(poly, context.modulusQ, context.zMstar.phiM());
        DoubleCRT a(poly, context), bb = a; a *= -1L; bb *= t;
        ZZX bCoeff; bb.toPoly(bCoeff);
        ZZX err; sampleGaussian(err, context.zMstar.phiM(), context.stdev);
        bCoeff += err; bCoeff += sCoeff[i];
        for (auto& c : sCoeff[i].rep) c <<= (long)(8 * context.decompSize);
        ReduceCoefficients(bCoeff, context.logQ);
        A.push_back(a); b.push_back(DoubleCRT(bCoeff, context));
      }
    drop_device_key();
    keySwitchMatrix.clear(); keySwitchMatrix.push_back(b); keySwitchMatrix.push_back(A);
  }
  void InitS2(const FHESISecKey& s) {   // FHE-SI.cpp:211-227
    std::vector<DoubleCRT> sKeys = s.GetRepresentation(), tKeys(sKeys.size() * 2 - 1, sKeys[1]);
    tKeys[0] = sKeys[0];
    for (size_t i = 2; i < tKeys.size(); ++i) tKeys[i] *= tKeys[i - 1];
    FHESISecKey tensored(s.GetContext()); tensored.UpdateRepresentation(tKeys);
    InitAny(tensored, s);
  }
  KeySwitchSI(const FHESISecKey& s, unsigned k) : context(s.GetContext()) { InitAutomorph(s, k); }     // FHE-SI.h: key for X -> X^k
  void InitAutomorph(const FHESISecKey& s, unsigned k) {   // FHE-SI.cpp:229-239
    std::vector<DoubleCRT> sKeys = s.GetRepresentation();
    FHESISecKey automorphedKey(s.GetContext());           // (its constructor draws a key that is replaced below, as in the reference)
    for (auto& sk : sKeys) sk.automorph((long)k);
    automorphedKey.UpdateRepresentation(sKeys);
    InitAny(automorphedKey, s);
  }
  // ApplyKeySwitch (FHE-SI.cpp:241-260).  The reference's body -- ScaleDown, ByteDecomp, one DoubleCRT per digit polynomial, two DotProducts,
  // toPoly, ReduceCoefficients -- is ApplyKeySwitchObjects below, one object at a time (2 s per call at the metric ring: the digits alone
  // are 66 polynomials through the host).  ApplyKeySwitch itself hands the ciphertext to the fused device call with the matrix resident in
  // HBM as one object (built on first use): the same bits (tests/host/test_wire.cpp compares the two), about 100 times faster.
  void ApplyKeySwitch(Ciphertext& ctxt) const {
    const size_t ncomp = keySwitchMatrix.empty() ? 0 : keySwitchMatrix[0].size() / context.ndigits;
    if (objectAtATime || ncomp < 2 || ctxt.size() != ncomp) { ApplyKeySwitchObjects(ctxt); return; }
    if (LazyCiphertexts()) {
      CtEngine& e = ct_engine(context);
      if (ctxt.scaledUp && !ctxt.terms.empty() && ncomp == 3) {       // a sum of recorded products: multiplied out and key-switched in one call of the next evaluation
        CtRef v = e.ks_sum(std::move(ctxt.terms), device_key_ref());
        ctxt.set_device_value(v);
        return;
      }
      if (!ctxt.scaledUp && ncomp == 2 && ctxt.parts.size() == 2 && ctxt.parts.resident()) {   // after an automorphism (Regression.h:170-172)
        CtRef in = ctxt.parts.value();
        CtRef v = (in->kind == CtValue::AUTO && in->pending()) ? e.auto_ks(in->a, in->s, device_key_ref()) : e.auto_ks(in, 1, device_key_ref());
        ctxt.set_device_value(v);
        return;
      }
    }
    ctxt.materialise();
    fhesi_ctx* h = context.handle(); fhesi_ksk* k = device_key();
    const long n = context.zMstar.phiM(), L = context.numPrimes(); const int nl = (int)((context.logQ + 63) / 64);
    void* out; ck(fhesi_dev_alloc(h, (size_t)2 * n * nl * 8, &out));
    if (ctxt.scaledUp) {
      void* rows; ck(fhesi_dev_alloc(h, ncomp * L * n * 8, &rows));
      for (size_t i = 0; i < ncomp; ++i) ck(fhesi_dev_copy(h, (uint64_t*)rows + i * L * n, fhesi_dcrt_device_ptr(ctxt.tProd[i].handle()), (size_t)L * n * 8));
      int rc = fhesi_apply_key_switch_dev(h, k, (int32_t)context.logQ, (int32_t)context.decompSize, (const uint64_t*)rows, 1, (uint64_t*)out, nl);
      fhesi_dev_free(h, rows);
      if (rc) { fhesi_dev_free(h, out); ck(rc); }
    } else {
      // an unscaled ciphertext (after an automorphism): ScaleDown returns at once (Ciphertext.cpp:195), ByteDecomp takes the positive residues
      std::vector<uint64_t> host(ncomp * n * nl, 0);
      for (size_t i = 0; i < ncomp; ++i) poly_to_limbs(ctxt.parts[i].poly, &host[(i * n) * nl], n, nl);
      void* in; ck(fhesi_dev_alloc(h, host.size() * 8, &in)); ck(fhesi_dev_upload(h, in, host.data(), host.size() * 8));
      int rc = fhesi_ct_automorph_key_switch_dev(h, k, (int32_t)context.logQ, (int32_t)context.decompSize, 1, (const uint64_t*)in, nl, 1, (uint64_t*)out, nl);
      fhesi_dev_free(h, in);
      if (rc) { fhesi_dev_free(h, out); ck(rc); }
    }
    std::vector<uint64_t> res((size_t)2 * n * nl);
    ck(fhesi_dev_download(h, res.data(), out, res.size() * 8)); ck(fhesi_dev_free(h, out));
    ctxt.tProd.clear(); ctxt.scaledUp = false; ctxt.parts.assign(2, CiphertextPart(context));
    for (int r = 0; r < 2; ++r) limbs_to_poly(ctxt.parts[r].poly, &res[(size_t)r * n * nl], n, nl);
  }
  void ApplyKeySwitchObjects(Ciphertext& ctxt) const {   // the reference's loop, one object at a time
    ctxt.ScaleDown(); ctxt.ByteDecomp();
    std::vector<DoubleCRT> bd; for (auto& p : ctxt.parts) bd.push_back(DoubleCRT(p.poly, context));
    std::vector<CiphertextPart> newCtxt(keySwitchMatrix.size(), CiphertextPart(context));
    for (size_t i = 0; i < keySwitchMatrix.size(); ++i) { DoubleCRT dp(context); DotProduct(dp, keySwitchMatrix[i], bd); dp.toPoly(newCtxt[i].poly); ReduceCoefficients(newCtxt[i].poly, context.logQ); }
    ctxt.parts = newCtxt;
  }
  // a[i] *= b[i]; ApplyKeySwitch(a[i]) for every i in ONE device call (fhesi_ct_mul_relin_batch): what a loop over a Matrix<Ciphertext> row or a
  // vector of ciphertexts should call instead of the two statements per object -- the objects cross the host boundary once per batch
  void MulRelinBatch(std::vector<Ciphertext>& a, const std::vector<Ciphertext>& b) const {
    if (a.size() != b.size()) Error("MulRelinBatch: the operand vectors differ in length");
    const size_t count = a.size(); if (!count) return;
    if (LazyCiphertexts() && !objectAtATime) {    // recorded: the two statements per object become one wave at the next evaluation, operands and results in HBM
      for (size_t c = 0; c < count; ++c) {
        if (a[c].isScaledUp() || b[c].isScaledUp() || a[c].size() != 2 || b[c].size() != 2) Error("MulRelinBatch: operands must be unscaled two-part ciphertexts");
        a[c] *= b[c]; ApplyKeySwitch(a[c]);
      }
      return;
    }
    const long n = context.zMstar.phiM(); const int nl = (int)((context.logQ + 63) / 64);
    std::vector<uint64_t> ha(count * 2 * n * nl, 0), hb(ha.size(), 0), ho(ha.size());
    for (size_t c = 0; c < count; ++c) {
      if (a[c].isScaledUp() || b[c].isScaledUp() || a[c].size() != 2 || b[c].size() != 2) Error("MulRelinBatch: operands must be unscaled two-part ciphertexts");
      for (int part = 0; part < 2; ++part) { poly_to_limbs(a[c].parts[part].poly, &ha[((c * 2 + part) * n) * nl], n, nl); poly_to_limbs(b[c].parts[part].poly, &hb[((c * 2 + part) * n) * nl], n, nl); }
    }
    ck(fhesi_ct_mul_relin_batch(context.handle(), device_key(), (int32_t)context.logQ, (uint64_t)context.ModulusP().to_long(), (int32_t)context.decompSize, ha.data(), hb.data(), ho.data(), nl, (int64_t)count));
    for (size_t c = 0; c < count; ++c) for (int part = 0; part < 2; ++part) limbs_to_poly(a[c].parts[part].poly, &ho[((c * 2 + part) * n) * nl], n, nl);
  }
  KeySwitchSI(const KeySwitchSI& o) : context(o.context), keySwitchMatrix(o.keySwitchMatrix), objectAtATime(o.objectAtATime), devKey(o.devKey) {}      // (the device object is immutable once built: shared)
  KeySwitchSI& operator=(const KeySwitchSI& o) { if (&context != &o.context) Error("Incompatible contexts."); keySwitchMatrix = o.keySwitchMatrix; objectAtATime = o.objectAtATime; devKey = o.devKey; return *this; }
 private:
  // keySwitchMatrix as one HBM-resident fhesi_ksk for the fused calls; shared with the recorded operations that will use it (fhesi_engine.h),
  // so a matrix that is replaced or destroyed before they run stays alive until they have
  mutable DeviceKeyRef devKey;
  void drop_device_key() const { devKey.reset(); }
  const DeviceKeyRef& device_key_ref() const {
    if (devKey) return devKey;
    const size_t ncol = keySwitchMatrix[0].size(), ncomp = ncol / context.ndigits; const size_t rowWords = (size_t)context.numPrimes() * context.zMstar.phiM();
    fhesi_ksk* k = nullptr;
    ck(fhesi_ksk_create(context.handle(), (int32_t)ncomp, (int32_t)context.ndigits, &k));
    devKey = std::make_shared<DeviceKey>(k, (int)ncomp, (int)context.ndigits);
    uint64_t* rows = (uint64_t*)fhesi_ksk_device_ptr(k);
    for (int r = 0; r < 2; ++r) for (size_t col = 0; col < ncol; ++col)
      ck(fhesi_dev_copy(context.handle(), rows + ((size_t)r * ncol + col) * rowWords, fhesi_dcrt_device_ptr(keySwitchMatrix[r][col].handle()), rowWords * 8));
    ck(fhesi_ksk_mark_dirty(k));
    return devKey;
  }
  fhesi_ksk* device_key() const { return device_key_ref()->k; }
 public:
  // the matrix as ONE device object, built on first use and shared (the wave executors of fhesi_matrix.h use it instead of a copy of their own)
  const DeviceKeyRef& DeviceMatrix() const { return device_key_ref(); }
};

}  // namespace fhesi
