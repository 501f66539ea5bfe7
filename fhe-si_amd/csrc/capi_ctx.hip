// capi_ctx.hip -- context, options, per-kernel stopwatch, plain device memory (include/fhesi_hip.h)
#include "capi_common.h"

// --------------------------------------------------------------------------------------------- helpers
int ws_reserve(fhesi_ctx* ctx, int slot, size_t bytes, void** out) {
  if (ctx->ws_bytes[slot] < bytes) {
    if (ctx->ws[slot]) { HIP_TRY(hipStreamSynchronize(ctx->stream)); HIP_TRY(hipFree(ctx->ws[slot])); ctx->ws[slot] = nullptr; ctx->ws_bytes[slot] = 0; }
    size_t want = bytes + (bytes >> 3) + 4096;
    if (hipMalloc(&ctx->ws[slot], want) != hipSuccess) {
      (void)hipGetLastError();
      ctx->ws[slot] = nullptr;
      if (hipMalloc(&ctx->ws[slot], bytes) != hipSuccess) {      // (without the growth margin)
        (void)hipGetLastError();
        ctx->ws[slot] = nullptr;
        ctx->ws_oom = true;                                       // callers that can work in smaller chunks look at this
        FHESI_FAIL("workspace slot %d: hipMalloc of %zu bytes failed", slot, bytes);
      }
      want = bytes;
    }
    ctx->ws_bytes[slot] = want;
  }
  *out = ctx->ws[slot];
  // FHESI_WS_POISON=1 (a test hook, read at context creation): every reservation is filled with 0xA5 bytes first, so that a kernel relying on
  // what an earlier call happened to leave in a workspace -- zeros, mostly -- fails the parity tests instead of passing by luck
  if (ctx->ws_poison && bytes) HIP_TRY(hipMemsetAsync(ctx->ws[slot], 0xA5, bytes, ctx->stream));
  return 0;
}

// device copy of an index list (small, cached per call in workspace slot 3)
int upload_idx(fhesi_ctx* ctx, const std::vector<int>& idx, int** d_out) {
  bool identity = (int)idx.size() == ctx->L;
  for (size_t i = 0; identity && i < idx.size(); ++i) identity = idx[i] == (int)i;
  if (identity) { *d_out = nullptr; return 0; }
  void* p;
  FHESI_TRY(ws_reserve(ctx, 3, idx.size() * sizeof(int) + 64, &p));
  HIP_TRY(hipMemcpyAsync(p, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  *d_out = (int*)p;
  return 0;
}

// --------------------------------------------------------------------------------------------- context
extern "C" int32_t fhesi_abi_version(void) { return FHESI_ABI_VERSION; }
extern "C" int fhesi_device_count(int32_t* count) {
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) { *count = 0; FHESI_FAIL("hipGetDeviceCount failed: %s", hipGetErrorString(e)); }
  *count = c;
  return 0;
}

static void build_tile_order(const std::vector<Shoup2>& tw, int logn, std::vector<Shoup2>& twt) {
  const i64 n = 1ll << logn;
  const int R = logn - 10;
  twt.assign(n, Shoup2{0, 0});
  for (int i = 0; i < 32; ++i) twt[i] = tw[i];
  for (int u = 0; u < 5; ++u)
    for (int x = 0; x < (1 << u); ++x)
      for (int p1 = 0; p1 < 32; ++p1) twt[32 + ((1 << u) - 1 + x) * 32 + p1] = tw[(1 << (5 + u)) + (p1 << u) + x];
  for (int u = 0; u < R; ++u)
    for (int x = 0; x < (1 << u); ++x)
      for (int jl = 0; jl < 1024; ++jl) twt[1024 + (i64)((1 << u) - 1 + x) * 1024 + jl] = tw[(1 << (10 + u)) + ((i64)hm::brv(jl, 10) << u) + x];
}
// the tile kernels use quotients scaled by 2^63 (modarith63.h), relative to the modulus they compute with (q_tile)
static void to_q63(std::vector<Shoup2>& t, u64 q) {
  for (auto& e : t) e.wp = hm::shoup63(e.w, q);
}

thread_local bool g_fhesi_internal_ctx = false;

struct OptDesc { const char* name; const char* env; size_t off; bool wide; };
static const OptDesc kOptions[] = {
  {"ks_direct", "FHESI_KS_DIRECT", offsetof(CtxOptions, ks_direct), false},
  {"ks_residues", "FHESI_KS_RESIDUES", offsetof(CtxOptions, ks_residues), false},
  {"ks_aux60", "FHESI_KS_AUX60", offsetof(CtxOptions, ks_aux60), false},
  {"crt_exact", "FHESI_CRT_EXACT", offsetof(CtxOptions, crt_exact), false},
  {"crt_skip_cleanup", "FHESI_CRT_SKIP_CLEANUP", offsetof(CtxOptions, crt_skip_cleanup), false},
  {"lanes", "FHESI_LANES", offsetof(CtxOptions, lanes), false},
  {"stagger", "FHESI_STAGGER", offsetof(CtxOptions, stagger), false},
  {"batch_chunk", "FHESI_BATCH_CHUNK", offsetof(CtxOptions, batch_chunk), true},
  {"wave_operands", "FHESI_WAVE_OPERANDS", offsetof(CtxOptions, wave_operands), true},
  {"wave_single", "FHESI_WAVE_SINGLE", offsetof(CtxOptions, wave_single), false},
  {"tensor32", "FHESI_TENSOR32", offsetof(CtxOptions, tensor32), false},
  {"tensor_bits", "FHESI_TENSOR_BITS", offsetof(CtxOptions, tensor_bits), false},
  {"dot32_k4", "FHESI_DOT32_K4", offsetof(CtxOptions, dot32_k4), false},
  {"parts_words", "FHESI_PARTS_WORDS", offsetof(CtxOptions, parts_words), false},
  {"automorph_rows", "FHESI_AUTOMORPH_ROWS", offsetof(CtxOptions, automorph_rows), false},
  {"ks_long_keys", "FHESI_KS_LONG_KEYS", offsetof(CtxOptions, ks_long_keys), false},
  {"host_chunk", "FHESI_HOST_CHUNK", offsetof(CtxOptions, host_chunk), true},
  {"host_threads", "FHESI_HOST_THREADS", offsetof(CtxOptions, host_threads), false},
};
static void opt_store(CtxOptions* o, const OptDesc& d, long long v) {
  if (d.wide) *(long long*)((char*)o + d.off) = v; else *(int*)((char*)o + d.off) = (int)v;
}
static long long opt_load(const CtxOptions* o, const OptDesc& d) {
  return d.wide ? *(const long long*)((const char*)o + d.off) : (long long)*(const int*)((const char*)o + d.off);
}
extern "C" int fhesi_ctx_set_option(fhesi_ctx* c, const char* name, int64_t value) {
  if (!c || !name) FHESI_FAIL("set_option: null argument");
  for (const OptDesc& d : kOptions)
    if (!strcmp(d.name, name)) { opt_store(&c->opt, d, value); return 0; }
  FHESI_FAIL("set_option: unknown option '%s'", name);
}
extern "C" int fhesi_ctx_get_option(const fhesi_ctx* c, const char* name, int64_t* value) {
  if (!c || !name || !value) FHESI_FAIL("get_option: null argument");
  for (const OptDesc& d : kOptions)
    if (!strcmp(d.name, name)) { *value = opt_load(&c->opt, d); return 0; }
  FHESI_FAIL("get_option: unknown option '%s'", name);
}
extern "C" int fhesi_ctx_copy_options(fhesi_ctx* dst, const fhesi_ctx* src) {
  if (!dst || !src) FHESI_FAIL("copy_options: null argument");
  dst->opt = src->opt;
  return 0;
}   // set by bluestein_init: its convolution context needs sizes up to 4m

extern "C" int fhesi_ctx_create(fhesi_ctx** out, int64_t m, int32_t nprimes, const uint64_t* q, const uint64_t* root, int32_t device) {
  if (!out) FHESI_FAIL("null output pointer");
  *out = nullptr;
  if (m < 2 || m > (g_fhesi_internal_ctx ? (1 << 23) : (1 << 20))) FHESI_FAIL("FHEcontext: m undefined or larger than 2^20");     // FHEContext.cpp:89
  if (nprimes < 1 || nprimes > 64) FHESI_FAIL("FHEcontext: number of primes %d outside [1,64]", nprimes);
  for (int i = 0; i < nprimes; ++i) {
    // FHEContext.cpp:31-34: assert( ProbPrime(p) && p % twoM == 1 && !inChain(p) )
    // single-precision moduli of the reference are below 2^NTL_SP_NBITS (60 in NTL >= 10: zz_p::init rejects larger ones), and
    // every lazy range of the kernels (4q + 2^32 < 2^63, 8 products of < 2^124 in a 128-bit sum, ...) is sized for that
    if (q[i] >= (1ull << 60)) FHESI_FAIL("AddPrime: prime %d does not fit 60 bits (NTL_SP_NBITS)", i);
    if (!hm::is_prime(q[i])) FHESI_FAIL("AddPrime: modulus %d (%llu) is not prime", i, (unsigned long long)q[i]);
    if (q[i] % (2 * (u64)m) != 1) FHESI_FAIL("AddPrime: prime %d (%llu) is not 1 mod 2m", i, (unsigned long long)q[i]);
    for (int j = 0; j < i; ++j)
      if (q[j] == q[i]) FHESI_FAIL("AddPrime: prime %d (%llu) already in chain", i, (unsigned long long)q[i]);
    if (!hm::is_primitive_2m_root(root[i], m, q[i])) FHESI_FAIL("Cmodulus: root %d is not a primitive 2m-th root of unity mod q", i);
  }
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) FHESI_FAIL("device %d not available (%d visible)", device, ndev);
  HIP_TRY(hipSetDevice(device));

  fhesi_ctx* c = new fhesi_ctx();
  for (const OptDesc& d : kOptions)          // initial values from the environment, read once per context
    if (const char* e = getenv(d.env)) opt_store(&c->opt, d, atoll(e));
  c->device = device;
  c->m = m;
  c->L = nprimes;
  c->q.assign(q, q + nprimes);
  c->root.assign(root, root + nprimes);
  c->zms_idx = hm::zms_idx(m, &c->phim);
  c->phi = hm::cyclotomic(m);
  c->pow2 = (m & (m - 1)) == 0 && m >= 4;
  c->logn = c->pow2 ? hm::ilog2_ceil(c->phim) : 0;
  if (!c->pow2 && 2 * c->phim - 1 <= 64 * kAux32N) {
    // rings whose products run as linear convolutions on padded power-of-two rows of 2^14 .. 2^20: every m = p - 1 (safe prime p) and every
    // prime m the reference admits (m < 2^20, FHEContext.cpp:89; its drivers: Test_AddMul.cpp:131, Test_Regression.cpp:122).  Rows up to 2^16
    // run the fused loaders, longer ones the simple path (head / tail stages as passes of their own, ntt32_core.inc)
    if (m % 2 == 0 && (m / 2) % 2 == 1 && hm::is_prime((u64)(m / 2))) c->lin_q = m / 2;
    else if (m % 2 == 1 && m > 2 && hm::is_prime((u64)m)) { c->lin_q = m; c->lin_prime = true; }
    if (c->lin_q) {
      c->lin_lg = 14;
      while (((i64)1 << c->lin_lg) < 2 * c->phim - 1) ++c->lin_lg;
      // FHESI_LIN_LG: LONGER padded rows than the ring needs (a test hook: the long-row paths on rings small enough for the oracle)
      if (const char* e = getenv("FHESI_LIN_LG")) { const int want = atoi(e); if (want > c->lin_lg && want <= 20) c->lin_lg = want; }
    }
  }
  if (const char* e = getenv("FHESI_WS_POISON")) c->ws_poison = atoi(e) != 0;
  HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIP_TRY(hipStreamCreateWithFlags(&c->lane_stream, hipStreamNonBlocking));
  HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&c->ev_mid, hipEventDisableTiming));
  HIP_TRY(hipEventCreate(&c->ev0));
  HIP_TRY(hipEventCreate(&c->ev1));

  const i64 n = c->phim;
  c->pc.resize(nprimes);
  for (int i = 0; i < nprimes; ++i) {
    PrimeConst& pc = c->pc[i];
    const u64 Q = q[i];
    pc.q = Q;
    pc.two_q = 2 * Q;
    int k = 64 - __builtin_clzll(Q);
    pc.bar_k = (u32)k;
    pc.bar_mu = (u64)((((u128)1) << (2 * k)) / Q);
    pc.r64 = (u64)(((u128)1 << 64) % Q);
    pc.r64_sh = hm::shoup(pc.r64, Q);
    pc.one_sh = hm::shoup(1, Q);
    // modulus of the tile kernels: q itself, or for small primes the largest multiple of q below 2^60 (ntt_tile.inc)
    pc.q_tile = Q >= (1ull << 48) ? Q : Q * (((1ull << 60) - 1) / Q);
    pc.one_q63 = hm::shoup63(1, pc.q_tile);
    {
      const u64 qh = pc.q_tile >> 32;                  // >= 2^16
      const int b = 64 - __builtin_clzll(qh);
      pc.norm_m = (u32)((((u128)1) << (31 + b)) / (qh + 1));   // in [2^31, 2^32): qh + 1 lies in (2^(b-1), 2^b]
    }
    if (Q < (1ull << 48)) c->has_small_prime = true; else c->n_big_primes++;
    pc.ninv = pc.ninv_sh = pc.ninv_w = pc.ninv_w_sh = 0;
  }
  if (c->pow2) {
    const int lg = c->logn;
    std::vector<Shoup2> twf((size_t)nprimes * n), twi((size_t)nprimes * n);
    for (int i = 0; i < nprimes; ++i) {
      const u64 Q = q[i];
      const u64 psi = hm::mulmod(root[i], root[i], Q), ipsi = hm::invmod(psi, Q);
      // powers in natural order, then scatter to bit-reversed slots
      std::vector<u64> pw(n), ipw(n);
      pw[0] = ipw[0] = 1;
      for (i64 e = 1; e < n; ++e) { pw[e] = hm::mulmod(pw[e - 1], psi, Q); ipw[e] = hm::mulmod(ipw[e - 1], ipsi, Q); }
      for (i64 j = 0; j < n; ++j) {
        const u64 e = hm::brv((u64)j, lg);
        twf[(size_t)i * n + j] = {pw[e], hm::shoup(pw[e], Q)};
        twi[(size_t)i * n + j] = {ipw[e], hm::shoup(ipw[e], Q)};
      }
      PrimeConst& pc = c->pc[i];
      pc.ninv = hm::invmod((u64)n % Q, Q);
      pc.ninv_sh = hm::shoup(pc.ninv, Q);
      pc.ninv_w = hm::mulmod(pc.ninv, n > 1 ? twi[(size_t)i * n + 1].w : 1, Q);
      pc.ninv_w_sh = hm::shoup(pc.ninv_w, Q);
      pc.ninv_q63 = hm::shoup63(pc.ninv, pc.q_tile);
      pc.ninv_w_q63 = hm::shoup63(pc.ninv_w, pc.q_tile);
    }
    const size_t tb = twf.size() * sizeof(Shoup2);
    HIP_TRY(hipMalloc(&c->d_tw_fwd, tb));
    HIP_TRY(hipMalloc(&c->d_tw_inv, tb));
    HIP_TRY(hipMemcpy(c->d_tw_fwd, twf.data(), tb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_tw_inv, twi.data(), tb, hipMemcpyHostToDevice));
    if (lg >= 11 && lg <= 14) {
      std::vector<Shoup2> all_f((size_t)nprimes * n), all_i((size_t)nprimes * n), one, tmp;
      for (int i = 0; i < nprimes; ++i) {
        one.assign(twf.begin() + (size_t)i * n, twf.begin() + (size_t)(i + 1) * n);
        build_tile_order(one, lg, tmp);
        to_q63(tmp, c->pc[i].q_tile);
        std::copy(tmp.begin(), tmp.end(), all_f.begin() + (size_t)i * n);
        one.assign(twi.begin() + (size_t)i * n, twi.begin() + (size_t)(i + 1) * n);
        build_tile_order(one, lg, tmp);
        to_q63(tmp, c->pc[i].q_tile);
        std::copy(tmp.begin(), tmp.end(), all_i.begin() + (size_t)i * n);
      }
      HIP_TRY(hipMalloc(&c->d_twt_fwd, tb));
      HIP_TRY(hipMalloc(&c->d_twt_inv, tb));
      HIP_TRY(hipMemcpy(c->d_twt_fwd, all_f.data(), tb, hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(c->d_twt_inv, all_i.data(), tb, hipMemcpyHostToDevice));
    }
    if (lg >= 15 && lg <= 17) {      // two-pass transforms (TileBig in ntt_tile.inc, ntt_*_tail in kernels_ntt.hip)
      const int s0 = lg - 14, nsub = 1 << s0;
      const i64 n1 = 1 << 14;
      std::vector<Shoup2> all_f((size_t)nprimes * n1), all_i((size_t)nprimes * n), all_fs((size_t)nprimes * n), tail((size_t)nprimes * n, Shoup2{0, 0}),
          fold((size_t)nprimes * nsub), one(n1), tmp;
      for (int i = 0; i < nprimes; ++i) {
        const u64 Q = q[i];
        // forward sub-transforms: ring of 2^14 points with root psi^nsub = the first 2^14 entries of the bit-reversed table
        one.assign(twf.begin() + (size_t)i * n, twf.begin() + (size_t)i * n + n1);
        build_tile_order(one, 14, tmp);
        to_q63(tmp, c->pc[i].q_tile);
        std::copy(tmp.begin(), tmp.end(), all_f.begin() + (size_t)i * n1);
        // inverse sub-transform `sub`: stage s0+u of the row, block (sub << u) + b
        for (int sub = 0; sub < nsub; ++sub) {
          one[0] = Shoup2{0, 0};
          for (int u = 0; u < 14; ++u)
            for (i64 b = 0; b < (1ll << u); ++b) one[(1ll << u) + b] = twi[(size_t)i * n + (1ll << (s0 + u)) + ((i64)sub << u) + b];
          build_tile_order(one, 14, tmp);
          to_q63(tmp, c->pc[i].q_tile);
          std::copy(tmp.begin(), tmp.end(), all_i.begin() + ((size_t)i * nsub + sub) * n1);
          // forward slice of the same sub-block (order-free sub-transforms)
          for (int u = 0; u < 14; ++u)
            for (i64 b = 0; b < (1ll << u); ++b) one[(1ll << u) + b] = twf[(size_t)i * n + (1ll << (s0 + u)) + ((i64)sub << u) + b];
          build_tile_order(one, 14, tmp);
          to_q63(tmp, c->pc[i].q_tile);
          std::copy(tmp.begin(), tmp.end(), all_fs.begin() + ((size_t)i * nsub + sub) * n1);
          const u64 f = hm::mulmod(c->pc[i].ninv, twi[(size_t)i * n + nsub + sub].w, Q);
          fold[(size_t)i * nsub + sub] = Shoup2{f, hm::shoup63(f, c->pc[i].q_tile)};
        }
        // forward tail: stage s builds M = 2^(14+s) points, entry j = psi^((n/M)(2j+1)), j < M/2
        const u64 psi = hm::mulmod(root[i], root[i], Q);
        for (int s = 1; s <= s0; ++s) {
          const i64 half = n1 << (s - 1);
          const u64 base = hm::powmod(psi, (u64)(n / (2 * half)), Q), step = hm::mulmod(base, base, Q);
          u64 w = base;
          Shoup2* dst = tail.data() + (size_t)i * n + n1 * ((1ll << (s - 1)) - 1);
          for (i64 j = 0; j < half; ++j) { dst[j] = Shoup2{w, hm::shoup(w, Q)}; w = hm::mulmod(w, step, Q); }
        }
      }
      HIP_TRY(hipMalloc(&c->d_twt_fwd, all_f.size() * sizeof(Shoup2)));
      HIP_TRY(hipMalloc(&c->d_twt_inv, all_i.size() * sizeof(Shoup2)));
      {
        std::vector<Shoup2> head(nprimes);
        for (int i = 0; i < nprimes; ++i) { const u64 w = twf[(size_t)i * n + 1].w; head[i] = Shoup2{w, hm::shoup63(w, c->pc[i].q_tile)}; }
        HIP_TRY(hipMalloc(&c->d_head_tw, head.size() * sizeof(Shoup2)));
        HIP_TRY(hipMemcpy(c->d_head_tw, head.data(), head.size() * sizeof(Shoup2), hipMemcpyHostToDevice));
      }
      HIP_TRY(hipMalloc(&c->d_twt_fwd_sub, all_fs.size() * sizeof(Shoup2)));
      HIP_TRY(hipMemcpy(c->d_twt_fwd_sub, all_fs.data(), all_fs.size() * sizeof(Shoup2), hipMemcpyHostToDevice));
      HIP_TRY(hipMalloc(&c->d_tail_fwd, tail.size() * sizeof(Shoup2)));
      HIP_TRY(hipMalloc(&c->d_sub_fold, fold.size() * sizeof(Shoup2)));
      HIP_TRY(hipMemcpy(c->d_twt_fwd, all_f.data(), all_f.size() * sizeof(Shoup2), hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(c->d_twt_inv, all_i.data(), all_i.size() * sizeof(Shoup2), hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(c->d_tail_fwd, tail.data(), tail.size() * sizeof(Shoup2), hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(c->d_sub_fold, fold.data(), fold.size() * sizeof(Shoup2), hipMemcpyHostToDevice));
    }
  }
  HIP_TRY(hipMalloc(&c->d_pc, sizeof(PrimeConst) * nprimes));
  HIP_TRY(hipMemcpy(c->d_pc, c->pc.data(), sizeof(PrimeConst) * nprimes, hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&c->d_zms_idx, sizeof(int) * m));
  HIP_TRY(hipMemcpy(c->d_zms_idx, c->zms_idx.data(), sizeof(int) * m, hipMemcpyHostToDevice));
  std::vector<int> zl(n);
  for (i64 i = 0; i < m; ++i)
    if (c->zms_idx[i] >= 0) zl[c->zms_idx[i]] = (int)i;
  HIP_TRY(hipMalloc(&c->d_zms_list, sizeof(int) * n));
  HIP_TRY(hipMemcpy(c->d_zms_list, zl.data(), sizeof(int) * n, hipMemcpyHostToDevice));
  if (!c->pow2) {
    int r = bluestein_init(c);
    if (r) { fhesi_ctx_destroy(c); return r; }
  }
  *out = c;
  return 0;
}

extern "C" int fhesi_ctx_destroy(fhesi_ctx* c) {
  if (!c) return 0;
  // DoubleCRT objects and key-switch matrices hold a pointer to their context (the reference's `const FHEcontext&`, DoubleCRT.h:84):
  // the context must outlive them, so destroying it while handles are alive is refused instead of leaving them dangling
  if (c->live_handles > 0) FHESI_FAIL("fhesi_ctx_destroy: %d DoubleCRT / key-switch handle(s) of this context are still alive", c->live_handles);
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  for (auto& r : c->prof) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
  c->prof.clear();
  bluestein_destroy(c);
  for (auto& kv : c->crt_cache) { hipFree(kv.second->d_blob); if (kv.second->d_flags) hipFree(kv.second->d_flags); delete kv.second; }
  for (auto& kv : c->pow64_cache) hipFree(kv.second);
  for (auto& kv : c->scalar_cache) hipFree(kv.second);
  aux32_free(c);
  tensor32_free(c);
  for (int i = 0; i < FHESI_WS_SLOTS; ++i) if (c->lane_ws[i]) hipFree(c->lane_ws[i]);
  host_stage_free(c);
  if (c->lane_stream) hipStreamDestroy(c->lane_stream);
  if (c->ev_fork) hipEventDestroy(c->ev_fork);
  if (c->ev_join) hipEventDestroy(c->ev_join);
  if (c->ev_mid) hipEventDestroy(c->ev_mid);
  for (int i = 0; i < FHESI_WS_SLOTS; ++i) if (c->ws[i]) hipFree(c->ws[i]);
  hipFree(c->d_pc); hipFree(c->d_tw_fwd); hipFree(c->d_tw_inv); hipFree(c->d_twt_fwd); hipFree(c->d_twt_inv); hipFree(c->d_tail_fwd); hipFree(c->d_sub_fold); hipFree(c->d_twt_fwd_sub); hipFree(c->d_head_tw);
  hipFree(c->d_zms_idx); hipFree(c->d_zms_list);
  if (c->ev0) hipEventDestroy(c->ev0);
  if (c->ev1) hipEventDestroy(c->ev1);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
  return 0;
}

extern "C" int64_t fhesi_ctx_m(const fhesi_ctx* c) { return c ? c->m : 0; }
extern "C" int64_t fhesi_ctx_phim(const fhesi_ctx* c) { return c ? c->phim : 0; }
extern "C" int32_t fhesi_ctx_nprimes(const fhesi_ctx* c) { return c ? c->L : 0; }
extern "C" int fhesi_ctx_prime(const fhesi_ctx* c, int32_t i, uint64_t* q, uint64_t* root) {
  if (!c || i < 0 || i >= c->L) FHESI_FAIL("ithPrime: index %d out of range", i);
  if (q) *q = c->q[i];
  if (root) *root = c->root[i];
  return 0;
}
extern "C" int fhesi_ctx_zms_idx(const fhesi_ctx* c, int32_t* out_m) {
  if (!c) FHESI_FAIL("null context");
  for (i64 i = 0; i < c->m; ++i) out_m[i] = c->zms_idx[i];
  return 0;
}
extern "C" int fhesi_ctx_phi_m(const fhesi_ctx* c, int64_t* o) {
  if (!c) FHESI_FAIL("null context");
  for (size_t i = 0; i < c->phi.size(); ++i) o[i] = c->phi[i];
  return 0;
}
extern "C" int fhesi_ctx_sync(fhesi_ctx* c) { CHECK_CTX(c); HIP_TRY(hipStreamSynchronize(c->stream)); return 0; }
extern "C" void* fhesi_ctx_stream(fhesi_ctx* c) { return c ? (void*)c->stream : nullptr; }
extern "C" int fhesi_timer_start(fhesi_ctx* c) { CHECK_CTX(c); HIP_TRY(hipEventRecord(c->ev0, c->stream)); return 0; }
extern "C" int fhesi_timer_stop(fhesi_ctx* c, float* ms) {
  CHECK_CTX(c);
  HIP_TRY(hipEventRecord(c->ev1, c->stream));
  HIP_TRY(hipEventSynchronize(c->ev1));
  HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
  return 0;
}

// --------------------------------------------------------------------------------------------- per-kernel stopwatch
extern "C" int fhesi_prof_enable(fhesi_ctx* c, int32_t on) {
  CHECK_CTX(c);
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (auto& r : c->prof) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
  c->prof.clear();
  if (on) for (auto& f : c->prof_fn) f = nullptr;
  c->prof_on = on != 0;
  return 0;
}
extern "C" int fhesi_prof_read(fhesi_ctx* c, int32_t cls, int64_t* launches, double* units, double* total_ms) {
  CHECK_CTX(c);
  if (cls < 0 || cls >= PROF_NCLASS) FHESI_FAIL("unknown kernel class %d", cls);
  HIP_TRY(hipStreamSynchronize(c->stream));
  int64_t n = 0; double u = 0, ms = 0;
  for (auto& r : c->prof) {
    if (r.cls != cls) continue;
    float t = 0;
    HIP_TRY(hipEventElapsedTime(&t, r.e0, r.e1));
    ++n; u += r.units; ms += t;
  }
  *launches = n; *units = u; *total_ms = ms;
  return 0;
}

extern "C" int fhesi_prof_kernel_name(fhesi_ctx* c, int32_t cls, char* out, size_t cap) {
  CHECK_CTX(c);
  if (cls < 0 || cls >= PROF_NCLASS || !out || !cap) FHESI_FAIL("prof_kernel_name: bad argument");
  out[0] = 0;
  if (!c->prof_fn[cls]) return 0;
  const char* mangled = hipKernelNameRefByPtr(c->prof_fn[cls], c->stream);
  if (!mangled) return 0;
  int st = 0;
  char* dm = abi::__cxa_demangle(mangled, nullptr, nullptr, &st);
  std::string name = (st == 0 && dm) ? dm : mangled;
  free(dm);
  // rocprofv3 prints "void kernel<args>(params)": keep "kernel<args>"
  if (name.compare(0, 5, "void ") == 0) name.erase(0, 5);
  int depth = 0; size_t cut = name.size();
  for (size_t i = 0; i < name.size(); ++i) { const char ch = name[i]; if (ch == '<') ++depth; else if (ch == '>') --depth; else if (ch == '(' && depth == 0) { cut = i; break; } }
  name.resize(cut);
  snprintf(out, cap, "%s", name.c_str());
  return 0;
}

// --------------------------------------------------------------------------------------------- plain device memory
extern "C" int fhesi_dev_copy(fhesi_ctx* c, void* dst_dev, const void* src_dev, size_t bytes) {
  CHECK_CTX(c);
  HIP_TRY(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}
extern "C" int fhesi_dev_alloc(fhesi_ctx* c, size_t bytes, void** out) { CHECK_CTX(c); HIP_TRY(hipMalloc(out, bytes ? bytes : 8)); return 0; }
extern "C" int fhesi_dev_free(fhesi_ctx* c, void* p) { CHECK_CTX(c); HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(p)); return 0; }
extern "C" int fhesi_dev_upload(fhesi_ctx* c, void* dst, const void* src, size_t bytes) {
  CHECK_CTX(c);
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}
extern "C" int fhesi_dev_download(fhesi_ctx* c, void* dst, const void* src, size_t bytes) {
  CHECK_CTX(c);
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

