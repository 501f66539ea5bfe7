// capi.hip -- implementation of include/fhesi_hip.h on top of the HIP kernels in this directory.
// Product path only: nothing here touches oracle/ and there is no CPU fallback -- every entry point that computes
// fails with an error string if HIP is unavailable.
#include "../../include/fhesi_hip.h"
#include "fhesi_internal.h"
#include <map>
#include <set>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <cxxabi.h>
#include <string>

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
  return 0;
}

#define CHECK_CTX(c) do { if (!(c)) FHESI_FAIL("null context"); HIP_TRY(hipSetDevice((c)->device)); } while (0)

static int row_fwd(fhesi_ctx* ctx, u64* d_rows, i64 count, int nslots, const int* d_pos, const int* h_pos) {
  if (ctx->pow2) return launch_ntt_fwd(ctx, d_rows, count, nslots, d_pos, true);
  return launch_bluestein_fwd(ctx, d_rows, count, nslots, h_pos);
}
static int row_inv(fhesi_ctx* ctx, u64* d_rows, i64 count, int nslots, const int* d_pos, const int* h_pos) {
  if (ctx->pow2) return launch_ntt_inv(ctx, d_rows, count, nslots, d_pos, true);
  return launch_bluestein_inv(ctx, d_rows, count, nslots, h_pos);
}

// device copy of an index list (small, cached per call in workspace slot 3)
static int upload_idx(fhesi_ctx* ctx, const std::vector<int>& idx, int** d_out) {
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
  {"tensor32", "FHESI_TENSOR32", offsetof(CtxOptions, tensor32), false},
  {"dot32_v3", "FHESI_DOT32_V3", offsetof(CtxOptions, dot32_v3), false},
  {"dot32_half", "FHESI_DOT32_HALF", offsetof(CtxOptions, dot32_half), false},
  {"dot32_mfma", "FHESI_DOT32_MFMA", offsetof(CtxOptions, dot32_mfma), false},
  {"automorph_rows", "FHESI_AUTOMORPH_ROWS", offsetof(CtxOptions, automorph_rows), false},
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
  if (!c->pow2 && m % 2 == 0 && (m / 2) % 2 == 1 && hm::is_prime((u64)(m / 2)) && 2 * c->phim - 1 <= kAux32N) c->lin_q = m / 2;
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
    pc.pad0 = 0;
    pc.r64 = (u64)(((u128)1 << 64) % Q);
    pc.r64_sh = hm::shoup(pc.r64, Q);
    pc.one_sh = hm::shoup(1, Q);
    // modulus of the tile kernels: q itself, or for small primes the largest multiple of q below 2^60 (ntt_tile.inc)
    pc.q_tile = Q >= (1ull << 48) ? Q : Q * (((1ull << 60) - 1) / Q);
    pc.one_q63 = hm::shoup63(1, pc.q_tile);
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

// --------------------------------------------------------------------------------------------- Cmodulus::FFT / iFFT
extern "C" int fhesi_cmod_fft(fhesi_ctx* c, int32_t prime, const uint64_t* limbs, int32_t nlimbs, int64_t ncoeffs, uint64_t* y) {
  CHECK_CTX(c);
  if (prime < 0 || prime >= c->L) FHESI_FAIL("Cmodulus::FFT: prime index %d out of range", prime);
  if (nlimbs < 1 || ncoeffs < 0) FHESI_FAIL("Cmodulus::FFT: bad coefficient shape");
  const i64 n = c->phim;
  // conv(in,x) (CModulus.cpp:96) on the host for this single-row compatibility entry; coefficients of degree >= m are
  // ignored (bluestein.cpp:111-113) and degrees phi(m)..m-1 are folded modulo Phi_m so that one length-phi(m) row goes in.
  std::vector<u64> res(c->m, 0);
  const u64 Q = c->q[prime];
  for (i64 k = 0; k < ncoeffs && k < c->m; ++k) res[k] = hm::bn_mod((const u64*)limbs + k * nlimbs, nlimbs, Q);
  for (i64 k = c->m - 1; k >= n; --k) {          // reduce modulo the monic Phi_m over Z_q
    const u64 cc = res[k];
    if (!cc) continue;
    res[k] = 0;
    for (i64 j = 0; j < n; ++j) {
      const i64 f = c->phi[j];
      if (!f) continue;
      const u64 fm = f < 0 ? (Q - ((u64)(-f) % Q)) % Q : (u64)f % Q;
      res[k - n + j] = (res[k - n + j] + Q - hm::mulmod(cc, fm, Q)) % Q;
    }
  }
  void* d;
  FHESI_TRY(ws_reserve(c, 0, n * 8, &d));
  HIP_TRY(hipMemcpyAsync(d, res.data(), n * 8, hipMemcpyHostToDevice, c->stream));
  std::vector<int> pos(1, prime);
  int* d_pos;
  FHESI_TRY(upload_idx(c, pos, &d_pos));
  FHESI_TRY(row_fwd(c, (u64*)d, 1, 1, d_pos, pos.data()));
  HIP_TRY(hipMemcpyAsync(y, d, n * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int fhesi_cmod_ifft(fhesi_ctx* c, int32_t prime, const uint64_t* y, uint64_t* x) {
  CHECK_CTX(c);
  if (prime < 0 || prime >= c->L) FHESI_FAIL("Cmodulus::iFFT: prime index %d out of range", prime);
  const i64 n = c->phim;
  void* d;
  FHESI_TRY(ws_reserve(c, 0, n * 8, &d));
  HIP_TRY(hipMemcpyAsync(d, y, n * 8, hipMemcpyHostToDevice, c->stream));
  std::vector<int> pos(1, prime);
  int* d_pos;
  FHESI_TRY(upload_idx(c, pos, &d_pos));
  FHESI_TRY(row_inv(c, (u64*)d, 1, 1, d_pos, pos.data()));
  HIP_TRY(hipMemcpyAsync(x, d, n * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// --------------------------------------------------------------------------------------------- DoubleCRT objects
static int slot_of(const fhesi_dcrt* d, int prime) {
  auto it = std::lower_bound(d->idx.begin(), d->idx.end(), prime);
  return (it != d->idx.end() && *it == prime) ? (int)(it - d->idx.begin()) : -1;
}

extern "C" int fhesi_dcrt_alloc(fhesi_ctx* c, const int32_t* prime_idx, int32_t nidx, fhesi_dcrt** out) {
  CHECK_CTX(c);
  fhesi_dcrt* d = new fhesi_dcrt();
  d->ctx = c;
  if (nidx == 0) { d->idx.resize(c->L); for (int i = 0; i < c->L; ++i) d->idx[i] = i; }
  else {
    d->idx.assign(prime_idx, prime_idx + nidx);
    for (int i = 0; i < nidx; ++i)
      if (d->idx[i] < 0 || d->idx[i] >= c->L || (i && d->idx[i] <= d->idx[i - 1])) { delete d; FHESI_FAIL("DoubleCRT: index set must be ascending and inside the chain"); }   // DoubleCRT.cpp:215
  }
  const size_t bytes = d->idx.size() * c->phim * 8;
  HIP_TRY(hipMalloc(&d->d_rows, bytes ? bytes : 8));
  HIP_TRY(hipMemsetAsync(d->d_rows, 0, bytes, c->stream));
  ++c->live_handles;
  *out = d;
  return 0;
}
extern "C" int fhesi_dcrt_free(fhesi_dcrt* d) {
  if (!d) return 0;
  hipSetDevice(d->ctx->device);
  hipStreamSynchronize(d->ctx->stream);
  hipFree(d->d_rows);
  --d->ctx->live_handles;
  delete d;
  return 0;
}
static int dcrt_resize(fhesi_dcrt* d, const std::vector<int>& idx) {
  if (idx.size() != d->idx.size()) {
    HIP_TRY(hipStreamSynchronize(d->ctx->stream));
    HIP_TRY(hipFree(d->d_rows));
    HIP_TRY(hipMalloc(&d->d_rows, std::max<size_t>(8, idx.size() * d->ctx->phim * 8)));
  }
  d->idx = idx;
  return 0;
}
extern "C" int fhesi_dcrt_copy(fhesi_dcrt* dst, const fhesi_dcrt* src) {
  if (!dst || !src) FHESI_FAIL("null DoubleCRT");
  if (dst->ctx != src->ctx) FHESI_FAIL("DoubleCRT assigment: incompatible contexts");   // DoubleCRT.cpp:315-316 (SingleCRT.cpp:223-224)
  if (dst->coeff_form != src->coeff_form) FHESI_FAIL("assignment between a DoubleCRT and a SingleCRT handle: use fhesi_dcrt_assign_scrt / fhesi_scrt_assign_dcrt");
  CHECK_CTX(dst->ctx);
  if (dst == src) return 0;
  FHESI_TRY(dcrt_resize(dst, src->idx));
  HIP_TRY(hipMemcpyAsync(dst->d_rows, src->d_rows, src->idx.size() * src->ctx->phim * 8, hipMemcpyDeviceToDevice, dst->ctx->stream));
  return 0;
}
extern "C" int fhesi_dcrt_index_set(const fhesi_dcrt* d, int32_t* idx_out, int32_t* nidx) {
  if (!d) FHESI_FAIL("null DoubleCRT");
  *nidx = (int32_t)d->idx.size();
  if (idx_out) for (size_t i = 0; i < d->idx.size(); ++i) idx_out[i] = d->idx[i];
  return 0;
}
extern "C" int fhesi_dcrt_equal(const fhesi_dcrt* a, const fhesi_dcrt* b, int32_t* equal) {
  if (!a || !b) FHESI_FAIL("null DoubleCRT");
  *equal = 0;
  if (a->ctx != b->ctx || a->idx != b->idx || a->coeff_form != b->coeff_form) return 0;    // DoubleCRT.h:167-169, SingleCRT.h:98-100
  CHECK_CTX(a->ctx);
  int eq = 0;
  FHESI_TRY(launch_rows_equal(a->ctx, a->d_rows, b->d_rows, (i64)a->idx.size() * a->ctx->phim, &eq));
  *equal = eq;
  return 0;
}
extern "C" int fhesi_dcrt_upload_row(fhesi_dcrt* d, int32_t prime, const uint64_t* row) {
  if (!d) FHESI_FAIL("null DoubleCRT");
  CHECK_CTX(d->ctx);
  const int s = slot_of(d, prime);
  if (s < 0) FHESI_FAIL("DoubleCRT: prime %d not in the index set", prime);
  const i64 n = d->ctx->phim;
  const u64 Q = d->ctx->q[prime];
  for (i64 j = 0; j < n; ++j) if (row[j] >= Q) FHESI_FAIL("DoubleCRT object has inconsistent data");   // DoubleCRT::verify, DoubleCRT.cpp:66-68
  HIP_TRY(hipMemcpyAsync(d->d_rows + (i64)s * n, row, n * 8, hipMemcpyHostToDevice, d->ctx->stream));
  HIP_TRY(hipStreamSynchronize(d->ctx->stream));
  return 0;
}
extern "C" int fhesi_dcrt_download_row(const fhesi_dcrt* d, int32_t prime, uint64_t* row) {
  if (!d) FHESI_FAIL("null DoubleCRT");
  CHECK_CTX(d->ctx);
  const int s = slot_of(d, prime);
  if (s < 0) FHESI_FAIL("DoubleCRT: prime %d not in the index set", prime);
  const i64 n = d->ctx->phim;
  HIP_TRY(hipMemcpyAsync(row, d->d_rows + (i64)s * n, n * 8, hipMemcpyDeviceToHost, d->ctx->stream));
  HIP_TRY(hipStreamSynchronize(d->ctx->stream));
  return 0;
}
extern "C" void* fhesi_dcrt_device_ptr(fhesi_dcrt* d) { return d ? d->d_rows : nullptr; }

extern "C" int fhesi_dcrt_from_poly(fhesi_dcrt* d, const uint64_t* limbs, int32_t nlimbs, int64_t ncoeffs) {
  if (!d) FHESI_FAIL("null DoubleCRT");
  if (d->coeff_form) FHESI_FAIL("DoubleCRT call on a SingleCRT handle");
  fhesi_ctx* c = d->ctx;
  CHECK_CTX(c);
  if (nlimbs < 1 || ncoeffs < 0) FHESI_FAIL("DoubleCRT(ZZX): bad coefficient shape");
  const i64 n = c->phim;
  const int K = (int)d->idx.size();
  if (!K) return 0;
  if (ncoeffs > n) {
    // a polynomial of degree >= phi(m): take the per-row compatibility path (host reduction modulo Phi_m, CModulus.cpp:96-99)
    std::vector<u64> y(n);
    for (int s = 0; s < K; ++s) {
      FHESI_TRY(fhesi_cmod_fft(c, d->idx[s], limbs, nlimbs, ncoeffs, y.data()));
      HIP_TRY(hipMemcpyAsync(d->d_rows + (i64)s * n, y.data(), n * 8, hipMemcpyHostToDevice, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return 0;
  }
  void* d_l;
  FHESI_TRY(ws_reserve(c, 0, std::max<size_t>(8, (size_t)ncoeffs * nlimbs * 8), &d_l));
  HIP_TRY(hipMemcpyAsync(d_l, limbs, (size_t)ncoeffs * nlimbs * 8, hipMemcpyHostToDevice, c->stream));
  int* d_pos;
  FHESI_TRY(upload_idx(c, d->idx, &d_pos));
  FHESI_TRY(launch_rns_reduce(c, (const u64*)d_l, nlimbs, ncoeffs, 1, 1, nullptr, d->d_rows, K, d_pos));
  FHESI_TRY(row_fwd(c, d->d_rows, 1, K, d_pos, d->idx.data()));
  HIP_TRY(hipStreamSynchronize(c->stream));   // caller's limbs buffer may be released
  return 0;
}

extern "C" int fhesi_dcrt_to_poly(const fhesi_dcrt* d, const int32_t* prime_idx, int32_t nidx, int32_t positive, uint64_t* out, int32_t nlimbs) {
  if (!d) FHESI_FAIL("null DoubleCRT");
  if (d->coeff_form) FHESI_FAIL("DoubleCRT call on a SingleCRT handle");
  fhesi_ctx* c = d->ctx;
  CHECK_CTX(c);
  const i64 n = c->phim;
  // s1 = map.getIndexSet() & s  (DoubleCRT.cpp:352)
  std::vector<int> s1;
  if (nidx == 0 && prime_idx == nullptr) s1 = d->idx;
  else for (int i = 0; i < nidx; ++i) if (slot_of(d, prime_idx[i]) >= 0) s1.push_back(prime_idx[i]);
  std::sort(s1.begin(), s1.end());
  s1.erase(std::unique(s1.begin(), s1.end()), s1.end());
  if (s1.empty()) { memset(out, 0, (size_t)n * nlimbs * 8); return 0; }      // :354-357
  const int K = (int)s1.size();
  // inverse transforms on a scratch copy of the selected rows
  void* d_tmp;
  FHESI_TRY(ws_reserve(c, 0, (size_t)K * n * 8, &d_tmp));
  for (int k = 0; k < K; ++k)
    HIP_TRY(hipMemcpyAsync((u64*)d_tmp + (i64)k * n, d->d_rows + (i64)slot_of(d, s1[k]) * n, n * 8, hipMemcpyDeviceToDevice, c->stream));
  int* d_pos;
  FHESI_TRY(upload_idx(c, s1, &d_pos));
  FHESI_TRY(row_inv(c, (u64*)d_tmp, 1, K, d_pos, s1.data()));
  CrtTables* t;
  FHESI_TRY(get_crt_tables(c, s1, &t));
  // slots in the scratch layout are 0..K-1
  std::vector<int> slots(K);
  for (int k = 0; k < K; ++k) slots[k] = k;
  void* d_slots;
  FHESI_TRY(ws_reserve(c, 4, K * sizeof(int) + 64, &d_slots));
  HIP_TRY(hipMemcpyAsync(d_slots, slots.data(), K * sizeof(int), hipMemcpyHostToDevice, c->stream));
  void* d_out;
  FHESI_TRY(ws_reserve(c, 1, (size_t)n * nlimbs * 8, &d_out));
  FHESI_TRY(launch_crt(c, t, (const u64*)d_tmp, K, (const int*)d_slots, 1, 0, positive, 0, (u64*)d_out, nlimbs));
  HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)n * nlimbs * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int fhesi_dcrt_op(fhesi_dcrt* dst, const fhesi_dcrt* src, int32_t op) {
  if (!dst || !src) FHESI_FAIL("null DoubleCRT");
  if (dst->ctx != src->ctx) FHESI_FAIL("DoubleCRT::Op: incompatible objects");           // DoubleCRT.cpp:82-83
  if (dst->idx != src->idx) FHESI_FAIL("DoubleCRT::Op: index sets differ (match them with add_primes first)");
  if (op < FHESI_OP_ADD || op > FHESI_OP_MUL) FHESI_FAIL("DoubleCRT::Op: unknown operation %d", op);
  if (dst->coeff_form != src->coeff_form) FHESI_FAIL("Op between a DoubleCRT and a SingleCRT handle");
  if (dst->coeff_form && op == FHESI_OP_MUL) FHESI_FAIL("SingleCRT::Op: only AddMod / SubMod exist (SingleCRT.h:127-133)");
  fhesi_ctx* c = dst->ctx;
  CHECK_CTX(c);
  int* d_pos;
  FHESI_TRY(upload_idx(c, dst->idx, &d_pos));
  return launch_ew_op(c, dst->d_rows, src->d_rows, 1, (int)dst->idx.size(), d_pos, op);
}

extern "C" int fhesi_dcrt_op_scalar(fhesi_dcrt* d, const uint64_t* num, int32_t nlimbs, int32_t op) {
  if (!d) FHESI_FAIL("null DoubleCRT");
  if (d->coeff_form) FHESI_FAIL("DoubleCRT call on a SingleCRT handle");
  fhesi_ctx* c = d->ctx;
  CHECK_CTX(c);
  if (op < FHESI_OP_ADD || op > FHESI_OP_SET) FHESI_FAIL("DoubleCRT scalar op: unknown operation %d", op);
  const int K = (int)d->idx.size();
  if (!K) return 0;
  std::vector<u64> sc(K);
  for (int s = 0; s < K; ++s) {
    const u64 Q = c->q[d->idx[s]];
    u64 v = hm::bn_mod((const u64*)num, nlimbs, Q);                 // n = rem(num, pi)  (DoubleCRT.cpp:123)
    if (op == FHESI_OP_DIV) {
      if (v == 0) FHESI_FAIL("DoubleCRT::operator/=: divisor is zero modulo prime %d", d->idx[s]);   // InvMod error
      v = hm::invmod(v, Q);                                         // :416
    }
    sc[s] = v;
  }
  void* d_sc;
  FHESI_TRY(ws_reserve(c, 4, K * 8 + 64, &d_sc));
  HIP_TRY(hipMemcpyAsync(d_sc, sc.data(), K * 8, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  int* d_pos;
  FHESI_TRY(upload_idx(c, d->idx, &d_pos));
  return launch_ew_scalar(c, d->d_rows, (const u64*)d_sc, 1, K, d_pos, op);
}

extern "C" int fhesi_dcrt_exp(fhesi_dcrt* d, int64_t e) {
  if (!d) FHESI_FAIL("null DoubleCRT");
  if (d->coeff_form) FHESI_FAIL("DoubleCRT call on a SingleCRT handle");
  fhesi_ctx* c = d->ctx;
  CHECK_CTX(c);
  const int K = (int)d->idx.size();
  if (!K) return 0;
  // PowerMod(a, e, q) with e < 0 is (a^-1)^|e| = a^((q-1) - |e| mod (q-1)) for a != 0, and NTL's InvMod error for a = 0
  std::vector<u64> ex(K);
  for (int s = 0; s < K; ++s) {
    const u64 ord = c->q[d->idx[s]] - 1;
    if (e >= 0) ex[s] = (u64)e;
    else { const u64 r = (0 - (u64)e) % ord; ex[s] = r ? ord - r : 0; }
  }
  void* d_ex;
  FHESI_TRY(ws_reserve(c, 4, K * 8 + 64, &d_ex));
  unsigned* d_flag = (unsigned*)((u64*)d_ex + K);
  HIP_TRY(hipMemcpyAsync(d_ex, ex.data(), K * 8, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemsetAsync(d_flag, 0, 4, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  int* d_pos;
  FHESI_TRY(upload_idx(c, d->idx, &d_pos));
  if (e < 0) {
    FHESI_TRY(launch_ew_exp(c, d->d_rows, (const u64*)d_ex, 1, K, d_pos, d_flag));
    unsigned flag = 0;
    HIP_TRY(hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (flag) FHESI_FAIL("DoubleCRT::Exp: negative exponent of a zero element (InvMod: inverse undefined)");
  }
  return launch_ew_exp(c, d->d_rows, (const u64*)d_ex, 1, K, d_pos, nullptr);
}

extern "C" int fhesi_dcrt_automorph(fhesi_dcrt* d, int64_t k) {
  if (!d) FHESI_FAIL("null DoubleCRT");
  if (d->coeff_form) FHESI_FAIL("DoubleCRT call on a SingleCRT handle");
  fhesi_ctx* c = d->ctx;
  CHECK_CTX(c);
  if (k <= 0 || k >= c->m || c->zms_idx[k] < 0) FHESI_FAIL("DoubleCRT::automorph: k not in Zm*");     // DoubleCRT.cpp:442-443
  const i64 K = (i64)d->idx.size();
  if (!K) return 0;
  void* tmp;
  FHESI_TRY(ws_reserve(c, 0, (size_t)K * c->phim * 8, &tmp));
  HIP_TRY(hipMemcpyAsync(tmp, d->d_rows, (size_t)K * c->phim * 8, hipMemcpyDeviceToDevice, c->stream));
  return launch_automorph(c, d->d_rows, (const u64*)tmp, K, k);
}

extern "C" int fhesi_dcrt_add_primes(fhesi_dcrt* d, const int32_t* prime_idx, int32_t nidx) {
  if (!d) FHESI_FAIL("null DoubleCRT");
  if (d->coeff_form) FHESI_FAIL("DoubleCRT call on a SingleCRT handle");
  fhesi_ctx* c = d->ctx;
  CHECK_CTX(c);
  if (nidx == 0) return 0;                                                               // DoubleCRT.cpp:144
  std::vector<int> add(prime_idx, prime_idx + nidx);
  std::sort(add.begin(), add.end());
  for (int p : add) {
    if (p < 0 || p >= c->L) FHESI_FAIL("addPrimes: prime index %d out of range", p);
    if (slot_of(d, p) >= 0) FHESI_FAIL("addPrimes: index sets must be disjoint");        // :145
  }
  const i64 n = c->phim;
  // toPoly over the current set (:147-148) -- wide enough for the product of the current primes
  const int W = (int)d->idx.size() + 2;
  std::vector<u64> poly((size_t)n * W);
  FHESI_TRY(fhesi_dcrt_to_poly(d, nullptr, 0, 0, poly.data(), W));
  // new object over the union; old rows kept, new rows = FFT of poly (:150-155)
  std::vector<int> uni(d->idx);
  uni.insert(uni.end(), add.begin(), add.end());
  std::sort(uni.begin(), uni.end());
  u64* d_new;
  HIP_TRY(hipMalloc(&d_new, uni.size() * n * 8));
  fhesi_dcrt tmp;
  tmp.ctx = c; tmp.idx = add;
  HIP_TRY(hipMalloc(&tmp.d_rows, add.size() * n * 8));
  int r = fhesi_dcrt_from_poly(&tmp, poly.data(), W, n);
  if (!r) {
    for (size_t u = 0; u < uni.size(); ++u) {
      const int so = slot_of(d, uni[u]);
      const u64* src = so >= 0 ? d->d_rows + (i64)so * n : tmp.d_rows + (i64)slot_of(&tmp, uni[u]) * n;
      hipMemcpyAsync(d_new + (i64)u * n, src, n * 8, hipMemcpyDeviceToDevice, c->stream);
    }
    hipStreamSynchronize(c->stream);
    hipFree(d->d_rows);
    d->d_rows = d_new;
    d->idx = uni;
  } else hipFree(d_new);
  hipFree(tmp.d_rows);
  return r;
}

extern "C" int fhesi_dcrt_remove_primes(fhesi_dcrt* d, const int32_t* prime_idx, int32_t nidx) {
  if (!d) FHESI_FAIL("null DoubleCRT");
  fhesi_ctx* c = d->ctx;
  CHECK_CTX(c);
  std::vector<int> keep;
  for (int p : d->idx) if (std::find(prime_idx, prime_idx + nidx, p) == prime_idx + nidx) keep.push_back(p);
  if (keep.size() == d->idx.size()) return 0;
  const i64 n = c->phim;
  u64* d_new;
  HIP_TRY(hipMalloc(&d_new, std::max<size_t>(8, keep.size() * n * 8)));
  for (size_t u = 0; u < keep.size(); ++u)
    HIP_TRY(hipMemcpyAsync(d_new + (i64)u * n, d->d_rows + (i64)slot_of(d, keep[u]) * n, n * 8, hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipFree(d->d_rows));
  d->d_rows = d_new;
  d->idx = keep;
  return 0;
}

// magnitude of a non-negative big integer modulo a word (host scalars of the modulus-switching methods)
static u64 bn_mag_mod(const std::vector<u64>& a, u64 q) { u64 r = 0; for (size_t i = a.size(); i-- > 0;) r = (u64)((((u128)r << 64) | a[i]) % q); return r; }
static u64 inv_mod_word(u64 a, u64 p) {      // a^-1 mod p for any p > 1 with gcd(a, p) = 1 (NTL InvMod); 0 if not invertible
  __int128 t = 0, nt = 1, r = p, nr = a % p;
  while (nr) { const __int128 qq = r / nr; __int128 tmp = t - qq * nt; t = nt; nt = tmp; tmp = r - qq * nr; r = nr; nr = tmp; }
  if (r != 1) return 0;
  if (t < 0) t += p;
  return (u64)t;
}

// replaces the row storage of d by the rows of the ascending set `idx_new`: rows present in the old set are copied, the others zero-filled
static int dcrt_reindex(fhesi_dcrt* d, const std::vector<int>& idx_new) {
  fhesi_ctx* c = d->ctx;
  const i64 n = c->phim;
  u64* d_new;
  HIP_TRY(hipMalloc(&d_new, std::max<size_t>(8, idx_new.size() * n * 8)));
  for (size_t u = 0; u < idx_new.size(); ++u) {
    const int so = slot_of(d, idx_new[u]);
    if (so >= 0) HIP_TRY(hipMemcpyAsync(d_new + (i64)u * n, d->d_rows + (i64)so * n, n * 8, hipMemcpyDeviceToDevice, c->stream));
    else HIP_TRY(hipMemsetAsync(d_new + (i64)u * n, 0, n * 8, c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipFree(d->d_rows));
  d->d_rows = d_new;
  d->idx = idx_new;
  return 0;
}

// rows of d (all slots) *= per-slot word scalars
static int dcrt_scale_rows(fhesi_dcrt* d, const std::vector<u64>& sc) {
  fhesi_ctx* c = d->ctx;
  const int K = (int)d->idx.size();
  void* d_sc;
  FHESI_TRY(ws_reserve(c, 4, K * 8 + 64, &d_sc));
  HIP_TRY(hipMemcpyAsync(d_sc, sc.data(), K * 8, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  int* d_pos;
  FHESI_TRY(upload_idx(c, d->idx, &d_pos));
  return launch_ew_scalar(c, d->d_rows, (const u64*)d_sc, 1, K, d_pos, FHESI_OP_MUL);
}

extern "C" int fhesi_dcrt_add_primes_and_scale(fhesi_dcrt* d, const int32_t* prime_idx, int32_t nidx, uint64_t p, double* log_factor_out) {
  if (!d) FHESI_FAIL("null DoubleCRT");
  if (d->coeff_form) FHESI_FAIL("DoubleCRT call on a SingleCRT handle");
  fhesi_ctx* c = d->ctx;
  CHECK_CTX(c);
  if (log_factor_out) *log_factor_out = 0.0;
  if (nidx == 0) return 0;                                                                // DoubleCRT.cpp:165
  if (p < 2) FHESI_FAIL("addPrimesAndScale: plaintext modulus must be at least 2");      // :166
  std::vector<int> add(prime_idx, prime_idx + nidx);
  std::sort(add.begin(), add.end());
  for (size_t i = 0; i < add.size(); ++i) {
    if (add[i] < 0 || add[i] >= c->L || (i && add[i] == add[i - 1])) FHESI_FAIL("addPrimesAndScale: prime index %d out of range or repeated", add[i]);
    if (slot_of(d, add[i]) >= 0) FHESI_FAIL("addPrimesAndScale: index sets must be disjoint");   // :167
  }
  // factor = prod q_i * ((prod q_i)^-1 mod p)   (:170-182); only its residues modulo the existing primes reach the device
  std::vector<u64> factor{1};
  double lf = 0.0;
  for (int i : add) { factor = hm::bn_mul_small(factor, c->q[i]); lf += std::log((double)c->q[i]); }
  const u64 prodInv = inv_mod_word(bn_mag_mod(factor, p), p);
  if (!prodInv) FHESI_FAIL("addPrimesAndScale: product of the added primes is not invertible modulo p (InvMod)");
  factor = hm::bn_mul_small(factor, prodInv);
  lf += std::log((double)prodInv);
  if (!d->idx.empty()) {
    std::vector<u64> sc(d->idx.size());
    for (size_t s = 0; s < d->idx.size(); ++s) sc[s] = bn_mag_mod(factor, c->q[d->idx[s]]);    // f = factor % qi (:190)
    FHESI_TRY(dcrt_scale_rows(d, sc));                                                     // MulModPrecon loop (:193-196)
  }
  std::vector<int> uni(d->idx);
  uni.insert(uni.end(), add.begin(), add.end());
  std::sort(uni.begin(), uni.end());
  FHESI_TRY(dcrt_reindex(d, uni));                                                         // new rows filled with zeros (:200-205)
  if (log_factor_out) *log_factor_out = lf;
  return 0;
}

extern "C" int fhesi_dcrt_scale_down_to_set(fhesi_dcrt* d, const int32_t* prime_idx, int32_t nidx, uint64_t p) {
  if (!d) FHESI_FAIL("null DoubleCRT");
  if (d->coeff_form) FHESI_FAIL("DoubleCRT call on a SingleCRT handle");
  fhesi_ctx* c = d->ctx;
  CHECK_CTX(c);
  if (p < 2) FHESI_FAIL("scaleDownToSet: plaintext modulus must be at least 2");
  std::vector<int> keep, diff;
  for (int q : d->idx) (std::find(prime_idx, prime_idx + nidx, q) != prime_idx + nidx ? keep : diff).push_back(q);
  if (keep.empty()) FHESI_FAIL("scaleDownToSet: the target set does not intersect the index set");      // assert(card(intersect) > 0), DoubleCRT.cpp:525
  if (diff.empty()) FHESI_FAIL("scaleDownToSet: no prime to drop");                                       // assert(card(diff) > 0), :526
  const i64 n = c->phim;
  const int K = (int)d->idx.size(), Kd = (int)diff.size(), Kk = (int)keep.size();
  // diffProd and the scalars derived from it (:528, :538)
  std::vector<u64> D{1};
  for (int i : diff) D = hm::bn_mul_small(D, c->q[i]);
  const u64 dp = bn_mag_mod(D, p);
  const u64 u = inv_mod_word(dp, p);
  if (!u) FHESI_FAIL("scaleDownToSet: product of the dropped primes is not invertible modulo p (InvMod)");
  // *this *= (diffProd % p)   (:529) -- every row, the dropped ones included
  {
    std::vector<u64> sc(K);
    for (int s = 0; s < K; ++s) sc[s] = dp % c->q[d->idx[s]];
    FHESI_TRY(dcrt_scale_rows(d, sc));
  }
  // toPoly(delta, diff)   (:531-532): inverse transforms of the dropped rows + CRT over them, centred modulo D
  CrtTables* t;
  FHESI_TRY(get_crt_tables(c, diff, &t));
  const int W = t->W + 1;                      // room for D p and a sign
  void *d_tmp, *d_delta, *d_e, *d_slots;
  FHESI_TRY(ws_reserve(c, 0, (size_t)std::max(Kd, Kk) * n * 8, &d_tmp));
  for (int k = 0; k < Kd; ++k)
    HIP_TRY(hipMemcpyAsync((u64*)d_tmp + (i64)k * n, d->d_rows + (i64)slot_of(d, diff[k]) * n, n * 8, hipMemcpyDeviceToDevice, c->stream));
  int* d_pos;
  FHESI_TRY(upload_idx(c, diff, &d_pos));
  FHESI_TRY(row_inv(c, (u64*)d_tmp, 1, Kd, d_pos, diff.data()));
  std::vector<int> slots(Kd);
  for (int k = 0; k < Kd; ++k) slots[k] = k;
  FHESI_TRY(ws_reserve(c, 4, Kd * sizeof(int) + 64, &d_slots));
  HIP_TRY(hipMemcpyAsync(d_slots, slots.data(), Kd * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  FHESI_TRY(ws_reserve(c, 1, (size_t)n * W * 8, &d_delta));
  FHESI_TRY(ws_reserve(c, 2, (size_t)n * W * 8, &d_e));
  FHESI_TRY(launch_crt(c, t, (const u64*)d_tmp, Kd, (const int*)d_slots, 1, 0, 0, 0, (u64*)d_delta, W));
  // delta <- delta * factor - delta, centred modulo D p   (:538-545) -- modswitch_delta_kernel
  std::vector<u64> consts((size_t)3 * W, 0);
  std::vector<u64> M = hm::bn_mul_small(D, p);
  if ((int)M.size() > W || (M.size() == (size_t)W && (M.back() >> 63))) FHESI_FAIL("scaleDownToSet: D p does not fit %d limbs", W);
  for (size_t i = 0; i < D.size(); ++i) consts[i] = D[i];
  for (size_t i = 0; i < M.size(); ++i) consts[W + i] = M[i];
  for (int i = 0; i < W; ++i) consts[2 * W + i] = (consts[W + i] >> 1) | (i + 1 < W ? consts[W + i + 1] << 63 : 0);
  FHESI_TRY(launch_modswitch_delta(c, (const u64*)d_delta, W, consts.data(), p, u, (u64*)d_e));
  // removePrimes(diff); *this += delta; *this /= diffProd   (:555-557)
  FHESI_TRY(dcrt_reindex(d, keep));
  FHESI_TRY(upload_idx(c, keep, &d_pos));
  FHESI_TRY(launch_rns_reduce(c, (const u64*)d_e, W, n, 1, 1, nullptr, (u64*)d_tmp, Kk, d_pos));
  FHESI_TRY(row_fwd(c, (u64*)d_tmp, 1, Kk, d_pos, keep.data()));
  FHESI_TRY(launch_ew_op(c, d->d_rows, (const u64*)d_tmp, 1, Kk, d_pos, FHESI_OP_ADD));
  std::vector<u64> sc(Kk);
  for (int s = 0; s < Kk; ++s) { const u64 q = c->q[keep[s]]; sc[s] = hm::invmod(bn_mag_mod(D, q), q); }
  FHESI_TRY(dcrt_scale_rows(d, sc));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int fhesi_dcrt_from_scrt(fhesi_dcrt* d, const uint64_t* coeff_rows) {
  if (!d) FHESI_FAIL("null DoubleCRT");
  fhesi_ctx* c = d->ctx;
  CHECK_CTX(c);
  const int K = (int)d->idx.size();
  const i64 n = c->phim;
  for (int s = 0; s < K; ++s) { const u64 Q = c->q[d->idx[s]]; for (i64 j = 0; j < n; ++j) if (coeff_rows[(i64)s * n + j] >= Q) FHESI_FAIL("SingleCRT object has inconsistent data"); }
  HIP_TRY(hipMemcpyAsync(d->d_rows, coeff_rows, (size_t)K * n * 8, hipMemcpyHostToDevice, c->stream));
  int* d_pos;
  FHESI_TRY(upload_idx(c, d->idx, &d_pos));
  FHESI_TRY(row_fwd(c, d->d_rows, 1, K, d_pos, d->idx.data()));     // DoubleCRT.cpp:493-494
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}
extern "C" int fhesi_dcrt_to_scrt(const fhesi_dcrt* d, uint64_t* out) {
  if (!d) FHESI_FAIL("null DoubleCRT");
  fhesi_ctx* c = d->ctx;
  CHECK_CTX(c);
  const int K = (int)d->idx.size();
  const i64 n = c->phim;
  void* tmp;
  FHESI_TRY(ws_reserve(c, 0, std::max<size_t>(8, (size_t)K * n * 8), &tmp));
  HIP_TRY(hipMemcpyAsync(tmp, d->d_rows, (size_t)K * n * 8, hipMemcpyDeviceToDevice, c->stream));
  int* d_pos;
  FHESI_TRY(upload_idx(c, d->idx, &d_pos));
  FHESI_TRY(row_inv(c, (u64*)tmp, 1, K, d_pos, d->idx.data()));     // DoubleCRT.cpp:508-509
  HIP_TRY(hipMemcpyAsync(out, tmp, (size_t)K * n * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// --------------------------------------------------------------------------------------------- SingleCRT (SingleCRT.h:41-175)
// A SingleCRT is the coefficient-domain RNS form: for every prime of its index set the polynomial's coefficients modulo that prime.
// It shares the handle type (and the row storage) of a DoubleCRT; fhesi_scrt_alloc marks the handle as holding coefficient residues
// and the entry points below refuse a handle of the wrong form.
#define CHECK_SCRT(s) do { if (!(s)) FHESI_FAIL("null SingleCRT"); if (!(s)->coeff_form) FHESI_FAIL("SingleCRT call on a DoubleCRT handle"); } while (0)
extern "C" int fhesi_scrt_alloc(fhesi_ctx* c, const int32_t* prime_idx, int32_t nidx, fhesi_dcrt** out) {
  FHESI_TRY(fhesi_dcrt_alloc(c, prime_idx, nidx, out));
  (*out)->coeff_form = true;
  return 0;
}
// SingleCRT::operator=(const ZZX&) (SingleCRT.cpp:239-251): PolyRed(poly, p_i, abs = true) per prime = coefficient residues in [0, p_i)
extern "C" int fhesi_scrt_from_poly(fhesi_dcrt* s, const uint64_t* limbs, int32_t nlimbs, int64_t ncoeffs) {
  CHECK_SCRT(s);
  fhesi_ctx* c = s->ctx;
  CHECK_CTX(c);
  const i64 n = c->phim;
  if (nlimbs < 1 || ncoeffs < 0) FHESI_FAIL("SingleCRT = ZZX: bad coefficient shape");
  if (ncoeffs > n) FHESI_FAIL("SingleCRT = ZZX: %lld coefficients, rows hold phi(m) = %lld", (long long)ncoeffs, (long long)n);
  const int K = (int)s->idx.size();
  if (!K) return 0;
  void* d_l;
  FHESI_TRY(ws_reserve(c, 0, std::max<size_t>(8, (size_t)ncoeffs * nlimbs * 8), &d_l));
  HIP_TRY(hipMemcpyAsync(d_l, limbs, (size_t)ncoeffs * nlimbs * 8, hipMemcpyHostToDevice, c->stream));
  int* d_pos;
  FHESI_TRY(upload_idx(c, s->idx, &d_pos));
  FHESI_TRY(launch_rns_reduce(c, (const u64*)d_l, nlimbs, ncoeffs, 1, 1, nullptr, s->d_rows, K, d_pos));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}
// SingleCRT::toPoly (SingleCRT.cpp:299-334): the same incremental CRT as DoubleCRT::toPoly, without the inverse transforms
extern "C" int fhesi_scrt_to_poly(const fhesi_dcrt* s, const int32_t* prime_idx, int32_t nidx, uint64_t* out, int32_t nlimbs) {
  CHECK_SCRT(s);
  fhesi_ctx* c = s->ctx;
  CHECK_CTX(c);
  const i64 n = c->phim;
  std::vector<int> s1;
  if (nidx == 0 && prime_idx == nullptr) s1 = s->idx;
  else for (int i = 0; i < nidx; ++i) if (slot_of(s, prime_idx[i]) >= 0) s1.push_back(prime_idx[i]);
  std::sort(s1.begin(), s1.end());
  s1.erase(std::unique(s1.begin(), s1.end()), s1.end());
  if (s1.empty()) { memset(out, 0, (size_t)n * nlimbs * 8); return 0; }             // :303-306
  const int K = (int)s1.size();
  CrtTables* t;
  FHESI_TRY(get_crt_tables(c, s1, &t));
  std::vector<int> slots(K);
  for (int k = 0; k < K; ++k) slots[k] = slot_of(s, s1[k]);
  void *d_slots, *d_out;
  FHESI_TRY(ws_reserve(c, 4, K * sizeof(int) + 64, &d_slots));
  HIP_TRY(hipMemcpyAsync(d_slots, slots.data(), K * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  FHESI_TRY(ws_reserve(c, 1, (size_t)n * nlimbs * 8, &d_out));
  FHESI_TRY(launch_crt(c, t, s->d_rows, (int)s->idx.size(), (const int*)d_slots, 1, 0, 0, 0, (u64*)d_out, nlimbs));
  HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)n * nlimbs * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}
// SingleCRT::Op(const ZZ&, add / sub / mul) (SingleCRT.cpp:137-153) and operator/= (:279-296).  NTL's add(ZZX, ZZX, ZZ) / sub touch the
// CONSTANT coefficient only; mul and the division by a constant act on every coefficient.
extern "C" int fhesi_scrt_op_scalar(fhesi_dcrt* s, const uint64_t* num, int32_t nlimbs, int32_t op) {
  CHECK_SCRT(s);
  fhesi_ctx* c = s->ctx;
  CHECK_CTX(c);
  if (op < FHESI_OP_ADD || op > FHESI_OP_DIV) FHESI_FAIL("SingleCRT scalar op: unknown operation %d", op);
  const int K = (int)s->idx.size();
  if (!K) return 0;
  std::vector<u64> sc(K);
  for (int k = 0; k < K; ++k) {
    const u64 Q = c->q[s->idx[k]];
    u64 v = hm::bn_mod((const u64*)num, nlimbs, Q);                 // rem(n, num, pi)  (:146, :287)
    if (op == FHESI_OP_DIV) {
      if (v == 0) FHESI_FAIL("SingleCRT::operator/=: divisor is zero modulo prime %d", s->idx[k]);   // InvMod error (:288)
      v = hm::invmod(v, Q);
    }
    sc[k] = v;
  }
  void* d_sc;
  FHESI_TRY(ws_reserve(c, 4, K * 8 + 64, &d_sc));
  HIP_TRY(hipMemcpyAsync(d_sc, sc.data(), K * 8, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  int* d_pos;
  FHESI_TRY(upload_idx(c, s->idx, &d_pos));
  if (op == FHESI_OP_ADD || op == FHESI_OP_SUB) return launch_scrt_const(c, s->d_rows, (const u64*)d_sc, K, d_pos, op == FHESI_OP_ADD ? 0 : 1);
  return launch_ew_scalar(c, s->d_rows, (const u64*)d_sc, 1, K, d_pos, FHESI_OP_MUL);
}
// DoubleCRT::operator=(const SingleCRT&) (DoubleCRT.cpp:484-496): index set of the SingleCRT, one forward transform per row, in HBM
extern "C" int fhesi_dcrt_assign_scrt(fhesi_dcrt* d, const fhesi_dcrt* s) {
  if (!d) FHESI_FAIL("null DoubleCRT");
  CHECK_SCRT(s);
  if (d->coeff_form) FHESI_FAIL("DoubleCRT = SingleCRT: the target handle is a SingleCRT");
  if (d->ctx != s->ctx) FHESI_FAIL("DoubleCRT=SingleCRT -- incompatible contexts");          // :486-487
  fhesi_ctx* c = d->ctx;
  CHECK_CTX(c);
  FHESI_TRY(dcrt_resize(d, s->idx));
  const int K = (int)d->idx.size();
  HIP_TRY(hipMemcpyAsync(d->d_rows, s->d_rows, (size_t)K * c->phim * 8, hipMemcpyDeviceToDevice, c->stream));
  int* d_pos;
  FHESI_TRY(upload_idx(c, d->idx, &d_pos));
  return row_fwd(c, d->d_rows, 1, K, d_pos, d->idx.data());                                   // :493-494
}
// DoubleCRT::toSingleCRT(scrt, s) (DoubleCRT.cpp:498-515): index set = s & the DoubleCRT's, one inverse transform per row, in HBM
extern "C" int fhesi_scrt_assign_dcrt(fhesi_dcrt* s, const fhesi_dcrt* d, const int32_t* prime_idx, int32_t nidx) {
  CHECK_SCRT(s);
  if (!d || d->coeff_form) FHESI_FAIL("toSingleCRT: the source is not a DoubleCRT");
  if (d->ctx != s->ctx) FHESI_FAIL("DoubleCRT::toSingleCRT -- incompatible contexts");        // :500-501
  fhesi_ctx* c = d->ctx;
  CHECK_CTX(c);
  std::vector<int> s1;
  if (nidx == 0 && prime_idx == nullptr) s1 = d->idx;
  else for (int q : d->idx) if (std::find(prime_idx, prime_idx + nidx, q) != prime_idx + nidx) s1.push_back(q);
  FHESI_TRY(dcrt_resize(s, s1));
  const i64 n = c->phim;
  for (size_t k = 0; k < s1.size(); ++k)
    HIP_TRY(hipMemcpyAsync(s->d_rows + (i64)k * n, d->d_rows + (i64)slot_of(d, s1[k]) * n, n * 8, hipMemcpyDeviceToDevice, c->stream));
  if (s1.empty()) return 0;
  int* d_pos;
  FHESI_TRY(upload_idx(c, s1, &d_pos));
  return row_inv(c, s->d_rows, 1, (int)s1.size(), d_pos, s1.data());                          // :508-509
}

// --------------------------------------------------------------------------------------------- batched row kernels
extern "C" int fhesi_rows_ntt_fwd_dev(fhesi_ctx* c, uint64_t* rows, int64_t count) {
  CHECK_CTX(c);
  std::vector<int> all(c->L);
  for (int i = 0; i < c->L; ++i) all[i] = i;
  return row_fwd(c, (u64*)rows, count, c->L, nullptr, all.data());
}
extern "C" int fhesi_rows_ntt_inv_dev(fhesi_ctx* c, uint64_t* rows, int64_t count) {
  CHECK_CTX(c);
  std::vector<int> all(c->L);
  for (int i = 0; i < c->L; ++i) all[i] = i;
  return row_inv(c, (u64*)rows, count, c->L, nullptr, all.data());
}
extern "C" int fhesi_rows_op_dev(fhesi_ctx* c, uint64_t* dst, const uint64_t* src, int64_t count, int32_t op) {
  CHECK_CTX(c);
  if (op < FHESI_OP_ADD || op > FHESI_OP_MUL) FHESI_FAIL("DoubleCRT::Op: unknown operation %d", op);
  return launch_ew_op(c, (u64*)dst, (const u64*)src, count, c->L, nullptr, op);
}

// --------------------------------------------------------------------------------------------- key-switch matrix
extern "C" int fhesi_ksk_create(fhesi_ctx* c, int32_t ncomp, int32_t ndigits, fhesi_ksk** out) {
  CHECK_CTX(c);
  if (ncomp < 1 || ndigits < 1) FHESI_FAIL("KeySwitchSI: bad shape");
  fhesi_ksk* k = new fhesi_ksk();
  k->ctx = c; k->ncomp = ncomp; k->ndigits = ndigits;
  k->bytes = (size_t)2 * ncomp * ndigits * c->L * c->phim * 8;
  HIP_TRY(hipMalloc(&k->d_rows, k->bytes));
  HIP_TRY(hipMemsetAsync(k->d_rows, 0, k->bytes, c->stream));
  ++c->live_handles;
  *out = k;
  return 0;
}
extern "C" int fhesi_ksk_free(fhesi_ksk* k) {
  if (!k) return 0;
  hipSetDevice(k->ctx->device);
  hipStreamSynchronize(k->ctx->stream);
  hipFree(k->d_rows);
  if (k->d_aux) hipFree(k->d_aux);
  if (k->d_mfma) hipFree(k->d_mfma);
  if (k->d_aux_consts) hipFree(k->d_aux_consts);
  if (k->d_limb_consts) hipFree(k->d_limb_consts);
  --k->ctx->live_handles;
  delete k;
  return 0;
}
extern "C" int fhesi_ksk_upload(fhesi_ksk* k, const uint64_t* rows_host) {
  if (!k) FHESI_FAIL("null key-switch matrix");
  CHECK_CTX(k->ctx);
  HIP_TRY(hipMemcpyAsync(k->d_rows, rows_host, k->bytes, hipMemcpyHostToDevice, k->ctx->stream));
  HIP_TRY(hipStreamSynchronize(k->ctx->stream));
  k->aux_valid = false;
  return 0;
}
// The library keeps tables derived from the rows; whoever writes the rows directly (a collective receiving into them) says so with
// fhesi_ksk_mark_dirty, and the next key switch rebuilds the tables.  The getter itself has no side effect.
extern "C" void* fhesi_ksk_device_ptr(fhesi_ksk* k) { return k ? k->d_rows : nullptr; }
extern "C" int fhesi_ksk_download(const fhesi_ksk* k, uint64_t* rows_host) {
  if (!k || !rows_host) FHESI_FAIL("null key-switch matrix");
  CHECK_CTX(k->ctx);
  HIP_TRY(hipMemcpyAsync(rows_host, k->d_rows, k->bytes, hipMemcpyDeviceToHost, k->ctx->stream));
  HIP_TRY(hipStreamSynchronize(k->ctx->stream));
  return 0;
}
extern "C" int fhesi_ksk_mark_dirty(fhesi_ksk* k) {
  if (!k) FHESI_FAIL("null key-switch matrix");
  k->aux_valid = false;
  return 0;
}
extern "C" int fhesi_ksk_upload_dev(fhesi_ksk* k, const uint64_t* rows_dev) {
  if (!k || !rows_dev) FHESI_FAIL("null key-switch matrix");
  CHECK_CTX(k->ctx);
  HIP_TRY(hipMemcpyAsync(k->d_rows, rows_dev, k->bytes, hipMemcpyDeviceToDevice, k->ctx->stream));
  HIP_TRY(hipStreamSynchronize(k->ctx->stream));
  k->aux_valid = false;
  return 0;
}
extern "C" size_t fhesi_ksk_bytes(const fhesi_ksk* k) { return k ? k->bytes : 0; }
extern "C" int fhesi_ksk_form(const fhesi_ksk* k, int32_t* form, int32_t* rows, int32_t* limb_bits) {
  if (!k) FHESI_FAIL("null key-switch matrix");
  if (form) *form = k->last_form;
  if (rows) *rows = k->last_form > 0 ? k->aux_rows : (k->last_form == 0 ? k->ctx->L : 0);
  if (limb_bits) *limb_bits = k->last_form > 0 ? k->aux_limb_bits : 0;
  return 0;
}

// --------------------------------------------------------------------------------------------- ciphertext pipeline
static i64 batch_chunk(fhesi_ctx* c, int ncol, bool ks32);
static std::vector<int> full_set(const fhesi_ctx* c) { std::vector<int> v(c->L); for (int i = 0; i < c->L; ++i) v[i] = i; return v; }

extern "C" int fhesi_ct_mul_dev(fhesi_ctx* c, uint64_t p, const uint64_t* a, const uint64_t* b, int32_t nlimbs, int64_t count, uint64_t* tprod) {
  CHECK_CTX(c);
  if (!count) return 0;
  const i64 n = c->phim;
  const int L = c->L;
  const std::vector<int> all = full_set(c);
  // c1[i] = DoubleCRT(parts[i].poly * p), c2[j] = DoubleCRT(other.parts[j].poly)   (Ciphertext.cpp:169-176)
  void* d_c;
  FHESI_TRY(ws_reserve(c, 0, (size_t)count * 4 * L * n * 8, &d_c));
  u64* ca = (u64*)d_c;
  u64* cb = ca + (size_t)count * 2 * L * n;
  const u64 lift[2] = {p, p};
  FHESI_TRY(launch_rns_reduce(c, (const u64*)a, nlimbs, n, count, 2, lift, ca, L, nullptr));
  FHESI_TRY(launch_rns_reduce(c, (const u64*)b, nlimbs, n, count, 2, nullptr, cb, L, nullptr));
  FHESI_TRY(row_fwd(c, ca, count * 4, L, nullptr, all.data()));
  // tProd[i+j] += c1[i] * c2[j]   (Ciphertext.cpp:179-186)
  FHESI_TRY(launch_tensor2x2(c, ca, cb, (u64*)tprod, count));
  return 0;
}

// ByteDecomp + DoubleCRT(digit polys) + DotProduct + toPoly + ReduceCoefficients (FHE-SI.cpp:244-256) from parts that are already
// positive residues mod 2^logQ in limb-major layout [count*ncomp][nlq][n].  d_t: scratch for count*2 DoubleCRTs, needed by the per-prime
// and the residue forms only (null: reserved here, workspace slot 1, when one of those runs -- the limb forms never touch it).
static int key_switch_tail(fhesi_ctx* c, const fhesi_ksk* k, int32_t logQ, int32_t decomp_bytes, const u64* d_parts, int64_t count, u64* d_t,
                           uint64_t* out, int32_t nlimbs) {
  const i64 n = c->phim;
  const int L = c->L, ncomp = k->ncomp, nd = k->ndigits, ncol = ncomp * nd, nlq = (logQ + 63) / 64;
  const std::vector<int> all = full_set(c);
  CrtTables* t;
  FHESI_TRY(get_crt_tables(c, all, &t));
  // Dot product through the two largest chain primes (kernels_ksaux.hip): 2 transforms per digit polynomial instead of L.
  // option ks_direct keeps the per-prime dot product below (A/B measurements; also the path of every shape the other does not cover).
  const int ks_mode = ksaux_mode(c, t, ncol, 8 * decomp_bytes, logQ);
  const_cast<fhesi_ksk*>(k)->last_form = ks_mode;
  if (ks_mode != KS_MODE_DIRECT) {
    fhesi_ksk* km = const_cast<fhesi_ksk*>(k);
    if (!k->aux_valid || k->aux_mode != ks_mode || k->aux_suborder != ntt_digits_suborder(c, 8 * decomp_bytes) || k->aux_logQ != logQ) FHESI_TRY(ksaux_build(c, km, 8 * decomp_bytes, logQ, ks_mode));
    const int R = k->aux_rows;        // L chain-prime residues, or the limbs of the key's integer coefficients (limb mode)
    const i64 nrow = k->aux32 ? aux32_row_len(c) : n;      // the 32-bit auxiliary rows always have 2^14 elements (four 4-byte residues = two 8-byte ones)
    void *d_dig, *d_o;
    FHESI_TRY(ws_reserve(c, 0, (size_t)count * ncol * 2 * nrow * 8, &d_dig));
    FHESI_TRY(ws_reserve(c, 10, (size_t)count * 2 * R * 2 * nrow * 8, &d_o));
    if (k->aux32) {       // four 30-bit primes (kernels_aux32.hip): the same buffer sizes, u32 rows
      FHESI_TRY(launch_ntt32_fwd_digits(c, d_parts, nlq, 8 * decomp_bytes, nd, count * ncomp, (u32*)d_dig, kDigitSubCt * ncol));
      if (c->mark_mid) { HIP_TRY(hipEventRecord(c->ev_mid, c->stream)); c->mark_mid = false; }
      bool mont = true;                 // dot32_kernel2 leaves the factor 2^-32 of its Montgomery step; the matrix-core form does not
      FHESI_TRY(launch_dot32(c, km, (const u32*)d_dig, ncol, count, (u32*)d_o, &mont));
      FHESI_TRY(launch_ntt32_inv(c, (u32*)d_o, count * 2 * R, 4, 0, mont));
      return launch_ks_recombine(c, t, k, (const u64*)d_o, count * 2, (u64*)out, nlimbs);
    }
    FHESI_TRY(launch_ntt_fwd_digits(c, d_parts, nlq, logQ, 8 * decomp_bytes, nd, count * ncomp, (u64*)d_dig, 0, 2, 2));
    if (c->mark_mid) { HIP_TRY(hipEventRecord(c->ev_mid, c->stream)); c->mark_mid = false; }
    FHESI_TRY(launch_dot_aux(c, k, (const u64*)d_dig, ncol, count, (u64*)d_o));
    FHESI_TRY(launch_ntt_inv(c, (u64*)d_o, count * 2 * R, 2, (const int*)(k->d_aux_consts + L), !k->aux_suborder));
    if (k->aux_limb_bits) return launch_ks_recombine(c, t, k, (const u64*)d_o, count * 2, (u64*)out, nlimbs);
    if (!d_t) { void* q; FHESI_TRY(ws_reserve(c, 1, (size_t)count * 2 * L * n * 8, &q)); d_t = (u64*)q; }
    FHESI_TRY(launch_aux_crt(c, k, (const u64*)d_o, d_t, count * 2 * L));
    return launch_crt(c, t, d_t, L, nullptr, count * 2, 2, 0, logQ, (u64*)out, nlimbs);
  }
  // ByteDecomp + DoubleCRT(digit polys)   (Ciphertext.cpp:82-121, FHE-SI.cpp:244-249)
  if (!d_t) { void* q; FHESI_TRY(ws_reserve(c, 1, (size_t)count * 2 * L * n * 8, &q)); d_t = (u64*)q; }
  void* d_dig;
  FHESI_TRY(ws_reserve(c, 0, (size_t)count * ncol * L * n * 8, &d_dig));
  if (c->pow2) FHESI_TRY(launch_ntt_fwd_digits(c, d_parts, nlq, logQ, 8 * decomp_bytes, nd, count * ncomp, (u64*)d_dig));
  else {
    FHESI_TRY(launch_digits(c, d_parts, nlq, logQ, 8 * decomp_bytes, nd, count * ncomp, (u64*)d_dig));
    FHESI_TRY(row_fwd(c, (u64*)d_dig, count * ncol, L, nullptr, all.data()));
  }
  if (c->mark_mid) { HIP_TRY(hipEventRecord(c->ev_mid, c->stream)); c->mark_mid = false; }
  // DotProduct with both key rows (FHE-SI.cpp:251-254)
  FHESI_TRY(launch_dot_accum(c, k->d_rows, (const u64*)d_dig, ncol, count, d_t, 0, 0, c->pow2 && ntt_digits_suborder(c, 8 * decomp_bytes)));
  // toPoly + ReduceCoefficients (FHE-SI.cpp:255-256)
  FHESI_TRY(row_inv(c, d_t, count * 2, L, nullptr, all.data()));
  FHESI_TRY(launch_crt(c, t, d_t, L, nullptr, count * 2, 2, 0, logQ, (u64*)out, nlimbs));
  return 0;
}

static int key_switch_args(fhesi_ctx* c, const fhesi_ksk* k, int32_t logQ, int32_t decomp_bytes, int32_t nlimbs) {
  if (!k || k->ctx != c) FHESI_FAIL("KeySwitchSI: context mismatch");            // FHE-SI.cpp:279-281
  if (decomp_bytes < 1 || decomp_bytes > 7) FHESI_FAIL("decompSize %d not supported", decomp_bytes);
  const int nd = (logQ + 8 * decomp_bytes - 1) / (8 * decomp_bytes);            // FHEContext.h:115
  if (nd != k->ndigits) FHESI_FAIL("KeySwitchSI: matrix has %d digits per component, context needs %d", k->ndigits, nd);
  if (nlimbs * 64 < logQ) FHESI_FAIL("output coefficients of %d limbs cannot hold logQ=%d bits", nlimbs, logQ);
  return 0;
}

// d_t: the scaled-up ciphertexts [count][ncomp][L][n], with room for max(ncomp, 2) parts per ciphertext; consumed (transformed in place,
// then reused for the dot product's rows)
static int apply_key_switch_consume(fhesi_ctx* c, const fhesi_ksk* k, int32_t logQ, int32_t decomp_bytes, u64* d_t, int64_t count, uint64_t* out, int32_t nlimbs) {
  const i64 n = c->phim;
  const int L = c->L, ncomp = k->ncomp, nlq = (logQ + 63) / 64;
  const std::vector<int> all = full_set(c);
  CrtTables* t;
  FHESI_TRY(get_crt_tables(c, all, &t));
  // ScaleDown (Ciphertext.cpp:194-218): toPoly + round(x/q) + Reduce, kept as positive residues for ByteDecomp
  FHESI_TRY(row_inv(c, d_t, count * ncomp, L, nullptr, all.data()));
  void* d_parts;
  FHESI_TRY(ws_reserve(c, 2, (size_t)count * ncomp * nlq * n * 8, &d_parts));
  FHESI_TRY(launch_crt(c, t, (const u64*)d_t, L, nullptr, count * ncomp, 1, 0, logQ, (u64*)d_parts, nlq));
  return key_switch_tail(c, k, logQ, decomp_bytes, (const u64*)d_parts, count, d_t, out, nlimbs);
}

extern "C" int fhesi_apply_key_switch_dev(fhesi_ctx* c, const fhesi_ksk* k, int32_t logQ, int32_t decomp_bytes, const uint64_t* tprod, int64_t count,
                                          uint64_t* out, int32_t nlimbs) {
  CHECK_CTX(c);
  FHESI_TRY(key_switch_args(c, k, logQ, decomp_bytes, nlimbs));
  if (!count) return 0;
  const i64 n = c->phim;
  const int L = c->L, ncomp = k->ncomp;
  void* d_t;
  FHESI_TRY(ws_reserve(c, 1, (size_t)count * (ncomp > 2 ? ncomp : 2) * L * n * 8, &d_t));
  HIP_TRY(hipMemcpyAsync(d_t, tprod, (size_t)count * ncomp * L * n * 8, hipMemcpyDeviceToDevice, c->stream));      // the caller keeps its tProd
  return apply_key_switch_consume(c, k, logQ, decomp_bytes, (u64*)d_t, count, out, nlimbs);
}

// Ciphertext::operator>>= (Ciphertext.cpp:264-269 -> CiphertextPart::operator>>= :54-59: DoubleCRT(poly) >>= k; toPoly) for a batch
// of unscaled ciphertexts, rows left in d_rows [count*nparts][L][n] in coefficient (post-iFFT) form ready for the CRT.
static int automorph_rows(fhesi_ctx* c, int64_t kk, const uint64_t* in, int32_t nparts, int32_t nlimbs_in, int64_t count, u64** d_rows_out) {
  const i64 n = c->phim, m = c->m;
  if (kk <= 0 || kk >= m || c->zms_idx[kk] < 0) FHESI_FAIL("automorph: k=%lld is not in Zm*", (long long)kk);     // DoubleCRT.cpp:442-443
  const int L = c->L;
  const std::vector<int> all = full_set(c);
  const i64 nrows = count * nparts * L;
  void *d_a, *d_b;
  FHESI_TRY(ws_reserve(c, 3, (size_t)nrows * n * 8, &d_a));
  FHESI_TRY(ws_reserve(c, 1, (size_t)(nrows > count * 2 * L ? nrows : count * 2 * L) * n * 8, &d_b));
  FHESI_TRY(launch_rns_reduce(c, (const u64*)in, nlimbs_in, n, count, nparts, nullptr, (u64*)d_a, L, nullptr));
  FHESI_TRY(row_fwd(c, (u64*)d_a, count * nparts, L, nullptr, all.data()));
  FHESI_TRY(launch_automorph(c, (u64*)d_b, (const u64*)d_a, nrows, kk));
  FHESI_TRY(row_inv(c, (u64*)d_b, count * nparts, L, nullptr, all.data()));
  *d_rows_out = (u64*)d_b;
  return 0;
}

extern "C" int fhesi_ct_automorph_dev(fhesi_ctx* c, int64_t kk, const uint64_t* in, int32_t nparts, int32_t nlimbs_in, int64_t count, uint64_t* out,
                                      int32_t nlimbs_out) {
  CHECK_CTX(c);
  if (nparts < 1 || nlimbs_in < 1 || nlimbs_out < 1) FHESI_FAIL("Ciphertext >>= : bad shape");
  if (!count) return 0;
  if (kk <= 0 || kk >= c->m || c->zms_idx[kk] < 0) FHESI_FAIL("automorph: k=%lld is not in Zm*", (long long)kk);     // DoubleCRT.cpp:442-443
  // the coefficient gather gives the integers a(X^k) mod Phi_m themselves; the reference's toPoly centres modulo the chain product: the
  // same thing as long as the sum of two input coefficients (64 nlimbs_in + 1 bits) stays below half of it
  double chain = 0;
  for (int i = 0; i < c->L; ++i) chain += std::log2((double)c->q[i]);
  if (!c->opt.automorph_rows && 64.0 * nlimbs_in + 2 < chain) {        // (fewer output limbs truncate the two's complement value in both forms)
    const int r = launch_ct_automorph_parts(c, (const u64*)in, nlimbs_in, count * nparts, kk, 0, (u64*)out, nlimbs_out);
    if (r != 2) return r;
  }
  u64* d_rows;
  FHESI_TRY(automorph_rows(c, kk, in, nparts, nlimbs_in, count, &d_rows));
  CrtTables* t;
  FHESI_TRY(get_crt_tables(c, full_set(c), &t));
  return launch_crt(c, t, d_rows, c->L, nullptr, count * nparts, 0, 0, 0, (u64*)out, nlimbs_out);
}

extern "C" int fhesi_ct_automorph_key_switch_dev(fhesi_ctx* c, const fhesi_ksk* k, int32_t logQ, int32_t decomp_bytes, int64_t kk, const uint64_t* in,
                                                 int32_t nlimbs_in, int64_t count, uint64_t* out, int32_t nlimbs) {
  CHECK_CTX(c);
  FHESI_TRY(key_switch_args(c, k, logQ, decomp_bytes, nlimbs));
  if (nlimbs_in < 1) FHESI_FAIL("Ciphertext >>= : bad shape");
  if (!count) return 0;
  const i64 n = c->phim;
  const int ncomp = k->ncomp, nlq = (logQ + 63) / 64;
  void* d_parts;
  FHESI_TRY(ws_reserve(c, 2, (size_t)count * ncomp * nlq * n * 8, &d_parts));
  u64* d_rows;
  if (kk <= 0 || kk >= c->m || c->zms_idx[kk] < 0) FHESI_FAIL("automorph: k=%lld is not in Zm*", (long long)kk);     // DoubleCRT.cpp:442-443
  if (!c->opt.automorph_rows) {
    // on power-of-two, prime and 2 x prime rings a(X^k) mod Phi_m is a signed gather of the coefficients: no row transform (kernels_ct.hip)
    const int r = launch_ct_automorph_parts(c, (const u64*)in, nlimbs_in, count * ncomp, kk, logQ, (u64*)d_parts, nlq);
    if (r == 1) return 1;
    if (r == 0) {
      return key_switch_tail(c, k, logQ, decomp_bytes, (const u64*)d_parts, count, nullptr, out, nlimbs);
    }
  }
  if (kk == 1) {
    // no automorphism: ApplyKeySwitch on the unscaled ciphertext as it is; ByteDecomp needs the positive residues limb-major
    FHESI_TRY(automorph_rows(c, 1, in, ncomp, nlimbs_in, count, &d_rows));
  } else {
    FHESI_TRY(automorph_rows(c, kk, in, ncomp, nlimbs_in, count, &d_rows));
  }
  CrtTables* t;
  FHESI_TRY(get_crt_tables(c, full_set(c), &t));
  // toPoly (centred modulo the whole chain), then Reduce(..., positive) of ByteDecompPart (Ciphertext.cpp:94)
  FHESI_TRY(launch_crt(c, t, d_rows, c->L, nullptr, count * ncomp, 3, 0, logQ, (u64*)d_parts, nlq));
  return key_switch_tail(c, k, logQ, decomp_bytes, (const u64*)d_parts, count, d_rows, out, nlimbs);
}

// ---- coefficient-domain ciphertext algebra on device batches (kernels_ct.hip)
extern "C" int fhesi_ct_add_dev(fhesi_ctx* c, int32_t logQ, uint64_t* dst, const uint64_t* src, int32_t nparts, int32_t nlimbs, int64_t count) {
  CHECK_CTX(c);
  if (nparts < 1 || nlimbs < 1 || logQ < 1 || nlimbs * 64 < logQ) FHESI_FAIL("Ciphertext += : coefficients of %d limbs cannot hold logQ=%d bits", nlimbs, logQ);
  return launch_ct_add(c, (u64*)dst, (const u64*)src, count * nparts * c->phim, nlimbs, logQ);
}
extern "C" int fhesi_ct_mul_long_dev(fhesi_ctx* c, int32_t logQ, uint64_t* ct, int64_t l, int32_t nparts, int32_t nlimbs, int64_t count) {
  CHECK_CTX(c);
  if (nparts < 1 || nlimbs < 1 || logQ < 1 || nlimbs * 64 < logQ) FHESI_FAIL("Ciphertext *= long: coefficients of %d limbs cannot hold logQ=%d bits", nlimbs, logQ);
  return launch_ct_mul_long(c, (u64*)ct, count * nparts * c->phim, nlimbs, logQ, l);
}
// Ciphertext::operator+=(const ZZX&) / (const ZZ_pX&) on unscaled ciphertexts (Ciphertext.cpp:147-161)
extern "C" int fhesi_ct_add_const_dev(fhesi_ctx* c, int32_t logQ, uint64_t p, uint64_t* ct, int32_t nparts, int32_t nlimbs, int64_t count, const int64_t* poly_host, int32_t npoly) {
  CHECK_CTX(c);
  if (nparts < 1 || nlimbs < 1 || logQ < 1 || nlimbs * 64 < logQ) FHESI_FAIL("Ciphertext += ZZX: coefficients of %d limbs cannot hold logQ=%d bits", nlimbs, logQ);
  if (p < 2) FHESI_FAIL("Ciphertext += ZZX: plaintext modulus %llu", (unsigned long long)p);
  if (npoly != 1 && npoly != count) FHESI_FAIL("Ciphertext += ZZX: %d constants for %lld ciphertexts (one for all, or one each)", npoly, (long long)count);
  if (!count) return 0;
  void* d_poly;
  FHESI_TRY(ws_reserve(c, 9, (size_t)npoly * c->phim * 8, &d_poly));
  HIP_TRY(hipMemcpyAsync(d_poly, poly_host, (size_t)npoly * c->phim * 8, hipMemcpyHostToDevice, c->stream));
  const int rc = launch_ct_add_const(c, (u64*)ct, (const i64*)d_poly, npoly, nparts, nlimbs, logQ, p, count);
  HIP_TRY(hipStreamSynchronize(c->stream));        // poly_host may be released on return
  return rc;
}
// Ciphertext::operator*=(const ZZX&) / (const ZZ_pX&) on unscaled ciphertexts (Ciphertext.cpp:245-252 -> CiphertextPart::operator*=(ZZX) :29-36):
// parts[i].poly *= other as INTEGER polynomials, rem Phi_m, Reduce.  The integer product modulo Phi_m is formed in the chain (DoubleCRT of
// both factors, product, toPoly): exact because its coefficients stay below half the chain product, which is checked here.
extern "C" int fhesi_ct_mul_poly_dev(fhesi_ctx* c, int32_t logQ, uint64_t* ct, int32_t nparts, int32_t nlimbs, int64_t count, const int64_t* poly_host, int32_t npoly) {
  CHECK_CTX(c);
  if (nparts < 1 || nlimbs < 1 || logQ < 1 || nlimbs * 64 < logQ) FHESI_FAIL("Ciphertext *= ZZX: coefficients of %d limbs cannot hold logQ=%d bits", nlimbs, logQ);
  if (npoly != 1 && npoly != count) FHESI_FAIL("Ciphertext *= ZZX: %d polynomials for %lld ciphertexts (one for all, or one each)", npoly, (long long)count);
  if (!count) return 0;
  const i64 n = c->phim;
  const int L = c->L;
  // |coefficient of the product modulo Phi_m| <= growth * n * 2^(logQ-1) * max|other_j|, growth = 1 (X^n + 1), 2 (the two-term folds of prime and
  // 2 x prime rings) or, conservatively, n for a general Phi_m
  u64 maxc = 0;
  for (i64 i = 0; i < (i64)npoly * n; ++i) { const i64 v = poly_host[i]; const u64 a = v < 0 ? (u64)(-(v + 1)) + 1 : (u64)v; if (a > maxc) maxc = a; }
  double bits = (logQ - 1) + std::log2((double)n) + (maxc ? std::log2((double)maxc) + 1e-9 : 0.0) + 1.0;
  bits += c->pow2 ? 0.0 : ((c->lin_q || hm::is_prime((u64)c->m)) ? 1.0 : std::log2((double)n));
  double chain = 0.0;
  for (int i = 0; i < L; ++i) chain += std::log2((double)c->q[i]);
  if (bits + 1.0 >= chain) FHESI_FAIL("Ciphertext *= ZZX: the product needs %.0f bits, the chain holds %.0f", bits + 1.0, chain);
  const std::vector<int> all = full_set(c);
  CrtTables* t;
  FHESI_TRY(get_crt_tables(c, all, &t));
  void *d_rows, *d_prow, *d_pl;
  FHESI_TRY(ws_reserve(c, 0, (size_t)count * nparts * L * n * 8, &d_rows));
  FHESI_TRY(ws_reserve(c, 3, (size_t)npoly * L * n * 8, &d_prow));
  FHESI_TRY(ws_reserve(c, 9, (size_t)npoly * n * 8, &d_pl));
  HIP_TRY(hipMemcpyAsync(d_pl, poly_host, (size_t)npoly * n * 8, hipMemcpyHostToDevice, c->stream));      // one signed limb per coefficient
  FHESI_TRY(launch_rns_reduce(c, (const u64*)d_pl, 1, n, npoly, 1, nullptr, (u64*)d_prow, L, nullptr));
  FHESI_TRY(row_fwd(c, (u64*)d_prow, npoly, L, nullptr, all.data()));
  FHESI_TRY(launch_rns_reduce(c, (const u64*)ct, nlimbs, n, count, nparts, nullptr, (u64*)d_rows, L, nullptr));
  FHESI_TRY(row_fwd(c, (u64*)d_rows, count * nparts, L, nullptr, all.data()));
  if (npoly == 1) {
    for (i64 done = 0; done < count * nparts; done += 65535) FHESI_TRY(launch_rows_mul_bcast(c, (u64*)d_rows + (size_t)done * L * n, (const u64*)d_rows + (size_t)done * L * n, (const u64*)d_prow, std::min<i64>(65535, count * nparts - done)));
  } else {
    for (i64 ci = 0; ci < count; ++ci) FHESI_TRY(launch_rows_mul_bcast(c, (u64*)d_rows + (size_t)ci * nparts * L * n, (const u64*)d_rows + (size_t)ci * nparts * L * n, (const u64*)d_prow + (size_t)ci * L * n, nparts));
  }
  FHESI_TRY(row_inv(c, (u64*)d_rows, count * nparts, L, nullptr, all.data()));
  FHESI_TRY(launch_crt(c, t, (const u64*)d_rows, L, nullptr, count * nparts, 2, 0, logQ, (u64*)ct, nlimbs));
  HIP_TRY(hipStreamSynchronize(c->stream));        // poly_host may be released on return
  return 0;
}
extern "C" int fhesi_rows_mul_long_dev(fhesi_ctx* c, uint64_t* rows, int64_t l, int64_t count) {
  CHECK_CTX(c);
  if (!count) return 0;
  std::vector<u64> sc(c->L);
  for (int i = 0; i < c->L; ++i) { const u64 q = c->q[i]; sc[i] = l >= 0 ? (u64)l % q : (q - ((u64)(-(l + 1)) + 1) % q) % q; }
  void* d_sc;
  FHESI_TRY(ws_reserve(c, 9, sizeof(u64) * 64, &d_sc));
  HIP_TRY(hipMemcpyAsync(d_sc, sc.data(), sizeof(u64) * c->L, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));        // sc lives on this stack frame
  return launch_ew_scalar(c, (u64*)rows, (const u64*)d_sc, count, c->L, nullptr, FHESI_OP_MUL);
}
extern "C" int fhesi_ct_gather_dev(fhesi_ctx* c, const uint64_t* pool, const int32_t* idx_host, int64_t count, int64_t words, uint64_t* out) {
  CHECK_CTX(c);
  if (!count) return 0;
  void* d_idx;
  FHESI_TRY(ws_reserve(c, 8, sizeof(int) * (size_t)count, &d_idx));
  HIP_TRY(hipMemcpyAsync(d_idx, idx_host, sizeof(int) * (size_t)count, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));        // the caller's index array may be reused as soon as we return
  return launch_gather(c, (const u64*)pool, (const int*)d_idx, count, words, (u64*)out);
}

// --------------------------------------------------------------------------------------------- Encrypt / Decrypt batches
// FHESIPubKey::Encrypt (FHE-SI.cpp:10-36) for `count` plaintexts; the randomness is the caller's (the reference draws it from NTL's
// PRNG): rand_host = [count][3][phi(m)] int64 = (r binary, e0, e1 Gaussian samples before the multiplication by p)
static int encrypt_batch_impl(fhesi_ctx* c, const fhesi_dcrt* pk0, const fhesi_dcrt* pk1, int32_t logQ, uint64_t p, const int64_t* rand_host, bool seeded, u64 seed, u64 first,
                              const int64_t* msg_host, int64_t count, uint64_t* out_dev, int32_t nlimbs) {
  CHECK_CTX(c);
  if (!pk0 || !pk1 || pk0->ctx != c || pk1->ctx != c) FHESI_FAIL("Encrypt: public key belongs to another context");
  if ((int)pk0->idx.size() != c->L || (int)pk1->idx.size() != c->L) FHESI_FAIL("Encrypt: public key must be defined over all primes");
  if (logQ < 1 || nlimbs * 64 < logQ) FHESI_FAIL("Encrypt: coefficients of %d limbs cannot hold logQ=%d bits", nlimbs, logQ);
  if (p < 2) FHESI_FAIL("Encrypt: plaintext modulus must be at least 2");
  if (!count) return 0;
  const i64 n = c->phim;
  const int L = c->L;
  const std::vector<int> all = full_set(c);
  void *d_small, *d_rows, *d_ct, *d_pk, *d_msg, *d_delta;
  FHESI_TRY(ws_reserve(c, 2, (size_t)count * 3 * n * 8, &d_small));
  FHESI_TRY(ws_reserve(c, 0, (size_t)count * 3 * L * n * 8, &d_rows));
  FHESI_TRY(ws_reserve(c, 1, (size_t)count * 2 * L * n * 8, &d_ct));
  FHESI_TRY(ws_reserve(c, 3, (size_t)2 * L * n * 8, &d_pk));
  FHESI_TRY(ws_reserve(c, 5, (size_t)count * n * 8, &d_msg));
  FHESI_TRY(ws_reserve(c, 4, (size_t)(nlimbs + 1) * 8, &d_delta));
  // delta = floor(2^logQ / p) (FHE-SI.cpp:31), nlimbs limbs
  std::vector<u64> delta(nlimbs, 0);
  { u128 rem = 0; for (int i = nlimbs - 1; i >= 0; --i) { const u64 limb = (i == logQ / 64) ? (1ull << (logQ % 64)) : 0; const u128 cur = (rem << 64) | limb; delta[i] = (u64)(cur / p); rem = cur % p; }
    if (logQ == 64 * nlimbs) { /* 2^logQ needs limb nlimbs: redo with the extra limb */ rem = 1; for (int i = nlimbs - 1; i >= 0; --i) { const u128 cur = rem << 64; delta[i] = (u64)(cur / p); rem = cur % p; } } }
  if (seeded) FHESI_TRY(launch_sample_encrypt(c, (i64*)d_small, count, seed, first));      // r, e0, e1 drawn in HBM (kernels_sample.hip)
  else HIP_TRY(hipMemcpyAsync(d_small, rand_host, (size_t)count * 3 * n * 8, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(d_msg, msg_host, (size_t)count * n * 8, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(d_delta, delta.data(), (size_t)nlimbs * 8, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(d_pk, pk0->d_rows, (size_t)L * n * 8, hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync((u64*)d_pk + (size_t)L * n, pk1->d_rows, (size_t)L * n * 8, hipMemcpyDeviceToDevice, c->stream));
  // DoubleCRT(r), DoubleCRT(e_i) * p: one-limb signed coefficients, the noise lifted by p (FHE-SI.cpp:19-25)
  const u64 lift[3] = {0, p, p};
  FHESI_TRY(launch_rns_reduce(c, (const u64*)d_small, 1, n, count, 3, lift, (u64*)d_rows, L, nullptr));
  FHESI_TRY(row_fwd(c, (u64*)d_rows, count * 3, L, nullptr, all.data()));
  FHESI_TRY(launch_encrypt_combine(c, (const u64*)d_rows, (const u64*)d_pk, count, (u64*)d_ct));          // ct[i] = pk[i]*r + e_i (:26-27)
  FHESI_TRY(row_inv(c, (u64*)d_ct, count * 2, L, nullptr, all.data()));                                    // toPoly (:28)
  CrtTables* t;
  FHESI_TRY(get_crt_tables(c, all, &t));
  FHESI_TRY(launch_crt(c, t, (const u64*)d_ct, L, nullptr, count * 2, 2, 0, logQ, (u64*)out_dev, nlimbs));
  FHESI_TRY(launch_add_scaled_msg(c, (u64*)out_dev, (const i64*)d_msg, (const u64*)d_delta, count, nlimbs, logQ));   // += delta*msg, Reduce (:31-35)
  HIP_TRY(hipStreamSynchronize(c->stream));      // the host arrays may be released on return
  return 0;
}

extern "C" int fhesi_encrypt_batch(fhesi_ctx* c, const fhesi_dcrt* pk0, const fhesi_dcrt* pk1, int32_t logQ, uint64_t p, const int64_t* rand_host,
                                   const int64_t* msg_host, int64_t count, uint64_t* out_dev, int32_t nlimbs) {
  if (!rand_host) FHESI_FAIL("Encrypt: null randomness (fhesi_encrypt_batch_seeded draws it on the device)");
  return encrypt_batch_impl(c, pk0, pk1, logQ, p, rand_host, false, 0, 0, msg_host, count, out_dev, nlimbs);
}
// ... with the randomness drawn on the device: plaintext i takes the streams of object index first_index + i (philox.h)
extern "C" int fhesi_encrypt_batch_seeded(fhesi_ctx* c, const fhesi_dcrt* pk0, const fhesi_dcrt* pk1, int32_t logQ, uint64_t p, uint64_t seed, uint64_t first_index,
                                          const int64_t* msg_host, int64_t count, uint64_t* out_dev, int32_t nlimbs) {
  return encrypt_batch_impl(c, pk0, pk1, logQ, p, nullptr, true, seed, first_index, msg_host, count, out_dev, nlimbs);
}

// FHESISecKey::Decrypt (FHE-SI.cpp:93-119) of `count` unscaled 2-part ciphertexts [count][2][phi(m)][nlimbs] in HBM
extern "C" int fhesi_decrypt_batch(fhesi_ctx* c, const fhesi_dcrt* sk1, int32_t logQ, uint64_t p, const uint64_t* ct_dev, int32_t nlimbs, int64_t count,
                                   int64_t* msg_host) {
  CHECK_CTX(c);
  if (!sk1 || sk1->ctx != c) FHESI_FAIL("Decrypt: secret key belongs to another context");
  if ((int)sk1->idx.size() != c->L) FHESI_FAIL("Decrypt: secret key must be defined over all primes");
  if (logQ < 1 || nlimbs < 1) FHESI_FAIL("Decrypt: bad shape");
  if (p < 2 || p >= (1ull << 62)) FHESI_FAIL("Decrypt: plaintext modulus out of range");
  if (!count) return 0;
  const i64 n = c->phim;
  const int L = c->L, nw = (logQ + 1 + 63) / 64;
  const std::vector<int> all = full_set(c);
  void *d_rows, *d_z, *d_big, *d_msg;
  FHESI_TRY(ws_reserve(c, 0, (size_t)count * 2 * L * n * 8, &d_rows));
  FHESI_TRY(ws_reserve(c, 1, (size_t)count * L * n * 8, &d_z));
  FHESI_TRY(ws_reserve(c, 2, (size_t)count * n * nw * 8, &d_big));
  FHESI_TRY(ws_reserve(c, 5, (size_t)count * n * 8, &d_msg));
  FHESI_TRY(launch_rns_reduce(c, (const u64*)ct_dev, nlimbs, n, count, 2, nullptr, (u64*)d_rows, L, nullptr));     // DoubleCRT(parts[i]) (:98-101)
  FHESI_TRY(row_fwd(c, (u64*)d_rows, count * 2, L, nullptr, all.data()));
  FHESI_TRY(launch_decrypt_dot(c, (const u64*)d_rows, sk1->d_rows, count, (u64*)d_z));                              // DotProduct with (1, t) (:105-107)
  FHESI_TRY(row_inv(c, (u64*)d_z, count, L, nullptr, all.data()));
  CrtTables* t;
  FHESI_TRY(get_crt_tables(c, all, &t));
  FHESI_TRY(launch_crt(c, t, (const u64*)d_z, L, nullptr, count, 0, 0, 0, (u64*)d_big, nw));                         // toPoly, low logQ+1 bits kept
  FHESI_TRY(launch_decrypt_round(c, (const u64*)d_big, count * n, nw, logQ, p, (i64*)d_msg));                        // round(p z / q) mod p (:110-116)
  HIP_TRY(hipMemcpyAsync(msg_host, d_msg, (size_t)count * n * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// KeySwitchSI::Init (FHE-SI.cpp:153-209) for all columns of a matrix at once; the randomness is the caller's, in the reference's draw order
static int keyswitch_init_impl(fhesi_ksk* k, const fhesi_dcrt* const* src, int32_t nsrc, const fhesi_dcrt* dst_t, int32_t logQ, int32_t decomp_bytes,
                               const uint64_t* a_host, int32_t nlimbs, const int64_t* err_host, bool seeded, u64 seed, u64 first) {
  if (!k) FHESI_FAIL("null key-switch matrix");
  fhesi_ctx* c = k->ctx;
  CHECK_CTX(c);
  if (nsrc != k->ncomp) FHESI_FAIL("KeySwitchSI::Init: the source key has %d components, the matrix was created for %d", nsrc, k->ncomp);
  if (decomp_bytes < 1 || decomp_bytes > 7) FHESI_FAIL("decompSize %d not supported", decomp_bytes);
  const int nd = (logQ + 8 * decomp_bytes - 1) / (8 * decomp_bytes);
  if (nd != k->ndigits) FHESI_FAIL("KeySwitchSI::Init: matrix has %d digits per component, context needs %d", k->ndigits, nd);
  if (nlimbs < 1 || nlimbs * 64 < logQ) FHESI_FAIL("KeySwitchSI::Init: random coefficients of %d limbs cannot hold logQ=%d bits", nlimbs, logQ);
  const int L = c->L;
  if (!dst_t || dst_t->ctx != c || dst_t->coeff_form || (int)dst_t->idx.size() != L) FHESI_FAIL("KeySwitchSI::Init: the target key must be a DoubleCRT over all primes of this context");
  for (int i = 0; i < nsrc; ++i)
    if (!src[i] || src[i]->ctx != c || src[i]->coeff_form || (int)src[i]->idx.size() != L) FHESI_FAIL("KeySwitchSI::Init: source key component %d must be a DoubleCRT over all primes of this context", i);
  const i64 n = c->phim, ncol = (i64)nsrc * nd;
  const int nlq = (logQ + 63) / 64;
  const std::vector<int> all = full_set(c);
  CrtTables* t;
  FHESI_TRY(get_crt_tables(c, all, &t));
  const int W = t->W;
  u64* d_b = k->d_rows;                                  // keySwitchMatrix[0] = b
  u64* d_A = k->d_rows + (size_t)ncol * L * n;           // keySwitchMatrix[1] = A
  // one-off call: its scratch is allocated here and released on return (the transforms below own the context's workspace slots)
  struct Scratch { std::vector<void*> p; ~Scratch() { hipDeviceSynchronize(); for (void* q : p) hipFree(q); } int get(size_t bytes, void** out) { if (hipMalloc(out, bytes ? bytes : 8) != hipSuccess) return 1; p.push_back(*out); return 0; } } scratch;
  void *d_s, *d_scoef, *d_in, *d_err, *d_bcoef, *d_comb;
  if (scratch.get((size_t)nsrc * L * n * 8, &d_s) || scratch.get((size_t)nsrc * n * W * 8, &d_scoef) || scratch.get((size_t)ncol * n * std::max(nlimbs, W) * 8, &d_in) ||
      scratch.get((size_t)ncol * n * 8, &d_err) || scratch.get((size_t)ncol * n * nlq * 8, &d_comb)) FHESI_FAIL("KeySwitchSI::Init: out of device memory");
  // sCoeff[i] = toPoly(s[i])   (:163-166)
  for (int i = 0; i < nsrc; ++i) HIP_TRY(hipMemcpyAsync((u64*)d_s + (size_t)i * L * n, src[i]->d_rows, (size_t)L * n * 8, hipMemcpyDeviceToDevice, c->stream));
  FHESI_TRY(row_inv(c, (u64*)d_s, nsrc, L, nullptr, all.data()));
  FHESI_TRY(launch_crt(c, t, (const u64*)d_s, L, nullptr, nsrc, 0, 0, 0, (u64*)d_scoef, W));
  // A[ind] = DoubleCRT(poly)   (:176-179)
  if (seeded) FHESI_TRY(launch_sample_keygen(c, (u64*)d_in, (i64*)d_err, ncol, nlimbs, logQ, seed, first));      // polynomials and errors drawn in HBM
  else {
    HIP_TRY(hipMemcpyAsync(d_in, a_host, (size_t)ncol * n * nlimbs * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_err, err_host, (size_t)ncol * n * 8, hipMemcpyHostToDevice, c->stream));
  }
  FHESI_TRY(launch_rns_reduce(c, (const u64*)d_in, nlimbs, n, ncol, 1, nullptr, d_A, L, nullptr));
  FHESI_TRY(row_fwd(c, d_A, ncol, L, nullptr, all.data()));
  // b[ind] = A[ind] * t; toPoly   (:180-187)
  FHESI_TRY(launch_rows_mul_bcast(c, d_b, d_A, dst_t->d_rows, ncol));
  FHESI_TRY(row_inv(c, d_b, ncol, L, nullptr, all.data()));
  d_bcoef = d_in;                                        // (the random coefficients are consumed)
  FHESI_TRY(launch_crt(c, t, d_b, L, nullptr, ncol, 0, 0, 0, (u64*)d_bcoef, W));
  // bCoeff += err + sCoeff[i] << (8 decompSize j); ReduceCoefficients; b[ind] = DoubleCRT(bCoeff)   (:189-204)
  FHESI_TRY(launch_keygen_combine(c, (const u64*)d_bcoef, W, (const u64*)d_scoef, W, (const i64*)d_err, ncol, nd, 8 * decomp_bytes, nlq, logQ, (u64*)d_comb));
  FHESI_TRY(launch_rns_reduce(c, (const u64*)d_comb, nlq, n, ncol, 1, nullptr, d_b, L, nullptr));
  FHESI_TRY(row_fwd(c, d_b, ncol, L, nullptr, all.data()));
  // A[ind] *= -1   (:181)
  FHESI_TRY(fhesi_rows_mul_long_dev(c, d_A, -1, ncol));
  k->aux_valid = false;
  HIP_TRY(hipStreamSynchronize(c->stream));             // the host arrays and the scratch may be released on return
  return 0;
}
extern "C" int fhesi_keyswitch_init_batch(fhesi_ksk* k, const fhesi_dcrt* const* src, int32_t nsrc, const fhesi_dcrt* dst_t, int32_t logQ, int32_t decomp_bytes,
                                          const uint64_t* a_host, int32_t nlimbs, const int64_t* err_host) {
  if (!a_host || !err_host) FHESI_FAIL("KeySwitchSI::Init: null randomness (fhesi_keyswitch_init_batch_seeded draws it on the device)");
  return keyswitch_init_impl(k, src, nsrc, dst_t, logQ, decomp_bytes, a_host, nlimbs, err_host, false, 0, 0);
}
// ... with the column randomness drawn on the device: column i takes the streams of object index first_index + i (philox.h)
extern "C" int fhesi_keyswitch_init_batch_seeded(fhesi_ksk* k, const fhesi_dcrt* const* src, int32_t nsrc, const fhesi_dcrt* dst_t, int32_t logQ, int32_t decomp_bytes,
                                                 uint64_t seed, uint64_t first_index) {
  return keyswitch_init_impl(k, src, nsrc, dst_t, logQ, decomp_bytes, nullptr, (logQ + 63) / 64, nullptr, true, seed, first_index);
}
// DoubleCRT::sampleHWt / sampleGaussian (DoubleCRT.h; NumbTh.cpp:340-404) with the polynomial drawn on the device: kind 0 = Hamming weight
// `param` with +-1 entries (the secret key, FHE-SI.cpp:90), kind 1 = rounded Gaussian with the context's stdev 3.2 (FHEContext.h:106)
extern "C" int fhesi_dcrt_sample(fhesi_dcrt* d, int32_t kind, int64_t param, uint64_t seed, uint64_t index) {
  if (!d) FHESI_FAIL("null DoubleCRT");
  fhesi_ctx* c = d->ctx;
  CHECK_CTX(c);
  if (d->coeff_form) FHESI_FAIL("sample: the object is a SingleCRT");
  if (kind == 0 && param < 0) FHESI_FAIL("sampleHWt: negative weight");
  const i64 n = c->phim;
  const int K = (int)d->idx.size();
  void* d_poly;
  FHESI_TRY(ws_reserve(c, 9, (size_t)n * 8, &d_poly));
  FHESI_TRY(launch_sample_poly(c, (i64*)d_poly, kind, param, seed, index));
  int* d_pos = nullptr;
  FHESI_TRY(upload_idx(c, d->idx, &d_pos));
  FHESI_TRY(launch_rns_reduce(c, (const u64*)d_poly, 1, n, 1, 1, nullptr, d->d_rows, K, d_pos));
  FHESI_TRY(row_fwd(c, d->d_rows, 1, K, d_pos, d->idx.data()));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// One wave of Matrix<Ciphertext> arithmetic followed by the key switch (see include/fhesi_hip.h).
// Every distinct operand of a chunk is brought to evaluation form ONCE (a matrix entry or a minor typically feeds many products),
// the products are formed and summed per group in one pass (tensor_sum_kernel), then the groups are key-switched together.
extern "C" int fhesi_ct_mul_sum_relin_dev(fhesi_ctx* c, const fhesi_ksk* k, int32_t logQ, uint64_t p, int32_t decomp_bytes, const uint64_t* pool,
                                          int32_t nlimbs, const int32_t* a_idx, const int32_t* b_idx, const int32_t* seg, int64_t ngroups, uint64_t* out) {
  CHECK_CTX(c);
  FHESI_TRY(key_switch_args(c, k, logQ, decomp_bytes, nlimbs));
  if (k->ncomp != 3) FHESI_FAIL("ct_mul_sum_relin needs the s^2 -> s matrix (3 source components), got %d", k->ncomp);
  if (!ngroups) return 0;
  for (i64 g = 0; g < ngroups; ++g) if (seg[g + 1] <= seg[g]) FHESI_FAIL("ct_mul_sum_relin: group %lld is empty", (long long)g);
  const i64 n = c->phim;
  const int L = c->L;
  const i64 ct_words = (i64)2 * n * nlimbs, tp_words = (i64)3 * L * n;
  const std::vector<int> all = full_set(c);
  CrtTables* t_all;
  FHESI_TRY(get_crt_tables(c, all, &t_all));
  const i64 chunk = batch_chunk(c, 3 * k->ndigits, ksaux_mode(c, t_all, k->ncomp * k->ndigits, 8 * decomp_bytes, logQ) == KS_MODE_LIMB32);           // groups per key-switch call
  // distinct operands per pass: bound their evaluation-form rows (2 L n words each) to about 4 GiB
  i64 ucap = (i64)(4.0 * 1024 * 1024 * 1024 / ((double)2 * L * n * 8));
  // the sums' integers over primes below 2^30 where that path applies (kernels_tensor32.hip; tProd is not visible from here either)
  i64 gmax = 1;
  for (i64 g = 0; g < ngroups; ++g) gmax = std::max<i64>(gmax, seg[g + 1] - seg[g]);
  const bool t32 = tensor32_sum_applies(c, p, nlimbs, logQ, gmax);
  if (t32) {
    FHESI_TRY(tensor32_sum_begin(c, p, nlimbs, logQ, gmax));
    ucap = (i64)(4.0 * 1024 * 1024 * 1024 / ((double)tensor32_sum_bytes(c, 1) / 3 * 2));
  }
  if (c->opt.wave_operands > 1) ucap = c->opt.wave_operands;
  if (ucap < 2) ucap = 2;
  const u64 lift[2] = {p, p};
  std::vector<int> ua, ub, sa, sb, lseg, host_idx;
  std::map<int, int> ma, mb;
  // one pass: terms [t0, t1) of the groups [g, g2) (group boundaries in gseg, relative to t0), summed into d_sum[0 .. g2-g)
  auto pass = [&](i64 t0, i64 t1, const std::vector<int>& gseg, bool accumulate, u64* d_sum) -> int {      // (d_sum: u32 rows on the 30-bit path)
    ua.clear(); ub.clear(); ma.clear(); mb.clear();
    sa.resize(t1 - t0); sb.resize(t1 - t0);
    for (i64 t = t0; t < t1; ++t) {
      auto ia = ma.find(a_idx[t]); if (ia == ma.end()) { ia = ma.emplace(a_idx[t], (int)ua.size()).first; ua.push_back(a_idx[t]); }
      auto ib = mb.find(b_idx[t]); if (ib == mb.end()) { ib = mb.emplace(b_idx[t], (int)ub.size()).first; ub.push_back(b_idx[t]); }
      sa[t - t0] = ia->second; sb[t - t0] = ib->second;
    }
    const i64 nua = (i64)ua.size(), nub = (i64)ub.size(), nt = t1 - t0, ng = (i64)gseg.size() - 1;
    void *d_ops, *d_rows = nullptr, *d_ix;
    FHESI_TRY(ws_reserve(c, 7, (size_t)(nua + nub) * ct_words * 8, &d_ops));
    if (!t32) FHESI_TRY(ws_reserve(c, 0, (size_t)(nua + nub) * 2 * L * n * 8, &d_rows));
    FHESI_TRY(ws_reserve(c, 5, sizeof(int) * (size_t)(nua + nub + 2 * nt + ng + 1), &d_ix));
    host_idx.clear();
    host_idx.insert(host_idx.end(), ua.begin(), ua.end());
    host_idx.insert(host_idx.end(), ub.begin(), ub.end());
    host_idx.insert(host_idx.end(), sa.begin(), sa.end());
    host_idx.insert(host_idx.end(), sb.begin(), sb.end());
    host_idx.insert(host_idx.end(), gseg.begin(), gseg.end());
    HIP_TRY(hipMemcpyAsync(d_ix, host_idx.data(), sizeof(int) * host_idx.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));                 // host_idx is reused by the next pass
    const int* dix = (const int*)d_ix;
    u64* d_a = (u64*)d_ops;
    u64* d_b = d_a + (size_t)nua * ct_words;
    u64* ca = (u64*)d_rows;
    u64* cb = ca + (size_t)nua * 2 * L * n;
    FHESI_TRY(launch_gather(c, (const u64*)pool, dix, nua + nub, ct_words, d_a));
    if (t32) return tensor32_sum_pass(c, d_a, nua, nub, dix + nua + nub, dix + nua + nub + nt, dix + nua + nub + 2 * nt, ng, nt, accumulate, d_sum);
    // c1 = DoubleCRT(parts * p), c2 = DoubleCRT(other.parts)   (Ciphertext.cpp:169-176)
    FHESI_TRY(launch_rns_reduce(c, d_a, nlimbs, n, nua, 2, lift, ca, L, nullptr));
    FHESI_TRY(launch_rns_reduce(c, d_b, nlimbs, n, nub, 2, nullptr, cb, L, nullptr));
    FHESI_TRY(row_fwd(c, ca, (nua + nub) * 2, L, nullptr, all.data()));
    return launch_tensor_sum(c, ca, cb, dix + nua + nub, dix + nua + nub + nt, dix + nua + nub + 2 * nt, ng, accumulate, d_sum, (double)nt);
  };
  std::vector<int> gseg;
  std::set<int> seen_a, seen_b;
  i64 g = 0;
  while (g < ngroups) {
    // groups g..g2-1: at most `chunk` of them and at most `ucap` distinct operands (one group is always taken)
    seen_a.clear(); seen_b.clear();
    i64 g2 = g;
    while (g2 < ngroups && g2 - g < chunk) {
      std::set<int> na = seen_a, nb = seen_b;
      for (i64 t = seg[g2]; t < seg[g2 + 1]; ++t) { na.insert(a_idx[t]); nb.insert(b_idx[t]); }
      if (g2 > g && (i64)(na.size() + nb.size()) > ucap) break;
      seen_a.swap(na); seen_b.swap(nb);
      ++g2;
    }
    const i64 ng = g2 - g;
    void* d_sum;
    FHESI_TRY(ws_reserve(c, 4, t32 ? tensor32_sum_bytes(c, ng) : (size_t)ng * tp_words * 8, &d_sum));
    if (ng == 1 && (i64)(seen_a.size() + seen_b.size()) > ucap) {
      // one group with more distinct operands than a pass holds: its terms are summed piecewise into the same accumulator
      const i64 step = ucap / 2;
      for (i64 t0 = seg[g]; t0 < seg[g + 1]; t0 += step) {
        const i64 t1 = std::min<i64>(t0 + step, seg[g + 1]);
        gseg = {0, (int)(t1 - t0)};
        FHESI_TRY(pass(t0, t1, gseg, t0 != seg[g], (u64*)d_sum));
      }
    } else {
      gseg.resize(ng + 1);
      for (i64 i = 0; i <= ng; ++i) gseg[i] = seg[g + i] - seg[g];
      FHESI_TRY(pass(seg[g], seg[g2], gseg, false, (u64*)d_sum));
    }
    if (t32) {
      void* d_parts;
      FHESI_TRY(ws_reserve(c, 2, (size_t)ng * 3 * ((logQ + 63) / 64) * n * 8, &d_parts));
      FHESI_TRY(tensor32_sum_finish(c, d_sum, ng, (u64*)d_parts));
      FHESI_TRY(key_switch_tail(c, k, logQ, decomp_bytes, (const u64*)d_parts, ng, nullptr, out + (size_t)g * ct_words, nlimbs));
    } else {
      FHESI_TRY(fhesi_apply_key_switch_dev(c, k, logQ, decomp_bytes, (const uint64_t*)d_sum, ng, out + (size_t)g * ct_words, nlimbs));
    }
    g = g2;
  }
  return 0;
}

// Ciphertexts per launch of the fused multiplication.  ks32: the key switch really runs over the four 30-bit auxiliary primes (ksaux_mode
// said KS_MODE_LIMB32) -- only then does the large-launch policy below apply; the 64-bit forms keep the smaller chunks measured for them.
static i64 batch_chunk(fhesi_ctx* c, int ncol, bool ks32) {
  if (c->opt.batch_chunk > 0) return c->opt.batch_chunk;
  i64 ch;
  double per;                                              // workspace bytes per ciphertext of a chunk (estimate)
  if (ks32) {
    // the 32-bit pipelines (key switch over the four auxiliary primes, tensor half over primes below 2^30): the larger the launch the
    // better -- 64 per launch 22.1 k mults/s, 128: 22.5 k, 512: 22.7 k, 1024: 23.0 k at the metric ring (more ciphertext tiles per key
    // block in the dot product, fewer launch tails; running the tensor half and the digit transforms in sub-chunks of 64 so that their
    // intermediate rows stay in the Infinity Cache, and only the dot product and what follows per chunk, measured 22.8 k against 23.3 k
    // for every stage per chunk) -- so: what fits about 48 GiB of workspace (digit rows ncol * 2 * row * 8 bytes per
    // ciphertext, the dot product's outputs and the tensor half's rows about 1.6 times that again), at most 1024
    per = (double)ncol * 2 * (double)aux32_row_len(c) * 8.0 * 2.6;
    ch = (i64)(48.0 * 1024 * 1024 * 1024 / per);
    ch = ch < 1 ? 1 : (ch > 1024 ? 1024 : ch);
  } else {
    // about 75k digit rows per chunk (150 rounds of the transform's 512 resident workgroups): measured best on MI355X at both the
    // metric ring (64 mults, 9.3 GiB of digit rows) and the stress ring (16-17 mults, 18 GiB) -- smaller chunks pay launch tails in
    // every stage, larger ones push the key rows out of the Infinity Cache during the dot product.  Capped at 32 GiB of digit rows.
    const double rows_per = (double)ncol * c->L, bytes_per = rows_per * c->phim * 8.0;
    ch = (i64)(76800.0 / rows_per);
    const i64 cap = (i64)(32.0 * 1024 * 1024 * 1024 / bytes_per);
    if (ch > cap) ch = cap;
    per = bytes_per + 5.0 * c->L * c->phim * 8.0;          // + tProd and the two scratch DoubleCRTs
  }
  // never more than the device can hold: what is free now plus what this lane's workspace already owns, minus a margin (the other lane of
  // option lanes = 2 keeps a second set, hence half of the free memory each)
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
    double have = (double)free_b / (c->opt.lanes >= 2 ? 2.0 : 1.0);
    for (int i = 0; i < FHESI_WS_SLOTS; ++i) have += (double)c->ws_bytes[i];
    have -= 2.0 * 1024 * 1024 * 1024;
    const i64 fit = have > per ? (i64)(have / (per * 1.15)) : 1;
    if (ch > fit) ch = fit;
  } else (void)hipGetLastError();
  return ch < 1 ? 1 : ch;
}

static int mul_relin_chunks(fhesi_ctx* c, const fhesi_ksk* k, int32_t logQ, uint64_t p, int32_t decomp_bytes, const uint64_t* a, const uint64_t* b,
                            uint64_t* out, int32_t nlimbs, int64_t count) {
  const i64 n = c->phim;
  const int L = c->L;
  CrtTables* t;
  FHESI_TRY(get_crt_tables(c, full_set(c), &t));
  const int ks_mode = ksaux_mode(c, t, k->ncomp * k->ndigits, 8 * decomp_bytes, logQ);
  i64 chunk = batch_chunk(c, 3 * k->ndigits, ks_mode == KS_MODE_LIMB32);
  for (i64 done = 0; done < count;) {
    const i64 cnt = std::min(chunk, count - done);
    const size_t off = (size_t)done * 2 * n * nlimbs;
    c->ws_oom = false;
    int rc;
    if (k->ncomp == 3 && tensor32_applies(c, p, nlimbs, logQ)) {
      // tProd is not visible from here: its integers are formed over primes below 2^30 (kernels_tensor32.hip), straight to the scaled-down parts
      FHESI_TRY(key_switch_args(c, k, logQ, decomp_bytes, nlimbs));
      void* d_parts = nullptr;
      rc = ws_reserve(c, 2, (size_t)cnt * 3 * ((logQ + 63) / 64) * n * 8, &d_parts);
      if (!rc) rc = launch_tensor32(c, p, a + off, b + off, nlimbs, logQ, cnt, (u64*)d_parts);
      if (!rc) rc = key_switch_tail(c, k, logQ, decomp_bytes, (const u64*)d_parts, cnt, nullptr, out + off, nlimbs);
    } else {
      void* d_tp = nullptr;
      rc = ws_reserve(c, 5, (size_t)cnt * 3 * L * n * 8, &d_tp);
      if (!rc) rc = fhesi_ct_mul_dev(c, p, a + off, b + off, nlimbs, cnt, (uint64_t*)d_tp);
      if (!rc) rc = key_switch_args(c, k, logQ, decomp_bytes, nlimbs);
      if (!rc) rc = apply_key_switch_consume(c, k, logQ, decomp_bytes, (u64*)d_tp, cnt, out + off, nlimbs);      // the chunk's tProd is ours: no copy
    }
    if (rc) {
      // a workspace allocation failed (a device with less free memory than the estimate assumed): the chunk is redone at half the size --
      // every stage of a chunk writes only workspace and its own slice of `out`, so nothing of the failed attempt survives
      if (c->ws_oom && cnt > 1) { chunk = (cnt + 1) / 2; continue; }
      return rc;
    }
    done += cnt;
  }
  return 0;
}
static void swap_lane(fhesi_ctx* c) {
  std::swap(c->stream, c->lane_stream);
  for (int i = 0; i < FHESI_WS_SLOTS; ++i) { std::swap(c->ws[i], c->lane_ws[i]); std::swap(c->ws_bytes[i], c->lane_ws_bytes[i]); }
}

extern "C" int fhesi_ct_mul_relin_batch_dev(fhesi_ctx* c, const fhesi_ksk* k, int32_t logQ, uint64_t p, int32_t decomp_bytes, const uint64_t* a,
                                            const uint64_t* b, uint64_t* out, int32_t nlimbs, int64_t count) {
  CHECK_CTX(c);
  if (!k || k->ctx != c) FHESI_FAIL("KeySwitchSI: context mismatch");
  if (k->ncomp != 3) FHESI_FAIL("ct_mul_relin needs the s^2 -> s matrix (3 source components), got %d", k->ncomp);
  if (nlimbs * 64 < logQ) FHESI_FAIL("coefficients of %d limbs cannot hold logQ=%d bits", nlimbs, logQ);
  const int lanes = c->opt.lanes;    // 2: two concurrent half-batches (+4 % with launches of 64 ciphertexts, +0.6 % with launches of 1024; kernels of the halves time-share the GPU)
  if (lanes < 2 || count < 8 || !c->pow2) return mul_relin_chunks(c, k, logQ, p, decomp_bytes, a, b, out, nlimbs, count);
  // two lanes: the second half of the batch runs on a second stream with its own workspace.  Ciphertexts are independent, so
  // the halves never touch the same memory; the fork / join events keep the call's stream semantics (work is ordered after
  // what was enqueued on the context's stream before the call, and fhesi_ctx_sync covers both halves afterwards).
  const int64_t h0 = (count + 1) / 2, h1 = count - h0;
  const size_t off = (size_t)h0 * 2 * c->phim * nlimbs;
  const int stagger = c->opt.stagger;   // measured slower than starting both lanes together
  HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
  HIP_TRY(hipStreamWaitEvent(c->lane_stream, c->ev_fork, 0));
  c->mark_mid = stagger != 0;
  int r = mul_relin_chunks(c, k, logQ, p, decomp_bytes, a, b, out, nlimbs, h0);
  if (!r) {
    // stagger: the second lane starts when the first lane's digit NTT (VALU-bound) has been issued, so that its own NTT
    // runs against the first lane's HBM-bound dot product / CRT tail instead of in lockstep with its NTT
    if (stagger) HIP_TRY(hipStreamWaitEvent(c->lane_stream, c->ev_mid, 0));
    swap_lane(c);
    r = mul_relin_chunks(c, k, logQ, p, decomp_bytes, a + off, b + off, out + off, nlimbs, h1);
    hipEventRecord(c->ev_join, c->stream);        // (c->stream is the lane stream here)
    swap_lane(c);
    if (!r) HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join, 0));
  }
  return r;
}

extern "C" int fhesi_ct_mul_relin_batch(fhesi_ctx* c, const fhesi_ksk* k, int32_t logQ, uint64_t p, int32_t decomp_bytes, const uint64_t* a,
                                        const uint64_t* b, uint64_t* out, int32_t nlimbs, int64_t count) {
  CHECK_CTX(c);
  if (!count) return 0;
  const size_t bytes = (size_t)count * 2 * c->phim * nlimbs * 8;
  // staging for the two operand batches and the results: one grow-only workspace slot (a hipMalloc / hipFree pair per call costs
  // milliseconds for a single ciphertext and seconds for gigabytes -- object-at-a-time callers of the class surface come through here)
  void* stage;
  FHESI_TRY(ws_reserve(c, 11, 3 * bytes, &stage));
  u64 *da = (u64*)stage, *db = (u64*)((char*)stage + bytes), *dout = (u64*)((char*)stage + 2 * bytes);
  int r = 0;
  if (hipMemcpyAsync(da, a, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess || hipMemcpyAsync(db, b, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
    fhesi_set_error("upload of ciphertext batch failed"); r = 1;
  }
  if (!r) r = fhesi_ct_mul_relin_batch_dev(c, k, logQ, p, decomp_bytes, da, db, dout, nlimbs, count);
  if (!r && hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { fhesi_set_error("download of ciphertext batch failed"); r = 1; }
  hipStreamSynchronize(c->stream);
  return r;
}
