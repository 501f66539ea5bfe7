// mock_abi.cpp -- TEST HARNESS: a host-memory stand-in for the handful of C-ABI entry points the recording evaluator of the mirror's
// Ciphertext (fhe-si_amd/host/fhesi_engine.h) calls, so that its HOST logic -- the arena's run allocator, growth, the sharing of equal
// operations, dependency levelling, shard bounds, lifetimes in the graph of shared values -- runs on a machine without a GPU, under
// AddressSanitizer / UBSan (tests/test_engine_host_logic.py).  The "ciphertext arithmetic" is a toy on 64-bit words chosen so that every
// operation is distinguishable and order-sensitive mistakes show; it has nothing to do with the scheme.  Never linked into the product.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/fhesi_hip.h"

struct fhesi_ctx { int64_t m, phim; int live_allocs = 0; };
struct fhesi_ksk { uint64_t tag; };
static const char* g_err = "";
extern "C" {
const char* fhesi_last_error(void) { return g_err; }
int32_t fhesi_abi_version(void) { return FHESI_ABI_VERSION; }
int fhesi_ctx_create(fhesi_ctx** out, int64_t m, int32_t, const uint64_t*, const uint64_t*, int32_t) {
  int64_t phi = 0; for (int64_t i = 1; i < m; ++i) { int64_t a = i, b = m; while (b) { int64_t t = a % b; a = b; b = t; } if (a == 1) ++phi; }
  *out = new fhesi_ctx{m, phi}; return 0;
}
int fhesi_ctx_destroy(fhesi_ctx* c) { if (c->live_allocs) { g_err = "mock: device buffers alive at context destruction"; return 1; } delete c; return 0; }
int64_t fhesi_ctx_phim(const fhesi_ctx* c) { return c->phim; }
int fhesi_ctx_zms_idx(const fhesi_ctx* c, int32_t* out) { int k = 0; for (int64_t i = 0; i < c->m; ++i) { int64_t a = i, b = c->m; while (b) { int64_t t = a % b; a = b; b = t; } out[i] = a == 1 ? k++ : -1; } return 0; }
int fhesi_ctx_phi_m(const fhesi_ctx* c, int64_t* out) { for (int64_t i = 0; i <= c->phim; ++i) out[i] = 0; out[0] = 1; out[c->phim] = 1; return 0; }   // (X^phi + 1: exact for powers of two, enough here)
int fhesi_ctx_sync(fhesi_ctx*) { return 0; }
int fhesi_ctx_copy_options(fhesi_ctx*, const fhesi_ctx*) { return 0; }
int fhesi_dev_alloc(fhesi_ctx* c, size_t bytes, void** out) { *out = std::malloc(bytes ? bytes : 1); ++c->live_allocs; return *out ? 0 : 1; }
int fhesi_dev_free(fhesi_ctx* c, void* p) { std::free(p); --c->live_allocs; return 0; }
int fhesi_dev_upload(fhesi_ctx*, void* d, const void* s, size_t n) { std::memcpy(d, s, n); return 0; }
int fhesi_dev_download(fhesi_ctx*, void* d, const void* s, size_t n) { std::memcpy(d, s, n); return 0; }
int fhesi_dev_copy(fhesi_ctx*, void* d, const void* s, size_t n) { std::memmove(d, s, n); return 0; }
int fhesi_ksk_create(fhesi_ctx*, int32_t ncomp, int32_t nd, fhesi_ksk** out) { static uint64_t next = 1; *out = new fhesi_ksk{0x9e3779b97f4a7c15ull * next++ + (uint64_t)ncomp * 131 + nd}; return 0; }
int fhesi_ksk_free(fhesi_ksk* k) { delete k; return 0; }
// the toy arithmetic (the same formulas are in tests/host/test_engine_cpu.cpp's direct evaluator)
static size_t words_of(const fhesi_ctx* c, int32_t nl) { return (size_t)2 * c->phim * nl; }
int fhesi_ct_gather_dev(fhesi_ctx*, const uint64_t* pool, const int32_t* idx, int64_t count, int64_t words, uint64_t* out) {
  std::vector<uint64_t> tmp((size_t)count * words);                  // (out may overlap the pool)
  for (int64_t i = 0; i < count; ++i) std::memcpy(&tmp[i * words], pool + (size_t)idx[i] * words, words * 8);
  std::memcpy(out, tmp.data(), tmp.size() * 8); return 0;
}
int fhesi_ct_mul_sum_relin_dev(fhesi_ctx* c, const fhesi_ksk* k, int32_t, uint64_t p, int32_t, const uint64_t* pool, int32_t nl, const int32_t* a, const int32_t* b, const int32_t* seg,
                               int64_t ng, uint64_t* out) {
  const size_t W = words_of(c, nl);
  std::vector<uint64_t> res((size_t)ng * W, 0);
  for (int64_t g = 0; g < ng; ++g) {
    if (seg[g + 1] <= seg[g]) { g_err = "mock: empty group"; return 1; }
    for (int32_t t = seg[g]; t < seg[g + 1]; ++t) for (size_t w = 0; w < W; ++w) res[g * W + w] += (pool[(size_t)a[t] * W + w] * p + 1) * (pool[(size_t)b[t] * W + w] ^ (uint64_t)w);
    for (size_t w = 0; w < W; ++w) res[g * W + w] = res[g * W + w] * 3 + k->tag;
  }
  std::memcpy(out, res.data(), res.size() * 8); return 0;
}
int fhesi_ct_automorph_key_switch_dev(fhesi_ctx* c, const fhesi_ksk* k, int32_t, int32_t, int64_t kk, const uint64_t* in, int32_t nl, int64_t count, uint64_t* out, int32_t) {
  const size_t W = words_of(c, nl); std::vector<uint64_t> res((size_t)count * W);
  for (int64_t i = 0; i < count; ++i) for (size_t w = 0; w < W; ++w) res[i * W + w] = in[i * W + (w + (size_t)kk) % W] * 5 + k->tag + (uint64_t)kk;
  std::memcpy(out, res.data(), res.size() * 8); return 0;
}
int fhesi_ct_automorph_dev(fhesi_ctx* c, int64_t kk, const uint64_t* in, int32_t, int32_t nl, int64_t count, uint64_t* out, int32_t) {
  const size_t W = words_of(c, nl); std::vector<uint64_t> res((size_t)count * W);
  for (int64_t i = 0; i < count; ++i) for (size_t w = 0; w < W; ++w) res[i * W + w] = in[i * W + (w + (size_t)kk) % W] + 7;
  std::memcpy(out, res.data(), res.size() * 8); return 0;
}
int fhesi_ct_add_dev(fhesi_ctx* c, int32_t, uint64_t* dst, const uint64_t* src, int32_t, int32_t nl, int64_t count) { const size_t W = words_of(c, nl); for (size_t w = 0; w < (size_t)count * W; ++w) dst[w] += src[w] * 2; return 0; }   // (not commutative on purpose)
int fhesi_ct_mul_long_dev(fhesi_ctx* c, int32_t, uint64_t* ct, int64_t l, int32_t, int32_t nl, int64_t count) { const size_t W = words_of(c, nl); for (size_t w = 0; w < (size_t)count * W; ++w) ct[w] = ct[w] * (uint64_t)l + 11; return 0; }
// never reached on one rank, present for the linker
int fhesi_comm_init_all(int32_t, const int32_t*, fhesi_comm**) { g_err = "mock: no communicator"; return 1; }
int fhesi_comm_destroy(fhesi_comm*) { return 0; }
int fhesi_ksk_broadcast(fhesi_ksk*, fhesi_comm*, int32_t) { return 1; }
int fhesi_comm_broadcast_dev(fhesi_ctx*, fhesi_comm*, void*, size_t, int32_t) { return 1; }
int fhesi_comm_exchange(fhesi_ctx*, fhesi_comm*, uint64_t*, const int64_t*) { return 1; }
int fhesi_comm_exchange_begin(fhesi_ctx*, fhesi_comm*, uint64_t*, const int64_t*) { return 1; }
int fhesi_comm_exchange_end(fhesi_ctx*, fhesi_comm*) { return 1; }
}
