// capi_ct.hip -- ciphertext algebra between multiplications, Encrypt / Decrypt batches, key generation (include/fhesi_hip.h)
#include "capi_common.h"

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
                               const uint64_t* a_host, int32_t nlimbs, const int64_t* err_host, bool seeded, u64 seed, u64 pub_seed, u64 first) {
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
  if (seeded) FHESI_TRY(launch_sample_keygen(c, (u64*)d_in, (i64*)d_err, ncol, nlimbs, logQ, seed, pub_seed, first));      // polynomials and errors drawn in HBM
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
  return keyswitch_init_impl(k, src, nsrc, dst_t, logQ, decomp_bytes, a_host, nlimbs, err_host, false, 0, 0, 0);
}
// ... with the column randomness drawn on the device: column i takes the streams of object index first_index + i (philox.h); the public
// polynomials a draw from public_seed, the secret errors from seed
extern "C" int fhesi_keyswitch_init_batch_seeded(fhesi_ksk* k, const fhesi_dcrt* const* src, int32_t nsrc, const fhesi_dcrt* dst_t, int32_t logQ, int32_t decomp_bytes,
                                                 uint64_t seed, uint64_t public_seed, uint64_t first_index) {
  if (public_seed == seed) FHESI_FAIL("KeySwitchSI::Init (seeded): public_seed must differ from the secret seed (the polynomials a are public, the errors are not)");
  return keyswitch_init_impl(k, src, nsrc, dst_t, logQ, decomp_bytes, nullptr, (logQ + 63) / 64, nullptr, true, seed, public_seed, first_index);
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

