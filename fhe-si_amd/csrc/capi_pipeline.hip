// capi_pipeline.hip -- the key-switch matrix and the fused ciphertext multiplication + key switch, per stage and per batch (include/fhesi_hip.h)
#include "capi_common.h"
#include "copy_pool.h"

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

extern "C" int fhesi_ksk_key_bits(const fhesi_ksk* k, int32_t* centred, int32_t* key_bits) {
  if (!k) FHESI_FAIL("null key-switch matrix");
  if (centred) *centred = k->aux_valid && k->aux_centred ? 1 : 0;
  if (key_bits) *key_bits = k->aux_valid ? k->aux_key_bits : 0;
  return 0;
}

// --------------------------------------------------------------------------------------------- ciphertext pipeline
static i64 batch_chunk(fhesi_ctx* c, int ncol, bool ks32, i64 count = -1);
static int mul_relin_chunks(fhesi_ctx* c, const fhesi_ksk* k, int32_t logQ, uint64_t p, int32_t decomp_bytes, const uint64_t* a, const uint64_t* b, uint64_t* out, int32_t nlimbs, int64_t count);

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
// parts_wm: d_parts are 32-bit word rows (launch_tensor32 parts_wm) -- only the four-prime limb form reads those; the caller asks for them
// only when ksaux_mode says that form runs.
static int key_switch_tail(fhesi_ctx* c, const fhesi_ksk* k, int32_t logQ, int32_t decomp_bytes, const u64* d_parts, int64_t count, u64* d_t,
                           uint64_t* out, int32_t nlimbs, bool parts_wm = false) {
  const i64 n = c->phim;
  const int L = c->L, ncomp = k->ncomp, nd = k->ndigits, ncol = ncomp * nd, nlq = (logQ + 63) / 64;
  const std::vector<int> all = full_set(c);
  CrtTables* t;
  FHESI_TRY(get_crt_tables(c, all, &t));
  // Dot product through the two largest chain primes (kernels_ksaux.hip): 2 transforms per digit polynomial instead of L.
  // option ks_direct keeps the per-prime dot product below (A/B measurements; also the path of every shape the other does not cover).
  const int ks_mode = ksaux_mode(c, t, ncol, 8 * decomp_bytes, logQ);
  const_cast<fhesi_ksk*>(k)->last_form = ks_mode;
  if (parts_wm && ks_mode != KS_MODE_LIMB32) FHESI_FAIL("key switch: word-major parts handed to a form that reads limb rows (internal)");
  if (ks_mode != KS_MODE_DIRECT) {
    fhesi_ksk* km = const_cast<fhesi_ksk*>(k);
    if (!k->aux_valid || k->aux_mode != ks_mode || k->aux_suborder != ntt_digits_suborder(c, 8 * decomp_bytes) || k->aux_logQ != logQ || k->aux_long_opt != c->opt.ks_long_keys) FHESI_TRY(ksaux_build(c, km, 8 * decomp_bytes, logQ, ks_mode));
    const int R = k->aux_rows;        // L chain-prime residues, or the limbs of the key's integer coefficients (limb mode)
    const i64 nrow = k->aux32 ? aux32_row_len(c) : n;      // the 32-bit auxiliary rows always have 2^14 elements (four 4-byte residues = two 8-byte ones)
    void *d_dig, *d_o;
    FHESI_TRY(ws_reserve(c, 0, (size_t)count * ncol * 2 * nrow * 8, &d_dig));
    FHESI_TRY(ws_reserve(c, 10, (size_t)count * 2 * R * 2 * nrow * 8, &d_o));
    if (parts_wm && !k->aux32) FHESI_FAIL("key switch: word-major parts without the four-prime table (internal)");
    if (k->aux32) {       // four 30-bit primes (kernels_aux32.hip): the same buffer sizes, u32 rows
      FHESI_TRY(launch_ntt32_fwd_digits(c, d_parts, nlq, 8 * decomp_bytes, nd, count * ncomp, (u32*)d_dig, kDigitSubCt * ncol, parts_wm));
      if (c->mark_mid) { HIP_TRY(hipEventRecord(c->ev_mid, c->stream)); c->mark_mid = false; }
      bool mont = true;                 // dot32_kernel2 leaves the factor 2^-32 of its Montgomery step; the matrix-core form does not
      FHESI_TRY(launch_dot32(c, km, (const u32*)d_dig, ncol, count, (u32*)d_o, &mont));
      const bool fold_tail = ks_recombine_takes_tail(c, t, k);      // (rows of 2^15 at the stress chain: the inverse's tail stage runs in the recombination's loader)
      FHESI_TRY(launch_ntt32_inv(c, (u32*)d_o, count * 2 * R, 4, 0, mont, !fold_tail));
      return launch_ks_recombine(c, t, k, (const u64*)d_o, count * 2, (u64*)out, nlimbs, fold_tail);
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
  {
    // every group a single product (a loop of `c *= d; ApplyKeySwitch(c)` recorded by the host mirror, fhesi_engine.h): the operands are
    // gathered into two batches and take the batch pipeline of fhesi_ct_mul_relin_batch_dev -- the tensor products formed in the loader
    // of the inverse transform instead of the sum kernels below (19.6 k -> 22 k multiplications per second at the metric ring with the
    // operands gathered, more with them addressed through the indices)
    bool single = c->opt.wave_single != 0;
    for (i64 g = 0; single && g < ngroups; ++g) single = seg[g + 1] - seg[g] == 1;
    if (single) {
      const i64 step = 1024;
      std::vector<int> ix;
      for (i64 g0 = 0; g0 < ngroups; g0 += step) {
        const i64 cnt = std::min<i64>(step, ngroups - g0);
        const size_t ops_bytes = (size_t)2 * cnt * ct_words * 8;
        // where the tensor half runs over the 30-bit primes its first kernel takes the operands through the indices (no copy at all);
        // elsewhere they are gathered into two batches first
        const bool indexed = tensor32_applies(c, p, nlimbs, logQ);
        void* d_ops;
        FHESI_TRY(ws_reserve(c, 12, (indexed ? 0 : ops_bytes) + sizeof(int) * 2 * (size_t)cnt, &d_ops));      // (a slot of its own: the pipeline called below uses most of the others)
        int* d_ix = (int*)((char*)d_ops + (indexed ? 0 : ops_bytes));
        ix.resize(2 * cnt);
        for (i64 g = 0; g < cnt; ++g) { ix[g] = a_idx[seg[g0 + g]]; ix[cnt + g] = b_idx[seg[g0 + g]]; }
        HIP_TRY(hipMemcpyAsync(d_ix, ix.data(), sizeof(int) * ix.size(), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));             // ix is reused by the next chunk
        if (indexed) {
          c->op_idx = d_ix; c->op_idx_n = cnt; c->op_idx_done = 0;
          const int rc = mul_relin_chunks(c, k, logQ, p, decomp_bytes, (const u64*)pool, (const u64*)pool, out + (size_t)g0 * ct_words, nlimbs, cnt);
          c->op_idx = nullptr;
          if (rc) return rc;
        } else {
          FHESI_TRY(launch_gather(c, (const u64*)pool, d_ix, 2 * cnt, ct_words, (u64*)d_ops));
          FHESI_TRY(fhesi_ct_mul_relin_batch_dev(c, k, logQ, p, decomp_bytes, (const u64*)d_ops, (const u64*)d_ops + (size_t)cnt * ct_words, out + (size_t)g0 * ct_words, nlimbs, cnt));
        }
      }
      return 0;
    }
  }
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
      const bool wm = c->opt.parts_words && ksaux_mode(c, t_all, k->ncomp * k->ndigits, 8 * decomp_bytes, logQ) == KS_MODE_LIMB32;
      FHESI_TRY(tensor32_sum_finish(c, d_sum, ng, (u64*)d_parts, wm));
      FHESI_TRY(key_switch_tail(c, k, logQ, decomp_bytes, (const u64*)d_parts, ng, nullptr, out + (size_t)g * ct_words, nlimbs, wm));
    } else {
      FHESI_TRY(fhesi_apply_key_switch_dev(c, k, logQ, decomp_bytes, (const uint64_t*)d_sum, ng, out + (size_t)g * ct_words, nlimbs));
    }
    g = g2;
  }
  return 0;
}

// Ciphertexts per launch of the fused multiplication.  ks32: the key switch really runs over the four 30-bit auxiliary primes (ksaux_mode
// said KS_MODE_LIMB32) -- only then does the large-launch policy below apply; the 64-bit forms keep the smaller chunks measured for them.
static i64 batch_chunk(fhesi_ctx* c, int ncol, bool ks32, i64 count) {
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
  // (a call whose whole batch needs less than a gigabyte skips the query: hipMemGetInfo costs tens of microseconds, a quarter of the issue
  // time of a single multiplication; should the allocation fail after all, the chunk is halved and retried like any other)
  size_t free_b = 0, total_b = 0;
  if (count >= 0 && (double)count * per * 1.15 < 1024.0 * 1024 * 1024) return std::min<i64>(ch, std::max<i64>(count, 1));
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
  i64 chunk = batch_chunk(c, 3 * k->ndigits, ks_mode == KS_MODE_LIMB32, count);
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
      if (c->op_idx) c->op_idx_done = done;                  // indexed operands: the chunk is a window of the index arrays, the buffer stays put
      const bool wm = ks_mode == KS_MODE_LIMB32 && c->opt.parts_words;      // (the digit loader of the four-prime form reads word rows in whole lines)
      if (!rc) rc = launch_tensor32(c, p, c->op_idx ? a : a + off, c->op_idx ? b : b + off, nlimbs, logQ, cnt, (u64*)d_parts, wm);
      if (!rc) rc = key_switch_tail(c, k, logQ, decomp_bytes, (const u64*)d_parts, cnt, nullptr, out + off, nlimbs, wm);
    } else {
      if (c->op_idx) FHESI_FAIL("ct_mul_relin: indexed operands reached the chain path");      // (only the 30-bit tensor half reads through indices; the caller checks tensor32_applies)
      void* d_tp = nullptr;
      rc = ws_reserve(c, 5, (size_t)cnt * 3 * L * n * 8, &d_tp);
      if (!rc) rc = fhesi_ct_mul_dev(c, p, a + off, b + off, nlimbs, cnt, (uint64_t*)d_tp);
      if (!rc) rc = key_switch_args(c, k, logQ, decomp_bytes, nlimbs);
      if (!rc) rc = apply_key_switch_consume(c, k, logQ, decomp_bytes, (u64*)d_tp, cnt, out + off, nlimbs);      // the chunk's tProd is ours: no copy
    }
    if (rc) {
      // a workspace allocation failed (a device with less free memory than the estimate assumed): the chunk is redone at half the size --
      // every stage of a chunk writes only workspace and its own slice of `out`, so nothing of the failed attempt survives
      // The grow-only slots that already grew for the oversized attempt are released first: left in place they would compete with the smaller
      // attempt for the same memory and the halving could run down to one ciphertext on a device that fits a mid-sized chunk.
      if (c->ws_oom && cnt > 1) {
        hipStreamSynchronize(c->stream);
        for (int slot : {0, 1, 2, 5, 10}) if (c->ws[slot]) { hipFree(c->ws[slot]); c->ws[slot] = nullptr; c->ws_bytes[slot] = 0; }
        (void)hipGetLastError();
        fhesi_set_error("%s", "");
        chunk = (cnt + 1) / 2;
        continue;
      }
      c->ws_oom = false;
      return rc;
    }
    done += cnt;
  }
  c->ws_oom = false;
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

// ---------------------------------------------------------------------------------------------- host buffers
// fhesi_ct_mul_relin_batch takes and returns HOST buffers (what a caller of the class surface holds: Test_AddMul.cpp:59-67).  6 MiB cross the
// bus per multiplication at the metric ring (two operands in, one result out), and a pageable hipMemcpy moves them through the runtime's
// own bounce buffer on one thread, strictly before and after the compute: 2.9 k mults/s where the device does 24 k and the bus allows ~12 k.
// Here the batch runs as a pipeline of stages of `host_chunk` ciphertexts over a ring of two PINNED slots:
//     caller's a, b --(copy threads)--> pinned slot --(stream `up`, DMA)--> device slot --(context stream: the fused pipeline)-->
//     device slot --(stream `down`, DMA)--> pinned slot --(copy threads)--> caller's out
// so the upload of stage i + 1, the compute of stage i and the download of stage i - 1 overlap, and the pageable side is copied by several
// threads.  Buffers the caller allocated pinned (fhesi_host_alloc, or any hipHostMalloc / registered memory) skip the copy threads: the DMA
// reads and writes them directly.  Results are those of fhesi_ct_mul_relin_batch_dev bit for bit (the same calls on the same values).
struct HostStage {
  static constexpr int NS = 2;
  size_t slot_bytes = 0;                 // bytes per operand per slot
  void* pin[NS][3] = {};                 // a, b, out
  void* dev[NS][3] = {};
  hipStream_t up = nullptr, down = nullptr;
  hipEvent_t ev_up[NS] = {}, ev_comp[NS] = {}, ev_down[NS] = {};
  std::vector<hipEvent_t> ev_piece[NS];  // one per downloaded piece of a stage (pageable results)
  CopyPool pool;                         // copy threads between pageable memory and the ring (copy_pool.h)
  void copy(void* d, const void* s, size_t n) { pool.copy(d, s, n); }
  void release() {
    pool.stop();
    for (int s = 0; s < NS; ++s) for (int k = 0; k < 3; ++k) { if (pin[s][k]) hipHostFree(pin[s][k]); if (dev[s][k]) hipFree(dev[s][k]); pin[s][k] = dev[s][k] = nullptr; }
    for (int s = 0; s < NS; ++s) { if (ev_up[s]) hipEventDestroy(ev_up[s]); if (ev_comp[s]) hipEventDestroy(ev_comp[s]); if (ev_down[s]) hipEventDestroy(ev_down[s]); for (hipEvent_t e : ev_piece[s]) hipEventDestroy(e); ev_piece[s].clear(); }
    if (up) hipStreamDestroy(up);
    if (down) hipStreamDestroy(down);
  }
};
void host_stage_free(fhesi_ctx* c) {
  if (!c->host_stage) return;
  c->host_stage->release();
  delete c->host_stage;
  c->host_stage = nullptr;
}
static int host_stage_get(fhesi_ctx* c, size_t slot_bytes, HostStage** out) {
  HostStage* h = c->host_stage;
  if (!h) {
    h = new HostStage();
    c->host_stage = h;
    HIP_TRY(hipStreamCreateWithFlags(&h->up, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&h->down, hipStreamNonBlocking));
    for (int s = 0; s < HostStage::NS; ++s) {
      HIP_TRY(hipEventCreateWithFlags(&h->ev_up[s], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&h->ev_comp[s], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&h->ev_down[s], hipEventDisableTiming));
    }
  }
  {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const int T = c->opt.host_threads > 0 ? c->opt.host_threads : (int)std::min(8u, hw);      // (8 measured best on a 256-thread host: 4 slightly slower, 16 - 64 progressively slower -- the pieces are 8 MiB)
    if (h->pool.threads() != T) { h->pool.stop(); h->pool.start(T - 1); }      // (first use, or the option changed)
  }
  if (h->slot_bytes < slot_bytes) {
    HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipStreamSynchronize(h->up)); HIP_TRY(hipStreamSynchronize(h->down));
    for (int s = 0; s < HostStage::NS; ++s) for (int k = 0; k < 3; ++k) {
      if (h->pin[s][k]) { hipHostFree(h->pin[s][k]); h->pin[s][k] = nullptr; }
      if (h->dev[s][k]) { hipFree(h->dev[s][k]); h->dev[s][k] = nullptr; }
    }
    h->slot_bytes = 0;
    for (int s = 0; s < HostStage::NS; ++s) for (int k = 0; k < 3; ++k) {
      if (hipHostMalloc(&h->pin[s][k], slot_bytes, hipHostMallocDefault) != hipSuccess || hipMalloc(&h->dev[s][k], slot_bytes) != hipSuccess) { (void)hipGetLastError(); FHESI_FAIL("host staging: allocation of %zu bytes (pinned + device) failed", slot_bytes); }
    }
    h->slot_bytes = slot_bytes;
  }
  *out = h;
  return 0;
}
// 0: pageable (or unknown to the runtime), 1: pinned / registered host memory covering [p, p + bytes), 2: device or managed memory (rejected
// by the host-buffer entry: its staging copies are CPU memcpy).  A range only partly registered counts as pageable: the copy threads can
// read and write it, the DMA engines could not.
static int host_ptr_kind(const void* p, size_t bytes) {
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return 0; }
  if (at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeArray) return 2;
  if (at.type != hipMemoryTypeHost) return 0;          // (unregistered and MANAGED memory: the CPU can read it -- the staging path, like pageable memory)
  if (bytes > 1) {
    hipPointerAttribute_t end;
    if (hipPointerGetAttributes(&end, (const char*)p + bytes - 1) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (end.type != hipMemoryTypeHost) return 0;
  }
  return 1;
}
extern "C" int fhesi_host_alloc(fhesi_ctx* c, size_t bytes, void** out) {
  CHECK_CTX(c);
  if (!out) FHESI_FAIL("host_alloc: null output pointer");
  if (hipHostMalloc(out, bytes ? bytes : 8, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); *out = nullptr; FHESI_FAIL("host_alloc: %zu pinned bytes not available", bytes); }
  return 0;
}
// (the context is NOT dereferenced: pinned allocations are not counted in live_handles, so a caller -- or a garbage collector -- may free
// them after fhesi_ctx_destroy; hipHostFree needs no current device)
extern "C" int fhesi_host_free(fhesi_ctx* /*ctx: unused, may be null or already destroyed*/, void* p) {
  if (p) HIP_TRY(hipHostFree(p));
  return 0;
}
// the staging ring of fhesi_ct_mul_relin_batch (two slots of pinned + device memory for three operands, ~1.5 GiB of each at the stress
// ring) is kept between calls; a caller that is done with host-buffer batches hands it back here (the next call allocates it again)
extern "C" int fhesi_host_stage_release(fhesi_ctx* c) {
  CHECK_CTX(c);
  if (c->host_stage) { HIP_TRY(hipStreamSynchronize(c->stream)); host_stage_free(c); }
  return 0;
}

extern "C" int fhesi_ct_mul_relin_batch(fhesi_ctx* c, const fhesi_ksk* k, int32_t logQ, uint64_t p, int32_t decomp_bytes, const uint64_t* a,
                                        const uint64_t* b, uint64_t* out, int32_t nlimbs, int64_t count) {
  CHECK_CTX(c);
  if (!count) return 0;
  if (!a || !b || !out) FHESI_FAIL("ct_mul_relin: null host buffer");
  const size_t ct_bytes = (size_t)2 * c->phim * nlimbs * 8;
  // stage size: the whole batch in at least two stages (so that transfers and compute overlap), at most 32 ciphertexts (64 MiB per operand
  // at the metric ring; larger stages only lengthen the pipeline's fill and drain)
  // (measured at the metric ring, profiles/r04_host_buffers.txt: 1024 per call 7.9 / 9.6 / 9.7 / 10.6 k mults/s with stages of 8 / 16 / 32
  // ciphertexts and 4 - 8 copy threads, 7.7 k with 128; 64 per call 8.5 k with stages of 16, 7.6 k with 32; 8 per call 5.1 k with two stages of 4)
  i64 hc = c->opt.host_chunk > 0 ? c->opt.host_chunk : (count >= 256 ? 32 : (count >= 32 ? 16 : std::max<i64>(1, (count + 1) / 2)));
  if (hc > count) hc = count;
  const size_t total_bytes = (size_t)count * ct_bytes;
  const int ka = host_ptr_kind(a, total_bytes), kb = host_ptr_kind(b, total_bytes), ko = host_ptr_kind(out, total_bytes);
  if (ka == 2 || kb == 2 || ko == 2) FHESI_FAIL("ct_mul_relin_batch takes HOST buffers (operand %s is device memory): use fhesi_ct_mul_relin_batch_dev", ka == 2 ? "a" : kb == 2 ? "b" : "out");
  const bool pa = ka == 1, pb = kb == 1, po = ko == 1;
  HostStage* h;
  FHESI_TRY(host_stage_get(c, (size_t)hc * ct_bytes, &h));
  const i64 nst = (count + hc - 1) / hc;
  // inside a stage the pageable side moves in pieces: the copy threads fill piece j + 1 of the ring while the DMA engine takes piece j up,
  // and on the way back they empty piece j while piece j + 1 comes down (one event per piece)
  const size_t piece = (size_t)8 << 20;
  const size_t npc_max = ((size_t)hc * ct_bytes + piece - 1) / piece;
  for (int s = 0; s < HostStage::NS; ++s)
    while (h->ev_piece[s].size() < npc_max) { hipEvent_t e; HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); h->ev_piece[s].push_back(e); }
  int rc = 0;
  auto upload = [&](int s, int which, const void* src, bool pinned, size_t bytes) -> bool {
    if (pinned) return hipMemcpyAsync(h->dev[s][which], src, bytes, hipMemcpyHostToDevice, h->up) == hipSuccess;
    for (size_t o = 0; o < bytes; o += piece) {
      const size_t len = std::min(piece, bytes - o);
      h->copy((char*)h->pin[s][which] + o, (const char*)src + o, len);
      if (hipMemcpyAsync((char*)h->dev[s][which] + o, (char*)h->pin[s][which] + o, len, hipMemcpyHostToDevice, h->up) != hipSuccess) return false;
    }
    return true;
  };
  auto finish = [&](i64 st) -> int {            // stage st: as its pieces arrive, hand them to the caller's buffer
    const int s = (int)(st % HostStage::NS);
    const size_t bytes = (size_t)std::min(hc, count - st * hc) * ct_bytes, off = (size_t)st * hc * ct_bytes;
    if (po) { if (hipEventSynchronize(h->ev_down[s]) != hipSuccess) { fhesi_set_error("download of ciphertext batch failed"); return 1; } return 0; }
    size_t j = 0;
    for (size_t o = 0; o < bytes; o += piece, ++j) {
      if (hipEventSynchronize(h->ev_piece[s][j]) != hipSuccess) { fhesi_set_error("download of ciphertext batch failed"); return 1; }
      h->copy((char*)out + off + o, (char*)h->pin[s][2] + o, std::min(piece, bytes - o));
    }
    return 0;
  };
  for (i64 st = 0; st < nst && !rc; ++st) {
    const int s = (int)(st % HostStage::NS);
    const i64 cnt = std::min(hc, count - st * hc);
    const size_t bytes = (size_t)cnt * ct_bytes, off = (size_t)st * hc * ct_bytes;
    // the slot's previous user (stage st - 2) was finished before stage st - 1 was issued: its pinned and device buffers are free
    if (!upload(s, 0, (const char*)a + off, pa, bytes) || !upload(s, 1, (const char*)b + off, pb, bytes) ||
        hipEventRecord(h->ev_up[s], h->up) != hipSuccess || hipStreamWaitEvent(c->stream, h->ev_up[s], 0) != hipSuccess) { fhesi_set_error("upload of ciphertext batch failed"); rc = 1; break; }
    rc = fhesi_ct_mul_relin_batch_dev(c, k, logQ, p, decomp_bytes, (const u64*)h->dev[s][0], (const u64*)h->dev[s][1], (u64*)h->dev[s][2], nlimbs, cnt);
    if (rc) break;
    bool ok = hipEventRecord(h->ev_comp[s], c->stream) == hipSuccess && hipStreamWaitEvent(h->down, h->ev_comp[s], 0) == hipSuccess;
    if (ok && po) ok = hipMemcpyAsync((char*)out + off, h->dev[s][2], bytes, hipMemcpyDeviceToHost, h->down) == hipSuccess;
    else if (ok) {
      size_t j = 0;
      for (size_t o = 0; ok && o < bytes; o += piece, ++j)
        ok = hipMemcpyAsync((char*)h->pin[s][2] + o, (char*)h->dev[s][2] + o, std::min(piece, bytes - o), hipMemcpyDeviceToHost, h->down) == hipSuccess &&
             hipEventRecord(h->ev_piece[s][j], h->down) == hipSuccess;
    }
    if (!ok || hipEventRecord(h->ev_down[s], h->down) != hipSuccess) { fhesi_set_error("download of ciphertext batch failed"); rc = 1; break; }
    if (st >= 1) rc = finish(st - 1);              // (overlaps stage st on the device; frees the other slot for stage st + 1)
  }
  if (!rc) rc = finish(nst - 1);
  if (rc) { hipStreamSynchronize(h->up); hipStreamSynchronize(c->stream); hipStreamSynchronize(h->down); }
  return rc;
}
