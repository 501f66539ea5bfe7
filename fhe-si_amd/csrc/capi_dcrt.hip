// capi_dcrt.hip -- Cmodulus::FFT / iFFT, DoubleCRT and SingleCRT objects, batched row kernels (include/fhesi_hip.h)
#include "capi_common.h"

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

