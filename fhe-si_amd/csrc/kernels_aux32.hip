// kernels_aux32.hip -- the key-switch dot product through FOUR 30-bit auxiliary primes (n = 2^14, 2^15, and the linear-convolution rings).
//
// kernels_ksaux.hip computes the key switch's integer dot product S_l = sum_k digit_k (*) K_{k,l} modulo two 60-bit chain primes.
// The same integers (|S_l| < 2^119) are determined just as well by their residues modulo four primes below 2^30, and there a
// multiply-accumulate is ONE v_mad_u64_u32 (30 x 30 -> 60 bits, sixteen of them fit a 64-bit accumulator) instead of the four that
// a 60 x 60-bit product needs -- half the multiplies for the same information, and half the key bytes per multiply through L2.
// The price is an own set of transforms over those primes: 32-bit negacyclic NTTs (Harvey butterflies on lazy values below 4p),
// written for this one size.  Evaluation order is whatever the forward transform produces (no bit reversal anywhere): the dot
// product is element-wise and the inverse transform is the exact mirror.
//
//   p_a = the four largest primes below 2^30 with p = 1 mod 2n  (aux32_init; roots and tables per context, on first use)
//   ntt32_fwd_kernel3<DIGITS>  (ntt32_core.inc) 32 values per thread, 512 threads per row of 2^14: 5 stages in registers, LDS exchange, 5
//                              stages, exchange, 4 stages; rows of 2^15 = head stage + two sub-transforms
//   ntt32_inv_kernel3          the mirror (Gentleman-Sande), 1/n folded into a final multiplication
//   dot32_kernel2              O[ct][r][l][a] = sum_k D[ct][k][a] * K[a][l][r][k]  mod p_a
// The recombination (Garner over the four residues, then exactly as ks_recombine_kernel) is in kernels_crt.hip.
#include "fhesi_internal.h"

#include "ntt32_core.inc"

// rows of 2^15 elements, plain (non-digit) forward transform: the head stage in place before the sub-transforms (key-table build, self-test)
__global__ void __launch_bounds__(256) ntt32_head_kernel(u32* __restrict__ rows, i64 count, int nslots, int a0, Aux32Primes pr, Aux32Head hd) {
  const i64 row = blockIdx.y;
  const int a = a0 + (int)(row % nslots);
  const u32 p = pr.p[a];
  u32* g = rows + row * 2 * A32_N;
  for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < A32_N; e += (i64)gridDim.x * blockDim.x) {
    u32 x = g[e];
    const u32 t = mul_lazy32(g[e + A32_N], hd.head[a], p);
    x = x >= 2 * p ? x - 2 * p : x;
    g[e] = x + t;
    g[e + A32_N] = x + 2 * p - t;
  }
}
// ... and the tail of the inverse: x[e] = (A + B) / 2, x[e + 2^14] = (A - B) psi^-brv(1) / 2 from the two sub-inverses (each already scaled by 2^-14)
__global__ void __launch_bounds__(256) ntt32_tail_kernel(u32* __restrict__ rows, i64 count, int nslots, int a0, Aux32Primes pr, Aux32Head hd) {
  const i64 row = blockIdx.y;
  const int a = a0 + (int)(row % nslots);
  const u32 p = pr.p[a];
  u32* g = rows + row * 2 * A32_N;
  for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < A32_N; e += (i64)gridDim.x * blockDim.x) {
    const u32 A = g[e], B = g[e + A32_N];          // both below p
    u32 s0 = mul_lazy32(A + B, hd.tail_sum[a], p), s1 = mul_lazy32(A + p - B, hd.tail_dif[a], p);
    g[e] = s0 >= p ? s0 - p : s0;
    g[e + A32_N] = s1 >= p ? s1 - p : s1;
  }
}

// ---------------------------------------------------------------------------------------------- host side
bool aux32_applies(const fhesi_ctx* ctx) { return (ctx->pow2 && (ctx->logn == A32_LOGN || ctx->logn == A32_LOGN + 1)) || ctx->lin_q != 0; }
// 2^14, 2^15 (power-of-two rings and padded rows) or 2^16 (padded rows of the linear-convolution rings with 2^15 < 2 phi(m) - 1 <= 2^16)
i64 aux32_row_len(const fhesi_ctx* ctx) {
  if (ctx->lin_q) return (i64)1 << ctx->lin_lg;
  return (ctx->pow2 && ctx->logn == A32_LOGN + 1) ? 2 * (i64)A32_N : (i64)A32_N;
}
static int aux32_init(fhesi_ctx* ctx) {
  if (ctx->aux32) return 0;
  if (!aux32_applies(ctx)) FHESI_FAIL("aux32: only for n = 2^14, 2^15 and for rings m = prime or 2 * prime with 2 phi(m) - 1 <= 2^20");
  fhesi_aux32* x = new fhesi_aux32();
  const i64 n = aux32_row_len(ctx);                // 2^14, or 2^15 .. 2^20 = 2^S sub-transforms per row
  int S = 0;
  while (((i64)A32_N << S) < n) ++S;
  if (S > 6) { delete x; FHESI_FAIL("aux32: rows of 2^%d", A32_LOGN + S); }
  const int lg = A32_LOGN + S, NS = 1 << S;
  x->S = S;
  // the four largest primes below 2^30 that are 1 mod 2n
  int found = 0;
  for (u64 k = ((u64)1 << (29 - lg)) - 1; k > 0 && found < 4; --k) {
    const u64 cand = (k << (lg + 1)) + 1;
    if (cand < ((u64)1 << 30) && hm::is_prime(cand)) x->pr.p[found++] = (u32)cand;
  }
  if (found < 4) { delete x; FHESI_FAIL("aux32: no primes"); }
  const size_t per_prime = (size_t)A32_N << S;
  std::vector<Tw32> hf(4 * per_prime, Tw32{0, 0}), hi(4 * per_prime, Tw32{0, 0}), ff((size_t)n), fi((size_t)n), ht(4 * A32_HT, Tw32{0, 0});
  const int HSN = A32_HSN(S);
  std::vector<Tw32> hs(S >= 3 ? (size_t)4 * 2 * HSN : 0, Tw32{0, 0});      // S >= 3: head / tail twiddles of the stand-alone passes (ntt32_core.inc)
  for (int a = 0; a < 4; ++a) {
    const u64 p = x->pr.p[a];
    u64 psi = 0;
    for (u64 gq = 2; gq < 1000 && !psi; ++gq) {
      const u64 cand = hm::powmod(gq, (p - 1) / (2 * (u64)n), p);
      if (hm::powmod(cand, (u64)n, p) == p - 1) psi = cand;
    }
    if (!psi) { delete x; FHESI_FAIL("aux32: no 2n-th root"); }
    const u64 ipsi = hm::invmod(psi, p);
    auto tw = [&](u64 w) { return Tw32{(u32)w, (u32)((w << 32) / p)}; };
    a32_bitrev_powers(p, psi, lg, ff); a32_bitrev_powers(p, ipsi, lg, fi);       // the full table of the n-point transform: psi^brv(idx)
    auto fwd_form = [](Tw32 t) { return Tw32{0u - t.w, t.wp}; };      // what a32_ct takes: (-w mod 2^32, floor(w 2^32 / p))
    if (!S) {
      std::transform(ff.begin(), ff.end(), hf.begin() + (size_t)a * per_prime, fwd_form);
      std::copy(fi.begin(), fi.end(), hi.begin() + (size_t)a * per_prime);
    } else {
      // sub-block h runs stage s >= S of the row on its groups i = h 2^(s-S) + i':  own index m' + i' (m' = 2^(s-S))  <->  (m' << S) + h m' + i'
      for (int h = 0; h < NS; ++h)
        for (u64 mp = 1; mp < (u64)A32_N; mp <<= 1)
          for (u64 ip = 0; ip < mp; ++ip) {
            hf[((size_t)a * NS + h) * A32_N + mp + ip] = fwd_form(ff[(mp << S) + h * mp + ip]);
            hi[((size_t)a * NS + h) * A32_N + mp + ip] = fi[(mp << S) + h * mp + ip];
          }
      const u64 inv2 = (p + 1) / 2;
      x->hd.head[a] = ff[1];
      x->hd.tail_sum[a] = tw(inv2);
      x->hd.tail_dif[a] = tw(hm::mulmod(fi[1].w, inv2, p));
      if (S == 2) {
        x->hd.head1[a][0] = ff[2]; x->hd.head1[a][1] = ff[3];
        Tw32* c = &ht[(size_t)a * A32_HT];
        c[0] = ff[1]; c[1] = ff[2]; c[2] = ff[3];
        c[4] = tw(inv2); c[5] = tw(hm::mulmod(fi[2].w, inv2, p)); c[6] = tw(hm::mulmod(fi[3].w, inv2, p)); c[7] = tw(hm::mulmod(fi[1].w, inv2, p));
      }
      if (S >= 3) {
        Tw32* f = &hs[(size_t)a * 2 * HSN];
        Tw32* b = f + HSN;
        for (int idx = 1; idx < NS; ++idx) { f[idx] = ff[idx]; b[idx] = fi[idx]; }
        const u64 c = hm::invmod((u64)NS % p, p), cm = hm::mulmod(c, ((u64)1 << 32) % p, p);      // 2^-S and 2^-S 2^32
        b[NS] = tw(c); b[NS + 1] = tw(hm::mulmod(fi[1].w, c, p)); b[NS + 2] = tw(cm); b[NS + 3] = tw(hm::mulmod(fi[1].w, cm, p));
      }
    }
    const u64 ninv = hm::invmod(A32_N % p, p);     // of the 2^14-point (sub-)transform; the tail of a 2^15-point row carries the other 1/2
    x->pr.ninv[a] = (u32)ninv;
    x->pr.ninv_p[a] = (u32)((ninv << 32) / p);
    x->pr.pinv64[a] = ~(u64)0 / p;
    x->pr.r64[a] = (u64)(((u128)1 << 64) % p);
    x->pr.r48[a] = ((u64)1 << 48) % p;
    { u32 inv = 1; for (int it = 0; it < 5; ++it) inv *= 2u - (u32)p * inv; x->pr.mont[a] = 0u - inv; }      // Newton: p^-1 mod 2^32
    const u64 nm = hm::mulmod(ninv, ((u64)1 << 32) % p, p);
    x->pr.ninv_m[a] = (u32)nm; x->pr.ninv_m_p[a] = (u32)((nm << 32) / p);
    // the inverse transform's last stage carries the final scaling (ntt32_inv_kernel3): table entries 0 and 1 of every (prime, sub-block) become
    // 1/n and w / n, w = that sub-block's distance-16 twiddle; the Montgomery pair (2^32 / n, w 2^32 / n) travels with the primes
    for (int h = 0; h < NS; ++h) {
      Tw32* t0 = &hi[((size_t)a * NS + h) * A32_N];
      const u64 w1 = t0[1].w, wn = hm::mulmod(w1, ninv, p), wm = hm::mulmod(w1, nm, p);
      t0[0] = tw(ninv); t0[1] = tw(wn);
      if (h < 4) { x->pr.ninv_mw[a][h] = (u32)wm; x->pr.ninv_mw_p[a][h] = (u32)((wm << 32) / p); }      // (S >= 3 runs the plain inverse: the 2^32 sits in the tail pass)
    }
    for (int h = NS; h < 4; ++h) { x->pr.ninv_mw[a][h] = x->pr.ninv_mw[a][0]; x->pr.ninv_mw_p[a][h] = x->pr.ninv_mw_p[a][0]; }
    if (p > ((u64)1 << 30) - ((u64)1 << 15) + 1) { delete x; FHESI_FAIL("aux32: prime above 2^30 - 2^15 + 1"); }     // the bound dot32_kernel2's accumulation relies on
  }
  a32_permute_phase_c(hf); a32_permute_phase_c(hi);      // (the last four stages' twiddles in the order the waves load them: A32_TWC)
  if (hipMalloc(&x->d_fwd, hf.size() * sizeof(Tw32)) != hipSuccess || hipMalloc(&x->d_inv, hi.size() * sizeof(Tw32)) != hipSuccess) { delete x; FHESI_FAIL("aux32: hipMalloc failed"); }
  HIP_TRY(hipMemcpy(x->d_fwd, hf.data(), hf.size() * sizeof(Tw32), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(x->d_inv, hi.data(), hi.size() * sizeof(Tw32), hipMemcpyHostToDevice));
  if (S >= 2) {
    if (hipMalloc(&x->d_ht, ht.size() * sizeof(Tw32)) != hipSuccess || hipMalloc(&x->d_p, 4 * sizeof(u32)) != hipSuccess) { delete x; FHESI_FAIL("aux32: hipMalloc failed"); }
    HIP_TRY(hipMemcpy(x->d_ht, ht.data(), ht.size() * sizeof(Tw32), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(x->d_p, x->pr.p, 4 * sizeof(u32), hipMemcpyHostToDevice));
  }
  if (S >= 3) {
    if (hipMalloc(&x->d_hs, hs.size() * sizeof(Tw32)) != hipSuccess) { delete x; FHESI_FAIL("aux32: hipMalloc failed"); }
    HIP_TRY(hipMemcpy(x->d_hs, hs.data(), hs.size() * sizeof(Tw32), hipMemcpyHostToDevice));
  }
  ctx->aux32 = x;
  return 0;
}
void aux32_free(fhesi_ctx* ctx) {
  if (!ctx->aux32) return;
  hipFree(ctx->aux32->d_fwd); hipFree(ctx->aux32->d_inv); hipFree(ctx->aux32->d_ht); hipFree(ctx->aux32->d_p); hipFree(ctx->aux32->d_hs);
  delete ctx->aux32;
  ctx->aux32 = nullptr;
}

// rows: [count][nslots][row length]; row length = 2^14 << S.  The 2^15-point forms exist for the third-form kernels only.
int launch_ntt32_fwd(fhesi_ctx* ctx, u32* d_rows, i64 count, int nslots, int a0) {
  FHESI_TRY(aux32_init(ctx));
  if (!count) return 0;
  const fhesi_aux32* x = ctx->aux32;
  const int S = x->S;
  if (S >= 2) {      // head stages as a pass of their own, at most 65535 rows (a multiple of nslots) per launch
    const i64 nr = count * nslots, step = (65535 / nslots) * (i64)nslots;
    for (i64 r0 = 0; r0 < nr; r0 += step) {
      const unsigned ny = (unsigned)std::min(step, nr - r0);
      u32* rp = d_rows + (r0 << (A32_LOGN + S));
      if (S == 2) ntt32_head2_kernel<<<dim3(16, ny), 256, 0, ctx->stream>>>(rp, nslots, a0, x->d_p, x->d_ht);
      else if (S == 3) ntt32_headS_kernel<3><<<dim3(A32_N / 256, ny), 256, 0, ctx->stream>>>(rp, nslots, a0, x->d_p, x->d_hs);
      else if (S == 4) ntt32_headS_kernel<4><<<dim3(A32_N / 256, ny), 256, 0, ctx->stream>>>(rp, nslots, a0, x->d_p, x->d_hs);
      else if (S == 5) ntt32_headS_kernel<5><<<dim3(A32_N / 256, ny), 256, 0, ctx->stream>>>(rp, nslots, a0, x->d_p, x->d_hs);
      else ntt32_headS_kernel<6><<<dim3(A32_N / 256, ny), 256, 0, ctx->stream>>>(rp, nslots, a0, x->d_p, x->d_hs);
    }
  }
  else if (S) ntt32_head_kernel<<<dim3(16, (unsigned)(count * nslots)), 256, 0, ctx->stream>>>(d_rows, count, nslots, a0, x->pr, x->hd);
  if (S) HIP_TRY(hipGetLastError());
  if (count > 0x7fffffff || (nslots << S) > 65535) FHESI_FAIL("ntt32: too many rows per launch");
  const dim3 grid((unsigned)count, (unsigned)(nslots << S));
#define A32_PLAIN_GO(SS) do { PROF_KERNEL(ctx, PROF_NTT_FWD, ntt32_fwd_kernel3<false, SS>); ntt32_fwd_kernel3<false, SS><<<grid, A32_T, 0, ctx->stream>>>(d_rows, count, nslots, a0, x->pr, x->d_fwd, Dig32Src{}, x->hd); } while (0)
  if (S == 3) A32_PLAIN_GO(3); else if (S == 4) A32_PLAIN_GO(4); else if (S == 5) A32_PLAIN_GO(5); else if (S == 6) A32_PLAIN_GO(6);
#undef A32_PLAIN_GO
  else if (S == 2) { PROF_KERNEL(ctx, PROF_NTT_FWD, ntt32_fwd_kernel3<false, 2>); ntt32_fwd_kernel3<false, 2><<<grid, A32_T, 0, ctx->stream>>>(d_rows, count, nslots, a0, x->pr, x->d_fwd, Dig32Src{}, x->hd); }
  else if (S) { PROF_KERNEL(ctx, PROF_NTT_FWD, ntt32_fwd_kernel3<false, 1>); ntt32_fwd_kernel3<false, 1><<<grid, A32_T, 0, ctx->stream>>>(d_rows, count, nslots, a0, x->pr, x->d_fwd, Dig32Src{}, x->hd); }
  else { PROF_KERNEL(ctx, PROF_NTT_FWD, ntt32_fwd_kernel3<false, 0>); ntt32_fwd_kernel3<false, 0><<<grid, A32_T, 0, ctx->stream>>>(d_rows, count, nslots, a0, x->pr, x->d_fwd, Dig32Src{}, x->hd); }
  HIP_TRY(hipGetLastError());
  return 0;
}
// mont: the rows carry the factor 2^-32 of dot32_kernel2's Montgomery step
int launch_ntt32_inv(fhesi_ctx* ctx, u32* d_rows, i64 count, int nslots, int a0, bool mont, bool tail) {
  FHESI_TRY(aux32_init(ctx));
  if (!count) return 0;
  const fhesi_aux32* x = ctx->aux32;
  const int S = x->S;
  ProfScope prof(ctx, PROF_NTT_INV, (double)(count * nslots));
  if (count > 0x7fffffff || (nslots << S) > 65535) FHESI_FAIL("ntt32: too many rows per launch");
  const dim3 grid((unsigned)count, (unsigned)(nslots << S));
  // (S >= 3: the plain inverse; the 2^32 of a Montgomery-form row is a constant of the tail pass)
  if (mont && S < 3) { PROF_KERNEL(ctx, PROF_NTT_INV, ntt32_inv_kernel3<true>); ntt32_inv_kernel3<true><<<grid, A32_T, 0, ctx->stream>>>(d_rows, count, nslots, a0, x->pr, x->d_inv, S, nullptr); }
  else { PROF_KERNEL(ctx, PROF_NTT_INV, ntt32_inv_kernel3<false>); ntt32_inv_kernel3<false><<<grid, A32_T, 0, ctx->stream>>>(d_rows, count, nslots, a0, x->pr, x->d_inv, S, nullptr); }
  HIP_TRY(hipGetLastError());
  if (S >= 2 && !tail) FHESI_FAIL("ntt32: rows of 2^16 and longer have no consumer that takes their tail stages");
  if (S && tail) {
    if (S >= 2) {
      const i64 nr = count * nslots, step = (65535 / nslots) * (i64)nslots;
      for (i64 r0 = 0; r0 < nr; r0 += step) {
        const unsigned ny = (unsigned)std::min(step, nr - r0);
        u32* rp = d_rows + (r0 << (A32_LOGN + S));
        if (S == 2) ntt32_tail2_kernel<<<dim3(16, ny), 256, 0, ctx->stream>>>(rp, nslots, a0, x->d_p, x->d_ht);
        else if (S == 3) ntt32_tailS_kernel<3><<<dim3(A32_N / 256, ny), 256, 0, ctx->stream>>>(rp, nslots, a0, x->d_p, x->d_hs, mont ? 1 : 0);
        else if (S == 4) ntt32_tailS_kernel<4><<<dim3(A32_N / 256, ny), 256, 0, ctx->stream>>>(rp, nslots, a0, x->d_p, x->d_hs, mont ? 1 : 0);
        else if (S == 5) ntt32_tailS_kernel<5><<<dim3(A32_N / 256, ny), 256, 0, ctx->stream>>>(rp, nslots, a0, x->d_p, x->d_hs, mont ? 1 : 0);
        else ntt32_tailS_kernel<6><<<dim3(A32_N / 256, ny), 256, 0, ctx->stream>>>(rp, nslots, a0, x->d_p, x->d_hs, mont ? 1 : 0);
      }
    }
    else ntt32_tail_kernel<<<dim3(16, (unsigned)(count * nslots)), 256, 0, ctx->stream>>>(d_rows, count, nslots, a0, x->pr, x->hd);
    HIP_TRY(hipGetLastError());
  }
  return 0;
}
int aux32_tail_consts(fhesi_ctx* ctx, uint32_t (*tw)[2], uint32_t (*twp)[2]) {
  FHESI_TRY(aux32_init(ctx));
  const fhesi_aux32* x = ctx->aux32;
  for (int a = 0; a < 4; ++a) { tw[a][0] = x->hd.tail_sum[a].w; twp[a][0] = x->hd.tail_sum[a].wp; tw[a][1] = x->hd.tail_dif[a].w; twp[a][1] = x->hd.tail_dif[a].wp; }
  return 0;
}
// digit rows, tiled [4][row length / 64][npolys * nd][64] u32, straight from the scaled-down parts
int launch_ntt32_fwd_digits(fhesi_ctx* ctx, const u64* d_parts, int nl, int digit_bits, int nd, i64 npolys, u32* d_out, i64 sub_units, bool wm) {
  FHESI_TRY(aux32_init(ctx));
  if (!npolys) return 0;
  const fhesi_aux32* x = ctx->aux32;
  const int S = x->S;
  ProfScope prof(ctx, PROF_NTT_FWD, (double)(npolys * nd * 4));
  ProfScope main_prof(ctx, PROF_NTT_FWD_DIGITS_MAIN, (double)(npolys * nd * 4));
  const i64 units = npolys * nd;
  if (units > 0x7fffffff) FHESI_FAIL("ntt32: too many digit rows per launch");
  if (digit_bits > 30) FHESI_FAIL("ntt32: digits of %d bits (the first stage of a digit row assumes values below 2p)", digit_bits);
  Dig32Src src{d_parts, nl, digit_bits, nd, (u32)ctx->phim, (u32)sub_units, div32_inv((u32)nd), div32_inv((u32)sub_units)};
  if (S >= 3) {
    // rows of 2^17 .. 2^20, the simple path: digits + head stages into plain rows (workspace slot 11)
    if (!ctx->lin_q) FHESI_FAIL("ntt32: digit rows of 2^%d exist for the padded linear-convolution rings only", A32_LOGN + S);
    if (units > 65535) FHESI_FAIL("ntt32: %lld digit rows of 2^%d in one launch (at least 2 MB each: the caller's chunks are smaller)", (long long)units, A32_LOGN + S);
    void* tmp;
    FHESI_TRY(ws_reserve(ctx, 11, (size_t)units * 4 * ((size_t)A32_N << S) * 4, &tmp));
    u32* d_plain = (u32*)tmp;
    const dim3 hg(A32_N / 256, (unsigned)units);
#define A32_DH(SS) do { if (wm) dig32_headS_kernel<SS, true><<<hg, 256, 0, ctx->stream>>>(src, d_plain, x->pr, x->d_hs); else dig32_headS_kernel<SS, false><<<hg, 256, 0, ctx->stream>>>(src, d_plain, x->pr, x->d_hs); } while (0)
    if (S == 3) A32_DH(3); else if (S == 4) A32_DH(4); else if (S == 5) A32_DH(5); else A32_DH(6);
#undef A32_DH
    HIP_TRY(hipGetLastError());
    // the sub-transforms read those rows and store the tiled layout the dot product reads (ntt32_fwd_kernel3<DIGITS, S >= 3>: row source)
    Dig32Src rs = src;
    rs.parts = reinterpret_cast<const u64*>(d_plain);
    const i64 blocks3 = (units + 7) / 8 * 8 * (4 << S);
    if (blocks3 > 0x7fffffff) FHESI_FAIL("ntt32: too many workgroups per launch");
#define A32_ROWS_GO(SS) do { PROF_KERNEL(ctx, PROF_NTT_FWD_DIGITS_MAIN, (ntt32_fwd_kernel3<true, SS, true, Aux32Primes, true, false>)); \
    ntt32_fwd_kernel3<true, SS, true, Aux32Primes, true, false><<<(unsigned)blocks3, A32_T, 0, ctx->stream>>>(d_out, units, 4, 0, x->pr, x->d_fwd, rs, x->hd); } while (0)
    if (S == 3) A32_ROWS_GO(3); else if (S == 4) A32_ROWS_GO(4); else if (S == 5) A32_ROWS_GO(5); else A32_ROWS_GO(6);
#undef A32_ROWS_GO
    HIP_TRY(hipGetLastError());
    return 0;
  }
  const int PS = 4 << S;
  const i64 blocks = (units + 7) / 8 * 8 * PS;      // units dealt round-robin to the 8 XCDs, PS workgroups (primes x sub-blocks) each
  if (blocks > 0x7fffffff) FHESI_FAIL("ntt32: too many workgroups per launch");
#define A32_DIG_GO(SS, PP, WW) do { PROF_KERNEL(ctx, PROF_NTT_FWD_DIGITS_MAIN, (ntt32_fwd_kernel3<true, SS, PP, Aux32Primes, true, WW>)); \
    ntt32_fwd_kernel3<true, SS, PP, Aux32Primes, true, WW><<<(unsigned)blocks, A32_T, 0, ctx->stream>>>(d_out, npolys * nd, 4, 0, x->pr, x->d_fwd, src, x->hd); } while (0)
#define A32_DIG_W(SS, PP) do { if (wm) A32_DIG_GO(SS, PP, true); else A32_DIG_GO(SS, PP, false); } while (0)
  if (S == 2 && !ctx->lin_q) FHESI_FAIL("ntt32: digit rows of 2^16 exist for the padded linear-convolution rings only");
  if (S == 2) A32_DIG_W(2, true);
  else if (S && ctx->lin_q) A32_DIG_W(1, true);
  else if (S) A32_DIG_W(1, false);
  else if (ctx->phim < A32_N) A32_DIG_W(0, true);
  else A32_DIG_W(0, false);
#undef A32_DIG_W
#undef A32_DIG_GO
  HIP_TRY(hipGetLastError());
  return 0;
}
const u32* aux32_primes(fhesi_ctx* ctx) { return aux32_init(ctx) ? nullptr : ctx->aux32->pr.p; }

// ---------------------------------------------------------------------------------------------- key table and dot product
// kint [2*ncol][n][W]: the key's integer coefficients;  rows32[a][(l*2 + r)*ncol + k][n] = (limb l of B bits) mod p_a
// centred: kint holds CENTRED two's complement values in [-2^(B NLB), 2^(B NLB)] (ks32_key_bits): limbs 0 .. NLB - 2 are the unsigned B-bit fields,
// the top limb is floor(x / 2^(B (NLB - 1))) in [-2^B, 2^B], a signed field of B + 2 bits whose residue is taken with its sign
__global__ void __launch_bounds__(256) ks32_limb_scatter_kernel(const u64* __restrict__ kint, u32* __restrict__ rows32, int ncol, int NLB, int B, int W, Aux32Primes pr, i64 n_src, i64 nrow, int centred) {
  const i64 row = blockIdx.y;                 // (r * ncol + k) * NLB + l
  const int l = (int)(row % NLB);
  const i64 rk = row / NLB;
  const int k = (int)(rk % ncol), r = (int)(rk / ncol);
  const int s = B * l, wd = s >> 6, bt = s & 63;
  const i64 rows_per_a = (i64)NLB * 2 * ncol;
  const bool top = centred && l == NLB - 1;
  const int fb = top ? B + 2 : B;             // field width: 66 .. 102 bits
  for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < nrow; j += (i64)gridDim.x * blockDim.x) {
    const u64* x = kint + (rk * n_src + (j < n_src ? j : 0)) * W;          // key polynomials of n_src coefficients, zero above (linear convolution)
    const u64 sfill = (centred && j < n_src && (x[W - 1] >> 63)) ? ~0ull : 0ull;      // words above W: the sign extension of a centred value
    auto word = [&](int i) -> u64 { return j < n_src ? (i < W ? x[i] : sfill) : 0; };
    const u64 w0 = word(wd), w1 = word(wd + 1), w2 = word(wd + 2);
    const u64 lo = bt ? ((w0 >> bt) | (w1 << (64 - bt))) : w0;
    u64 hi = bt ? ((w1 >> bt) | (w2 << (64 - bt))) : w1;
    hi &= ((u64)1 << (fb - 64)) - 1;
    const bool neg = top && ((hi >> (fb - 65)) & 1);                        // sign bit of the top field
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const u64 p = pr.p[a];
      const u64 r64 = (u64)(((u128)1 << 64) % p);
      u64 v = (lo % p + (hi % p) * r64) % p;
      if (neg) { const u64 m = (u64)(((u128)1 << fb) % p); v = (v + p - m) % p; }      // field - 2^fb
      rows32[((i64)a * rows_per_a + ((i64)l * 2 + r) * ncol + k) * nrow + j] = (u32)v;
    }
  }
}
// nbits[0] = max over the values of the smallest nb with -2^nb <= x <= 2^nb, i.e. the bit length of |x| - 1 (x: W-word two's complement)
__global__ void __launch_bounds__(256) ks32_key_bits_kernel(const u64* __restrict__ kint, i64 count, int W, int* __restrict__ nbits) {
  int best = 0;
  for (i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x; g < count; g += (i64)gridDim.x * blockDim.x) {
    const u64* x = kint + g * W;
    const bool neg = x[W - 1] >> 63;
    // m = |x| - 1 as W words: x - 1 for x > 0, ~x = -x - 1 for x < 0; x = 0 counts as 0 bits
    int nb = 0;
    u64 borrow = neg ? 0 : 1;
    bool zero = true;
    for (int i = 0; i < W; ++i) zero = zero && x[i] == 0;
    if (!zero) {
      for (int i = 0; i < W; ++i) {
        u64 w;
        if (neg) w = ~x[i];
        else { w = x[i] - borrow; borrow = (borrow && x[i] == 0) ? 1 : 0; }
        if (w) nb = 64 * i + 64 - __clzll((long long)w);
      }
    }
    best = nb > best ? nb : best;
  }
  for (int o = 32; o; o >>= 1) { const int v = __shfl_xor(best, o); best = v > best ? v : best; }
  if ((threadIdx.x & 63) == 0 && best) atomicMax(nbits, best);
}
int ks32_key_bits(fhesi_ctx* ctx, const u64* d_kint, i64 count, int W, int* nbits) {
  int* d_n;
  HIP_TRY(hipMalloc(&d_n, sizeof(int)));
  HIP_TRY(hipMemsetAsync(d_n, 0, sizeof(int), ctx->stream));
  ks32_key_bits_kernel<<<1024, 256, 0, ctx->stream>>>(d_kint, count, W, d_n);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(nbits, d_n, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  hipFree(d_n);
  if (e != hipSuccess) FHESI_FAIL("key switch: measuring the key coefficients failed: %s", hipGetErrorString(e));
  return 0;
}
// rows of one prime [(l*2 + r)*ncol + k][n]  ->  tiled [l][slice][r][k][64]
__global__ void __launch_bounds__(256) ks32_retile_kernel(const u32* __restrict__ src, u32* __restrict__ dst, int ncol, i64 nrow) {
  const i64 row = blockIdx.y;
  const int k = (int)(row % ncol);
  const i64 lr = row / ncol;
  const int r = (int)(lr & 1);
  const i64 l = lr >> 1;
  const i64 nsl = nrow >> 6;
  for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < nrow; j += (i64)gridDim.x * blockDim.x)
    dst[((((l * nsl + (j >> 6)) * 2 + r) * ncol + k) << 6) + (j & 63)] = src[row * nrow + j];
}

// O[ct][r][l][a][slice] = sum_k D[ct][k][a][slice] * K[a][l][r][k][slice]  mod p_a.   One workgroup = a 64-element slice of CT
// ciphertexts for one prime; the digit slices (reduced below p) sit in LDS, every wave walks its share of the limbs.

// A wave owns one limb l and BOTH key rows r = 0, 1 of it, so every digit value fetched from the LDS tile feeds two multiply-adds
// (round 1's form, one key row per wave, was bound by the LDS pipe: one 16-byte LDS read per 4 multiply-adds saturates it exactly
// when the VALU is saturated).  Accumulation: the v_mad_u64_u32 chain runs straight into a 64-bit total; every 16 columns the bits
// from 48 upwards move into a 32-bit counter.  Both operands are below p <= 2^30 - 2^15 + 1, so 16 products are at most
// 2^64 - 2^50 + 2^34 and a total below 2^48 cannot wrap: no carry detection, 3 registers per output, 2-3 extra instructions per
// output and 16 columns.
//
// The tile is 32 coefficients wide (half a slice) and a wave carries TWO limbs (lanes 0..31 limb 2w, lanes 32..63 limb 2w + 1; both
// halves read the same LDS words, which the LDS broadcasts).  Same multiply-adds per key word and per LDS read, but the tile is half
// as large (66 KB at the metric shape) and the workgroup has half the waves: TWO workgroups share a CU and the tile load, the barrier
// and the ragged end of one overlap the arithmetic of the other (with one 135 KB workgroup per CU the VALU sat idle 37 % of the time).
// (The whole-slice form of round 2 -- one 135 KB workgroup per CU -- and the wave-group split for few limbs left the source in round 6: neither
// is reached since dot32_kernel4 took the matrices with 7 or 8 limbs; profiles/HISTORY.md.)
template <int CT, int NW>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(4, 4)))
dot32_kernel2(const u32* __restrict__ k32, const u32* __restrict__ dig, int ncol, int NLB, i64 count,
                                                         u32* __restrict__ out, Aux32Primes pr, int ntiles, int nsl8, int lognsl /* log2 of the 64-element slices per row */, int sub_lg /* log2 of the ciphertexts per sub-chunk of the tiled digit rows */) {
  extern __shared__ __attribute__((aligned(16))) u32 dl32[];       // [ncol][CT][64 or 32 elements]
  constexpr int LG = 5;
  const u32 lane = threadIdx.x & 63;
  const u32 ln = lane & 31;                        // element within the tile
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // grid: x = s_lo + 8 (tile + ntiles hf), y = s_hi, z = prime -- the same linear order as one flat index (workgroups 8 apart share an
  // XCD), without the three integer divisions a flat index costs every wave (the SALU cannot divide: ~25 VALU instructions each)
  const u32 s_lo = blockIdx.x & 7;
  u32 tile = blockIdx.x >> 3, hf = 0;
  if (tile >= (u32)ntiles) { hf = 1; tile -= (u32)ntiles; }
  const u32 s_hi = blockIdx.y;
  const int a = (int)blockIdx.z;
  const i64 slice = (i64)(s_hi * 8 + s_lo), soff = slice * 64 + hf * 32;
  const int ct0 = (int)tile * CT;
  const u32 p = pr.p[a], twop = 2 * p;
#define DL32(k, c) (((((k) * CT) + (c)) << LG) + ln)
  {
    // One load instruction moves 1 KiB: lane i takes the 16 bytes = elements 4 (i & 15) .. + 3 of column k0 + (i >> 4) of one ciphertext
    // (four consecutive columns of a ciphertext are contiguous in the tiled digit rows), reduces them and writes them with one
    // conflict-free 16-byte LDS write; the arithmetic reads the CT ciphertexts of a column with CT 4-byte reads.
    typedef u32 v4u __attribute__((ext_vector_type(4)));
    const int sub = ct0 >> sub_lg, sub_ct = 1 << sub_lg, ct_in = ct0 & (sub_ct - 1);
    const i64 rest = count - ((i64)sub << sub_lg), cnt_s = rest < sub_ct ? rest : (i64)sub_ct;
    const u32* dbase = dig + ((i64)sub << sub_lg) * ncol * (((i64)4 << lognsl) * 64) + ((((i64)a << lognsl) + slice) * (cnt_s * ncol) + (i64)ct_in * ncol) * 64;
    // (8 lanes per 128-byte half row; 16 consecutive lanes take the same column of TWO ciphertexts, which are adjacent in LDS.)
    // Work split without divisions: NW / NCG waves share a group of ciphertexts and take its column quads round robin.
    constexpr int TB = 6, CG = 2, NCG = CT / CG, WPG = NW / NCG;
    static_assert(NW % NCG == 0, "waves per ciphertext group");
    const int nq = (ncol + 3) >> 2;
    const int cg = w % NCG, q0 = w / NCG;
    const u32 e0 = 4 * (lane & 7), dk = lane >> 4, dc = (lane >> 3) & 1;
    const u32 goff = hf * 32 + e0;
    const int c = cg * CG + (int)dc;
    const bool cok = ct0 + c < count;
    for (int qb = q0; qb < nq; qb += WPG * TB) {
      v4u v[TB];
#pragma unroll
      for (int u = 0; u < TB; ++u) {
        const int q = qb + u * WPG, k = q * 4 + (int)dk;
        v[u] = (q < nq && k < ncol && cok) ? __builtin_nontemporal_load(reinterpret_cast<const v4u*>(dbase + (((i64)c * ncol + k) << 6) + goff)) : v4u{0, 0, 0, 0};
      }
#pragma unroll
      for (int u = 0; u < TB; ++u) {
        const int q = qb + u * WPG, k = q * 4 + (int)dk;
        if (q < nq && k < ncol) {
          v4u y = v[u];
#pragma unroll
          for (int j = 0; j < 4; ++j) { u32 t = y[j]; t = t >= twop ? t - twop : t; y[j] = t >= p ? t - p : t; }
          *reinterpret_cast<v4u*>(&dl32[((k * CT + c) << LG) + e0]) = y;
        }
      }
    }
  }
  __syncthreads();
  const u32 r48 = (u32)pr.r48[a];                 // 2^48 mod p: 30 bits, one multiply-add
  const u32 mont = pr.mont[a];
  constexpr int CW = CT, NWG = NW;      // ciphertexts per wave, waves per group
  const int wl = w % NWG, c0 = (w / NWG) * CW;            // the wave's place in its group; first ciphertext of the group
  for (int lw = wl; lw * 2 < NLB; lw += NWG) {
    const int lraw = 2 * lw + (int)(lane >> 5);          // the upper lanes of the last wave may have no limb
    const bool lok = lraw < NLB;
    // the half wave without a limb (15 limbs on 16 half waves) leaves the loop: its lanes are masked off for the multiply-adds instead of
    // repeating the neighbour's -- same issue slots, but the step runs at the power limit and idle lanes draw less
    if (!lok) continue;
    const int l = lok ? lraw : NLB - 1;
    const u32* kp0 = k32 + (((((((i64)a * NLB + l) << lognsl) + slice) * 2) * ncol) << 6) + (hf * 32 + ln);      // row r = 0; row 1 follows after ncol slices
    const u32* kp1 = kp0 + ((i64)ncol << 6);
    u64 tot[2][CW];
    u32 th[2][CW];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < CW; ++c) { tot[r][c] = 0; th[r][c] = 0; }
    auto fold = [&]() {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < CW; ++c) { th[r][c] += (u32)(tot[r][c] >> 48); tot[r][c] &= 0x0000ffffffffffffull; }
    };
    // Key words are fetched one 4-column chunk ahead into two register sets used alternately (no copies: with a copy at the end of
    // the loop body the compiler waits for the fetch at the START of the body and the L2 latency is exposed once per chunk).
    constexpr int CH = 4;
    const int nfull = ncol & ~(CH - 1), n2 = ncol & ~(2 * CH - 1);
    u32 xa[2][CH], xb[2][CH];
    auto loadc = [&](u32 (&x)[2][CH], int k0) {
#pragma unroll
      for (int u = 0; u < CH; ++u) { x[0][u] = kp0[(k0 + u) << 6]; x[1][u] = kp1[(k0 + u) << 6]; }
    };
    auto macc = [&](const u32 (&x)[2][CH], int k0) {
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        u32 d[CW];
#pragma unroll
        for (int c = 0; c < CW; ++c) d[c] = dl32[DL32(k0 + u, c0 + c)];
#pragma unroll
        for (int c = 0; c < CW; ++c) { tot[0][c] += (u64)x[0][u] * d[c]; tot[1][c] += (u64)x[1][u] * d[c]; }
      }
    };
    // The 8-column pairs are taken in an order ROTATED by the tile number.  The workgroups of the 8 ciphertext tiles of a (slice, prime)
    // run side by side on one XCD and stream the same key block; in the same order they all miss on the same lines at the same moment and
    // every chunk arrives at HBM latency.  Rotated, each of them is the first reader of its own eighth: the whole block is requested in
    // the first microsecond and everything after that is an L2 hit.
    const int npair = n2 / (2 * CH);
    int pst = npair > 1 ? (int)(tile & (0xffffffffu >> __builtin_clz((u32)npair - 1))) : 0;     // (tile mod npair without a division when npair is a power of two; any start is correct)
    if (pst >= npair) pst -= npair;
    auto pk = [&](int i) { int q = i + pst; if (q >= npair) q -= npair; return q * 2 * CH; };
    if (npair) loadc(xa, pk(0)); else if (nfull) loadc(xa, 0);
    for (int i = 0; i < npair; ++i) {
      const int kb = pk(i);
      loadc(xb, kb + CH);
      // the digit words of column u + 1 are read from LDS while column u is multiplied, across the two key chunks (two or three columns
      // ahead measured the same; left to itself the compiler reads a column right before its multiply-adds in the second chunk).  The
      // next chunk of keys is fetched unconditionally -- after the last pair one unused chunk -- because a conditional fetch costs eight
      // register moves per round and 16 registers (-2.5 % together with the prefetch: -4 %).
      u32 d[2][CW];
#pragma unroll
      for (int c = 0; c < CW; ++c) d[0][c] = dl32[DL32(kb, c0 + c)];
#pragma unroll
      for (int u = 0; u < 2 * CH; ++u) {
        if (u == CH) loadc(xa, i + 1 < npair ? pk(i + 1) : (n2 < nfull ? n2 : 0));
        if (u + 1 < 2 * CH) {
#pragma unroll
          for (int c = 0; c < CW; ++c) d[(u + 1) & 1][c] = dl32[DL32(kb + u + 1, c0 + c)];
        }
        const u32 x0 = u < CH ? xa[0][u & (CH - 1)] : xb[0][u & (CH - 1)], x1 = u < CH ? xa[1][u & (CH - 1)] : xb[1][u & (CH - 1)];
#pragma unroll
        for (int c = 0; c < CW; ++c) { tot[0][c] += (u64)x0 * d[u & 1][c]; tot[1][c] += (u64)x1 * d[u & 1][c]; }
      }
      if (i & 1) fold();                             // 16 columns since the last fold
    }
    const int kb = n2;
    if (nfull & CH) { macc(xa, kb); }
    if ((nfull & (3 * CH)) != 0) fold();             // up to 12 columns pending, up to 3 more follow: fold here so that the tail starts below 2^48
    for (int k = nfull; k < ncol; ++k) {             // at most 3 columns
      const u32 x0 = kp0[k << 6], x1 = kp1[k << 6];
#pragma unroll
      for (int c = 0; c < CW; ++c) { const u32 d = dl32[DL32(k, c0 + c)]; tot[0][c] += (u64)x0 * d; tot[1][c] += (u64)x1 * d; }
    }
    fold();                                          // at most 15 columns since the last one; leaves every total below 2^48
    u32* obase = out + ((((i64)l * 4 + a) << (lognsl + 6)) + soff) + ln;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < CW; ++c) {
        if (ct0 + c0 + c < count) {
          // (th 2^48 + tot) 2^-32 mod p by one Montgomery step: v = tot + th (2^48 mod p) < 2^54, m = v (-p^-1) mod 2^32,
          // (v + m p) / 2^32 < p + 2^22.  The factor 2^-32 is undone by the inverse transform's final constant (ntt32_inv_kernel, mont).
          const u64 v = tot[r][c] + (u64)th[r][c] * r48;
          const u32 mq = (u32)v * mont;
          u32 o = (u32)((v + (u64)mq * p) >> 32);
          o = min(o, o - p);                          // o < 2p: o - p wraps to a large value exactly when o < p
          u32* q = obase + (((i64)((ct0 + c0 + c) * 2 + r) * NLB * 4) << (lognsl + 6));
          if (lok) __builtin_nontemporal_store(o, q);
        }
      }
  }
#undef DL32
}

// ---- dot32_kernel2 for MORE COLUMNS THAN ONE LDS TILE HOLDS (the stress ring: 129 columns x 8 ciphertexts x 128 bytes = 132 KB, where two
// workgroups per CU leave each other 80 KB).  Round 4 ran that shape on tiles of 4 ciphertexts -- every key word fetched from L2 feeds 4
// multiply-adds per row instead of 8, and the key stream is what the kernel waits for (0.27 of the roofline against 0.41 at the metric ring).
// Here the columns are taken in NH parts through the same LDS buffer: the accumulators of a wave (its two limbs, both key rows, 8
// ciphertexts) stay in registers across the parts, so the tile keeps its 8 ciphertexts AND the CU its two workgroups.  The form needs
// every wave to carry its limbs for the whole kernel: NLB <= 2 NW (15 limbs at the stress ring).  Half-slice layout.
template <int CT, int NW, int NH>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(4, 4)))
dot32_kernel2p(const u32* __restrict__ k32, const u32* __restrict__ dig, int ncol, int NLB, i64 count, u32* __restrict__ out, Aux32Primes pr, int ntiles, int nsl8,
               int lognsl, int sub_lg, int ncp /* columns per part: a multiple of 8 */) {
  extern __shared__ __attribute__((aligned(16))) u32 dl32[];       // [ncp][CT][32 elements]
  constexpr int LG = 5;
  const u32 lane = threadIdx.x & 63, ln = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32 s_lo = blockIdx.x & 7;
  u32 tile = blockIdx.x >> 3, hf = 0;
  if (tile >= (u32)ntiles) { hf = 1; tile -= (u32)ntiles; }
  const u32 s_hi = blockIdx.y;
  const int a = (int)blockIdx.z;
  const i64 slice = (i64)(s_hi * 8 + s_lo), soff = slice * 64 + hf * 32;
  const int ct0 = (int)tile * CT;
  const u32 p = pr.p[a], twop = 2 * p;
#define DL32(k, c) (((((k) * CT) + (c)) << LG) + ln)
  typedef u32 v4u __attribute__((ext_vector_type(4)));
  const int sub = ct0 >> sub_lg, sub_ct = 1 << sub_lg, ct_in = ct0 & (sub_ct - 1);
  const i64 rest = count - ((i64)sub << sub_lg), cnt_s = rest < sub_ct ? rest : (i64)sub_ct;
  const u32* dbase = dig + ((i64)sub << sub_lg) * ncol * (((i64)4 << lognsl) * 64) + ((((i64)a << lognsl) + slice) * (cnt_s * ncol) + (i64)ct_in * ncol) * 64;
  constexpr int TB = 6, CG = 2, NCG = CT / CG, WPG = NW / NCG;
  static_assert(NW % NCG == 0, "waves per ciphertext group");
  const int cg = w % NCG, q0 = w / NCG;
  const u32 e0 = 4 * (lane & 7), dk = lane >> 4, dc = (lane >> 3) & 1;
  const u32 goff = hf * 32 + e0;
  const int cl = cg * CG + (int)dc;
  const bool cok = ct0 + cl < count;
  // the wave's limbs: lanes 0..31 limb 2w, lanes 32..63 limb 2w + 1 (a half wave without a limb only helps with the tile loads)
  const int lraw = 2 * w + (int)(lane >> 5);
  const bool lok = lraw < NLB;
  const int l = lok ? lraw : NLB - 1;
  const u32* kpa = k32 + (((((((i64)a * NLB + l) << lognsl) + slice) * 2) * ncol) << 6) + hf * 32 + ln;
  const u32 r48 = (u32)pr.r48[a], mont = pr.mont[a];
  constexpr int CW = CT;
  u64 tot[2][CW];
  u32 th[2][CW];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < CW; ++c) { tot[r][c] = 0; th[r][c] = 0; }
  auto fold = [&]() {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < CW; ++c) { th[r][c] += (u32)(tot[r][c] >> 48); tot[r][c] &= 0x0000ffffffffffffull; }
  };
  for (int part = 0; part < NH; ++part) {
    const int kbeg = part * ncp, nc = (ncol - kbeg < ncp ? ncol - kbeg : ncp);
    if (part) __syncthreads();                       // every wave is done with the previous part's tile
    {
      const int nq = (nc + 3) >> 2;
      for (int qb = q0; qb < nq; qb += WPG * TB) {
        v4u v[TB];
#pragma unroll
        for (int u = 0; u < TB; ++u) {
          const int q = qb + u * WPG, k = q * 4 + (int)dk;
          v[u] = (q < nq && k < nc && cok) ? __builtin_nontemporal_load(reinterpret_cast<const v4u*>(dbase + (((i64)cl * ncol + kbeg + k) << 6) + goff)) : v4u{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < TB; ++u) {
          const int q = qb + u * WPG, k = q * 4 + (int)dk;
          if (q < nq && k < nc) {
            v4u y = v[u];
#pragma unroll
            for (int j = 0; j < 4; ++j) { u32 t = y[j]; t = t >= twop ? t - twop : t; y[j] = t >= p ? t - p : t; }
            *reinterpret_cast<v4u*>(&dl32[((k * CT + cl) << LG) + e0]) = y;
          }
        }
      }
    }
    __syncthreads();
    if (lok) {
      const u32* kp0 = kpa + ((i64)kbeg << 6);
      const u32* kp1 = kp0 + ((i64)ncol << 6);
      constexpr int CH = 4;
      const int nfull = nc & ~(CH - 1), n2 = nc & ~(2 * CH - 1);
      u32 xa[2][CH], xb[2][CH];
      auto loadc = [&](u32 (&x)[2][CH], int k0) {
#pragma unroll
        for (int u = 0; u < CH; ++u) { x[0][u] = kp0[(k0 + u) << 6]; x[1][u] = kp1[(k0 + u) << 6]; }
      };
      const int npair = n2 / (2 * CH);
      int pst = npair > 1 ? (int)(tile & (0xffffffffu >> __builtin_clz((u32)npair - 1))) : 0;
      if (pst >= npair) pst -= npair;
      auto pk = [&](int i) { int q = i + pst; if (q >= npair) q -= npair; return q * 2 * CH; };
      if (npair) loadc(xa, pk(0)); else if (nfull) loadc(xa, 0);
      for (int i = 0; i < npair; ++i) {
        const int kb = pk(i);
        loadc(xb, kb + CH);
        u32 d[2][CW];
#pragma unroll
        for (int c = 0; c < CW; ++c) d[0][c] = dl32[DL32(kb, c)];
#pragma unroll
        for (int u = 0; u < 2 * CH; ++u) {
          if (u == CH) loadc(xa, i + 1 < npair ? pk(i + 1) : (n2 < nfull ? n2 : 0));
          if (u + 1 < 2 * CH) {
#pragma unroll
            for (int c = 0; c < CW; ++c) d[(u + 1) & 1][c] = dl32[DL32(kb + u + 1, c)];
          }
          const u32 x0 = u < CH ? xa[0][u & (CH - 1)] : xb[0][u & (CH - 1)], x1 = u < CH ? xa[1][u & (CH - 1)] : xb[1][u & (CH - 1)];
#pragma unroll
          for (int c = 0; c < CW; ++c) { tot[0][c] += (u64)x0 * d[u & 1][c]; tot[1][c] += (u64)x1 * d[u & 1][c]; }
        }
        if (i & 1) fold();                           // 16 columns since the last fold
      }
      if (nfull & CH) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
#pragma unroll
          for (int c = 0; c < CW; ++c) { const u32 d = dl32[DL32(n2 + u, c)]; tot[0][c] += (u64)xa[0][u] * d; tot[1][c] += (u64)xa[1][u] * d; }
        }
      }
      if ((nfull & (3 * CH)) != 0) fold();
      for (int k = nfull; k < nc; ++k) {             // at most 3 columns
        const u32 x0 = kp0[k << 6], x1 = kp1[k << 6];
#pragma unroll
        for (int c = 0; c < CW; ++c) { const u32 d = dl32[DL32(k, c)]; tot[0][c] += (u64)x0 * d; tot[1][c] += (u64)x1 * d; }
      }
      fold();                                        // every total below 2^48 again: the next part (or the epilogue) starts clean
    }
  }
  if (lok) {
    u32* obase = out + ((((i64)l * 4 + a) << (lognsl + 6)) + soff) + ln;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < CW; ++c) {
        if (ct0 + c < count) {
          const u64 v = tot[r][c] + (u64)th[r][c] * r48;          // one Montgomery step, as in dot32_kernel2
          const u32 mq = (u32)v * mont;
          u32 o = (u32)((v + (u64)mq * p) >> 32);
          o = min(o, o - p);
          __builtin_nontemporal_store(o, obase + (((i64)((ct0 + c) * 2 + r) * NLB * 4) << (lognsl + 6)));
        }
      }
  }
#undef DL32
}

// ---- dot32_kernel4: the KEY words in LDS, the DIGIT words straight from memory into registers (round 5).
// dot32_kernel2 keeps a tile of digit words in LDS and streams the key words: every digit word read from LDS feeds two multiply-adds (both key
// rows), so the LDS pipe (128 bytes per clock and CU) saturates exactly when the VALU does, and the kernel runs at 64 % VALU busy with the key
// block (62 GB per launch) streaming L2 -> L1 on top.  Here a lane owns ONE coefficient position and CW ciphertexts: it holds all NOUT = 2 NLB
// accumulators (limb, key row) of its CW ciphertexts in registers (NOUT CW 64-bit totals, two waves per SIMD), so a digit word -- loaded once
// from memory in whole 256-byte lines, never through LDS -- feeds NOUT multiply-adds and a key word read from LDS feeds CW.  The key block of
// the workgroup's (prime, slice) passes through LDS in chunks of KC columns (a double buffer; every wave fetches its rows of the next chunk
// one column per step and writes them two steps later), once per 8 CW ciphertexts instead of once per 8.
// What it takes to make this run at the VALU's pace with TWO waves per SIMD, learned the slow way:
//  * the digit words come from HBM (~1.5 us under load, a step lasts ~0.4 us): they are fetched PD steps ahead into a ring of registers;
//  * the loop body is BRANCH-FREE -- with a branch around a load hipcc drains every outstanding load at the next join (`s_waitcnt vmcnt(0)`
//    at each step: the ring is empty and the kernel runs at the load latency, 6.4 ms per launch like dot32_kernel2) -- so columns past the
//    last are clamped loads multiplied by key rows written as ZEROS, and waves without a second output row re-write a neighbour's row with
//    the same words;
//  * the written order is the schedule (sched_barrier between the phases of a step): left alone hipcc hoists the reads of later steps to the
//    top of the 4000-instruction block and spills the accumulators.
// Accumulation: products of a reduced digit (below p) and a key word (below p) are below 2^60; a total is folded once per chunk as
// (hi 2^32 + lo) -> hi (2^32 mod p) + lo < 2^60 + 2^32 (2^32 mod p = 2^32 - 4p is below 2^28 for every prime launch_dot32_k4 admits -- below 2^24 on
// rows of 2^14 and 2^15, 2^24.4 for the fourth prime on rows of 2^16, 2^27.9 on rows of 2^19; the primes of rows of 2^20 sit above the guard and
// take the LDS-tile kernels), and KC <= 12 more products keep it below 2^60 + 2^32 + 12 2^60 < 2^64
// (tests/test_arith32_models.py follows the chain on worst-case operands).  The epilogue is
// dot32_kernel2's Montgomery step (outputs carry the factor 2^-32, undone by the inverse transform).
// Workgroup = 8 waves = 8 CW ciphertexts of one 64-coefficient slice of one prime; grid as dot32_kernel2 (x = slice low bits + 8 * group,
// y = slice high bits, z = prime): the groups of a (slice, prime) sit on one XCD and share its key block in L2.
// PK (below): steps between the fetch of a key row and its write to LDS.  Loads return IN ORDER (one vmcnt counter): the wait for the key rows
// fetched PK steps ago also waits for every digit word requested before them, so the digit ring is never more than PK steps ahead whatever PD
// says -- which is why ring depths 2 / 3 / 4 measured the same in round 5.  PK = 4 with 4 ciphertexts per lane (the registers for it) measured
// 5.04 ms against 5.14 at PK = 2 and 4.95 for the shipped 6 ciphertexts per lane: profiles/r06_ab_stress_dot.txt.
template <int NLBT, int CW, int KC, int PD, int NW = 8, int NSP = 1, int TAIL = 0>
__global__ void __launch_bounds__(NW * 64, NW / 4)
dot32_kernel4(const u32* __restrict__ k32, const u32* __restrict__ dig, int ncol, i64 count, u32* __restrict__ out, Aux32Primes pr, int ngroups, int lognsl, int sub_lg) {
  // NSP > 1 (many limbs: 15 at the stress ring): the outputs are split over NSP wave groups -- a wave carries NO = NOUT / NSP outputs of its CW
  // ciphertexts, the NSP waves of a ciphertext group load the same digit words (the second load hits in L1 / L2)
  constexpr int NOUT = 2 * NLBT, NO = NOUT / NSP, NHA = (NO + 1) / 2, NHB = NO - NHA, ROWS = KC * NOUT, RPS = (NOUT + NW - 1) / NW, PK = 2;
  static_assert(NOUT % NSP == 0 && NW % NSP == 0, "output split");
  static_assert(TAIL >= 0 && TAIL < KC && (TAIL == 0 || TAIL >= PK), "tail chunk");
  static_assert(KC <= 12 && KC % PD == 0 && KC % PK == 0, "ring slots are compile-time; KC products on top of a folded total stay below 2^64");
  extern __shared__ __attribute__((aligned(16))) u32 kl[];        // [2][KC * NOUT][64]
  const u32 lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32 s_lo = blockIdx.x & 7, g = blockIdx.x >> 3, s_hi = blockIdx.y;
  const int a = (int)blockIdx.z;
  const i64 slice = (i64)(s_hi * 8 + s_lo);
  const u32 p = pr.p[a], twop = 2 * p;
  const u32 r32 = 0u - 4u * p;                                    // 2^32 mod p for p in (2^32 / 5, 2^30): below 2^28 for every prime the launcher admits
  const int osel = w % NSP, o_base = osel * NO;                     // this wave's block of outputs
  const i64 ct0 = (i64)g * (NW / NSP * CW) + (i64)(w / NSP) * CW;
  // the CW digit streams of this wave: scalar bases (the tiled digit rows [sub-chunk][prime][slice][ciphertext * ncol + column][64])
  const u32* dbase[CW];
  bool live[CW];
#pragma unroll
  for (int c = 0; c < CW; ++c) {
    const i64 ct = ct0 + c;
    live[c] = ct < count;
    const i64 cc = live[c] ? ct : count - 1;                      // (a wave past the end repeats the last ciphertext's loads and stores nothing)
    const i64 sub = cc >> sub_lg, ct_in = cc & (((i64)1 << sub_lg) - 1);
    const i64 rest = count - (sub << sub_lg), cnt_s = rest < ((i64)1 << sub_lg) ? rest : ((i64)1 << sub_lg);
    dbase[c] = dig + (sub << sub_lg) * ncol * (((i64)4 << lognsl) * 64) + ((((i64)a << lognsl) + slice) * (cnt_s * ncol) + ct_in * ncol) * 64;
  }
  // key rows k32 [a][l][slice][r][column][64]: this wave fetches the rows of outputs o = w, w + 8, ... (o = 2 l + r); a wave without such an
  // output takes the last one again (same words, same LDS row: a benign double write)
  const u32* kwave[RPS];
  int orow[RPS];
#pragma unroll
  for (int j = 0; j < RPS; ++j) {
    orow[j] = w + NW * j < NOUT ? w + NW * j : NOUT - 1;
    kwave[j] = k32 + (((((i64)a * NLBT + (orow[j] >> 1)) << lognsl) + slice) * 2 + (orow[j] & 1)) * ncol * 64;
  }
  const u32 l4 = lane * 4;
  const u32 koff_max = l4 + (u32)(ncol - 1) * 256;
  u64 acc[NO][CW];
#pragma unroll
  for (int o = 0; o < NO; ++o)
#pragma unroll
    for (int c = 0; c < CW; ++c) acc[o][c] = 0;
  // chunk 0 of the key block (columns that do not exist: zeros)
#pragma unroll
  for (int kk = 0; kk < KC; ++kk) {
    const u32 ko = min(l4 + (u32)kk * 256u, koff_max);
#pragma unroll
    for (int j = 0; j < RPS; ++j) { const u32 v = ld32(kwave[j], ko); kl[(kk * NOUT + orow[j]) * 64 + lane] = kk < ncol ? v : 0u; }
  }
  u32 dn[PD][CW];
  u32 koff = l4;
#pragma unroll
  for (int s_ = 0; s_ < PD; ++s_) {
    const u32 ko = min(koff, koff_max);
#pragma unroll
    for (int c = 0; c < CW; ++c) dn[s_][c] = ld32(dbase[c], ko);
    koff += 256;
  }
  __syncthreads();
  u32 kcol = l4 + (u32)KC * 256u;                                  // byte offset of the column whose key rows are fetched next (the next chunk's)
  int kleft = ncol - KC;                                           // columns that exist from there on
  // one chunk of NS column steps (NS = KC, or TAIL for a shorter last chunk that is compiled on its own instead of padded with zero columns);
  // FETCH: the key rows of a next chunk are fetched.  (A macro, not a lambda: through a generic lambda the same body is allocated into 64 bytes
  // of spills.)  Per step: [write the key rows fetched two steps ago to LDS -- masked HERE, not next to their load, where the mask made every
  // step wait an L2 round trip] [fetch the next chunk's rows of this step's column] [reduce the digit words of this column, fetch column + PD
  // into their ring slot] [read key half B from LDS] | multiply-adds of half A | [read the next step's half A] | multiply-adds of half B;
  // then the rows still on their way, the fold hi (2^32 mod p) + lo (below 2^60 + 2^32) and the barrier.
#define K4_CHUNK(NS, FETCH, ci) do { \
    const u32* __restrict__ cur = kl + (ci & 1) * (ROWS * 64); \
    u32* __restrict__ nxt = kl + ((ci & 1) ^ 1) * (ROWS * 64); \
    u32 kr[PK][RPS]; \
    u32 kqa[NHA], kqb[NHB > 0 ? NHB : 1]; \
_Pragma("unroll") \
    for (int o = 0; o < NHA; ++o) kqa[o] = cur[(o_base + o) * 64 + lane]; \
_Pragma("unroll") \
    for (int kk = 0; kk < NS; ++kk) { \
      __builtin_amdgcn_sched_barrier(0); \
      if (FETCH && kk >= PK) { \
        const u32 keep = kleft + PK > 0 ? 0xffffffffu : 0u; \
_Pragma("unroll") \
        for (int j = 0; j < RPS; ++j) nxt[((kk - PK) * NOUT + orow[j]) * 64 + lane] = kr[kk % PK][j] & keep; \
      } \
      if (FETCH) { \
        const u32 ko = min(kcol, koff_max); \
_Pragma("unroll") \
        for (int j = 0; j < RPS; ++j) kr[kk % PK][j] = ld32(kwave[j], ko); \
        kcol += 256; --kleft; \
      } \
      u32 d[CW]; \
_Pragma("unroll") \
      for (int c = 0; c < CW; ++c) { u32 t = dn[kk % PD][c]; t = min(t, t - twop); d[c] = min(t, t - p); } \
      { \
        const u32 ko = min(koff, koff_max); \
_Pragma("unroll") \
        for (int c = 0; c < CW; ++c) dn[kk % PD][c] = ld32(dbase[c], ko); \
        koff += 256; \
      } \
_Pragma("unroll") \
      for (int o = 0; o < NHB; ++o) kqb[o] = cur[(kk * NOUT + o_base + NHA + o) * 64 + lane]; \
      __builtin_amdgcn_sched_barrier(0); \
_Pragma("unroll") \
      for (int o = 0; o < NHA; ++o) { \
_Pragma("unroll") \
        for (int c = 0; c < CW; ++c) acc[o][c] += (u64)kqa[o] * d[c]; \
      } \
      __builtin_amdgcn_sched_barrier(0); \
      if (kk + 1 < NS) { \
_Pragma("unroll") \
        for (int o = 0; o < NHA; ++o) kqa[o] = cur[((kk + 1) * NOUT + o_base + o) * 64 + lane]; \
      } \
      __builtin_amdgcn_sched_barrier(0); \
_Pragma("unroll") \
      for (int o = 0; o < NHB; ++o) { \
_Pragma("unroll") \
        for (int c = 0; c < CW; ++c) acc[NHA + o][c] += (u64)kqb[o] * d[c]; \
      } \
    } \
    __builtin_amdgcn_sched_barrier(0); \
    if (FETCH) { \
_Pragma("unroll") \
      for (int s_ = NS - PK; s_ < NS; ++s_) { \
        const u32 keep = kleft + (NS - s_) > 0 ? 0xffffffffu : 0u; \
_Pragma("unroll") \
        for (int j = 0; j < RPS; ++j) nxt[(s_ * NOUT + orow[j]) * 64 + lane] = kr[s_ % PK][j] & keep; \
      } \
    } \
_Pragma("unroll") \
    for (int o = 0; o < NO; ++o) \
_Pragma("unroll") \
      for (int c = 0; c < CW; ++c) acc[o][c] = (u64)(u32)(acc[o][c] >> 32) * r32 + (u32)acc[o][c]; \
    __syncthreads(); \
  } while (0)
  if constexpr (TAIL == 0) {                                       // any column count: the last chunk is padded with zero key rows
    const int nch = (ncol + KC - 1) / KC;
    for (int ci = 0; ci < nch; ++ci) K4_CHUNK(KC, true, ci);
  } else {                                                         // ncol = nfull KC + TAIL: the last TAIL columns run a body of their own (66 = 5 x 12 + 6: no padded step)
    const int nfull = ncol / KC;
    for (int ci = 0; ci < nfull; ++ci) K4_CHUNK(KC, true, ci);
    K4_CHUNK((TAIL ? TAIL : KC), false, nfull);
  }
#undef K4_CHUNK
  const u32 mont = pr.mont[a];
  const i64 soff = slice * 64;
#pragma unroll
  for (int oo = 0; oo < NO; ++oo) {
    const int o = o_base + oo;
    u32* obase = out + ((((i64)(o >> 1) * 4 + a) << (lognsl + 6)) + soff) + lane;
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      if (live[c]) {
        const u64 v = acc[oo][c];                         // below 2^60 + 2^32
        const u32 mq = (u32)v * mont;
        u32 ov = (u32)((v + (u64)mq * p) >> 32);          // v 2^-32 mod p, below p + 2^28 + 1 < 2p
        ov = min(ov, ov - p);
        __builtin_nontemporal_store(ov, obase + (((i64)((ct0 + c) * 2 + (o & 1)) * NLBT * 4) << (lognsl + 6)));
      }
    }
  }
}

int ks32_build(fhesi_ctx* ctx, fhesi_ksk* k, const u64* d_kint, int W, int B, int NLB, void* d_tmp, bool centred) {
  FHESI_TRY(aux32_init(ctx));
  const int ncol = k->ncomp * k->ndigits;
  const i64 rows_per_a = (i64)NLB * 2 * ncol, nrow = aux32_row_len(ctx);
  u32* rows32 = (u32*)k->d_aux;
  if ((size_t)4 * rows_per_a * nrow * 4 > k->aux_bytes) FHESI_FAIL("aux32: key table does not fit");
  dim3 grid(64, (unsigned)(2 * ncol * NLB));
  ks32_limb_scatter_kernel<<<grid, 256, 0, ctx->stream>>>(d_kint, rows32, ncol, NLB, B, W, ctx->aux32->pr, ctx->phim, nrow, centred ? 1 : 0);
  HIP_TRY(hipGetLastError());
  for (int a = 0; a < 4; ++a) FHESI_TRY(launch_ntt32_fwd(ctx, rows32 + (i64)a * rows_per_a * nrow, rows_per_a, 1, a));
  for (int a = 0; a < 4; ++a) {
    u32* q = rows32 + (i64)a * rows_per_a * nrow;
    HIP_TRY(hipMemcpyAsync(d_tmp, q, (size_t)rows_per_a * nrow * 4, hipMemcpyDeviceToDevice, ctx->stream));
    dim3 g2(64, (unsigned)rows_per_a);
    ks32_retile_kernel<<<g2, 256, 0, ctx->stream>>>((const u32*)d_tmp, q, ncol, nrow);
    HIP_TRY(hipGetLastError());
  }
  return 0;
}

// d_dig: tiled [4][n/64][count*ncol][64] u32; d_out: [count*2*NLB][4][n] u32
template <int CT, int NW>
static int launch_dot32_t(fhesi_ctx* ctx, const fhesi_ksk* k, const u32* d_dig, int ncol, i64 count, u32* d_out) {
  const size_t shmem = (size_t)ncol * CT * 32 * 4;
  static std::atomic<unsigned long long> attr_done{0};
  if (!(attr_done.load() >> ctx->device & 1)) {
    HIP_TRY(hipFuncSetAttribute((const void*)dot32_kernel2<CT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done.fetch_or(1ull << ctx->device);
  }
  const i64 nrow = aux32_row_len(ctx);
  const int lognsl = hm::ilog2_ceil((u64)nrow) - 6;              // log2 of the 64-element slices per row
  const int ntiles = (int)((count + CT - 1) / CT), nsl8 = (int)(nrow / 64 / 8);
  const i64 blocks = (i64)8 * ntiles * 2;
  if (blocks > 0x7fffffff || nsl8 > 65535) FHESI_FAIL("dot32: too many ciphertexts per call");
  static_assert((kDigitSubCt & (kDigitSubCt - 1)) == 0, "sub-chunks of a power of two");
  int sub_lg = 0;
  while (((i64)1 << sub_lg) < kDigitSubCt) ++sub_lg;
  PROF_KERNEL(ctx, PROF_DOT, (dot32_kernel2<CT, NW>));
  dot32_kernel2<CT, NW><<<dim3((unsigned)blocks, (unsigned)nsl8, 4), NW * 64, shmem, ctx->stream>>>((const u32*)k->d_aux, d_dig, ncol, k->aux_rows, count, d_out, ctx->aux32->pr, ntiles, nsl8, lognsl, sub_lg);
  HIP_TRY(hipGetLastError());
  return 0;
}
template <int CT, int NW, int NH>
static int launch_dot32_p(fhesi_ctx* ctx, const fhesi_ksk* k, const u32* d_dig, int ncol, i64 count, u32* d_out) {
  const int ncp = ((ncol + NH - 1) / NH + 7) & ~7;                 // columns per part: whole 8-column pairs in every part but the last
  const size_t shmem = (size_t)ncp * CT * 32 * 4;
  static std::atomic<unsigned long long> attr_done{0};
  if (!(attr_done.load() >> ctx->device & 1)) {
    HIP_TRY(hipFuncSetAttribute((const void*)dot32_kernel2p<CT, NW, NH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done.fetch_or(1ull << ctx->device);
  }
  const i64 nrow = aux32_row_len(ctx);
  const int lognsl = hm::ilog2_ceil((u64)nrow) - 6;              // log2 of the 64-element slices per row
  const int ntiles = (int)((count + CT - 1) / CT), nsl8 = (int)(nrow / 64 / 8);
  const i64 blocks = (i64)8 * ntiles * 2;
  if (blocks > 0x7fffffff || nsl8 > 65535) FHESI_FAIL("dot32: too many ciphertexts per call");
  int sub_lg = 0;
  while (((i64)1 << sub_lg) < kDigitSubCt) ++sub_lg;
  PROF_KERNEL(ctx, PROF_DOT, (dot32_kernel2p<CT, NW, NH>));
  dot32_kernel2p<CT, NW, NH><<<dim3((unsigned)blocks, (unsigned)nsl8, 4), NW * 64, shmem, ctx->stream>>>((const u32*)k->d_aux, d_dig, ncol, k->aux_rows, count, d_out, ctx->aux32->pr, ntiles, nsl8, lognsl, sub_lg, ncp);
  HIP_TRY(hipGetLastError());
  return 0;
}
template <int NLBT, int CW, int KC, int PD, int NW = 8, int NSP = 1, int TAIL = 0>
static int launch_dot32_k4(fhesi_ctx* ctx, const fhesi_ksk* k, const u32* d_dig, int ncol, i64 count, u32* d_out) {
  const size_t shmem = (size_t)2 * KC * 2 * NLBT * 64 * 4;
  static std::atomic<unsigned long long> attr_done{0};
  if (!(attr_done.load() >> ctx->device & 1)) {
    HIP_TRY(hipFuncSetAttribute((const void*)dot32_kernel4<NLBT, CW, KC, PD, NW, NSP, TAIL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done.fetch_or(1ull << ctx->device);
  }
  for (int a = 0; a < 4; ++a)
    if ((u32)(0u - 4u * ctx->aux32->pr.p[a]) >= (1u << 28) || ctx->aux32->pr.p[a] >= (1u << 30)) FHESI_FAIL("dot32: prime %u outside the range of dot32_kernel4's fold", ctx->aux32->pr.p[a]);
  const i64 nrow = aux32_row_len(ctx);
  const int lognsl = hm::ilog2_ceil((u64)nrow) - 6;
  const int ngroups = (int)((count + NW / NSP * CW - 1) / (NW / NSP * CW)), nsl8 = (int)(nrow / 64 / 8);
  const i64 blocks = (i64)8 * ngroups;
  if (blocks > 0x7fffffff || nsl8 > 65535) FHESI_FAIL("dot32: too many ciphertexts per call");
  int sub_lg = 0;
  while (((i64)1 << sub_lg) < kDigitSubCt) ++sub_lg;
  PROF_KERNEL(ctx, PROF_DOT, (dot32_kernel4<NLBT, CW, KC, PD, NW, NSP, TAIL>));
  dot32_kernel4<NLBT, CW, KC, PD, NW, NSP, TAIL><<<dim3((unsigned)blocks, (unsigned)nsl8, 4), NW * 64, shmem, ctx->stream>>>((const u32*)k->d_aux, d_dig, ncol, count, d_out, ctx->aux32->pr, ngroups, lognsl, sub_lg);
  HIP_TRY(hipGetLastError());
  return 0;
}
int launch_dot32(fhesi_ctx* ctx, fhesi_ksk* k, const u32* d_dig, int ncol, i64 count, u32* d_out, bool* mont) {
  *mont = true;
  if (!count) return 0;
  ProfScope prof(ctx, PROF_DOT, (double)count);
  // keys in LDS, digits in registers (dot32_kernel4): the limb counts of generated matrices at the benchmark rings
  bool k4_primes = true;                    // dot32_kernel4 folds through 2^32 mod p = 2^32 - 4p and needs it below 2^28 (true of every prime aux32_init picks for rows up to 2^19; any other ring takes the LDS-tile kernels)
  for (int a = 0; a < 4; ++a) k4_primes = k4_primes && ctx->aux32->pr.p[a] < (1u << 30) && (u32)(0u - 4u * ctx->aux32->pr.p[a]) < (1u << 28);
  if (ctx->opt.dot32_k4 && count >= 24 && k4_primes) {
    if (k->aux_rows == 7 && ncol % 12 == 6) return launch_dot32_k4<7, 6, 12, 3, 8, 1, 6>(ctx, k, d_dig, ncol, count, d_out);      // (66 columns = 5 x 12 + 6: the tail compiled on its own)
    if (k->aux_rows == 7) return launch_dot32_k4<7, 6, 12, 3>(ctx, k, d_dig, ncol, count, d_out);      // (measured: digit ring 2 / 3 / 4 steps ahead the same; 12 waves x 4 ciphertexts slower, profiles/r05_ab_dot_k4.txt)
    if (k->aux_rows == 8 && ncol % 12 == 6) return launch_dot32_k4<8, 4, 12, 3, 8, 1, 6>(ctx, k, d_dig, ncol, count, d_out);
    if (k->aux_rows == 8) return launch_dot32_k4<8, 4, 12, 3>(ctx, k, d_dig, ncol, count, d_out);      // (5 ciphertexts per lane: 160 bytes of spills -- odd counts leave the 64-bit pairs badly placed)
    // 15 limbs (the stress ring): the 30 outputs split over two wave groups (with all 30 in one lane only 3 ciphertexts fit: 81 ms per 1024 against 37.6 for dot32_kernel2p)
    // (15 limbs -- the stress ring -- keep dot32_kernel2p: with all 30 outputs in one lane only 3 ciphertexts fit (81 ms per 1024 against 37.6); the
    // outputs split over two wave groups, NSP = 2, measured 37.8-42 ms in six configurations of ciphertexts per lane, ring depth, key lead and chunk
    // size against 37.5-38.2 -- profiles/r06_ab_stress_dot.txt; that instantiation left the library in round 6)
  }
  // the digit-tile forms (dot32_kernel2): tiles of 32 coefficients x 8 ciphertexts x all columns in 80 KB, two workgroups per CU (ncol <= 80);
  // more columns (the stress ring's 129) in two parts through one such tile (dot32_kernel2p: the 8-fold reuse of a key word AND two workgroups
  // per CU); matrices with more than 16 limbs (general limbs at the stress ring) in tiles of 4 ciphertexts (ncol <= 160).  The forms measured
  // slower and removed in round 6 -- whole-slice tiles in one 135 KB workgroup, groups of four waves for few limbs, two limbs per wave on tiles
  // of 4 (dot32_kernel3), the int8 matrix-core product (dot_mfma_kernel) -- are in the history of this file and in profiles/HISTORY.md.
  if ((size_t)ncol * 8 * 128 <= 80 * 1024) return launch_dot32_t<8, 8>(ctx, k, d_dig, ncol, count, d_out);
  if (k->aux_rows <= 16 && (size_t)((((ncol + 1) / 2) + 7) & ~7) * 8 * 128 <= 80 * 1024) return launch_dot32_p<8, 8, 2>(ctx, k, d_dig, ncol, count, d_out);
  if ((size_t)ncol * 4 * 128 <= 80 * 1024) return launch_dot32_t<4, 8>(ctx, k, d_dig, ncol, count, d_out);
  FHESI_FAIL("dot32: %d columns do not fit the LDS tile", ncol);
}

// Self-test (tests/test_gpu_ntt.py): the transform pair is a ring isomorphism of Z_p[X]/(X^n + 1) -- it is linear by construction, so
// checking that monomials multiply like monomials (X^i X^j = +-X^(i+j mod n)) and that inverse(forward(x)) = x pins it.
__global__ void aux32_pointwise_kernel(u32* __restrict__ a, const u32* __restrict__ b, u32 p, i64 n) {
  const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) a[j] = (u32)(((u64)a[j] * b[j]) % p);
}
extern "C" int fhesi_selftest_aux32(fhesi_ctx* c) {
  if (!c) FHESI_FAIL("null context");
  HIP_TRY(hipSetDevice(c->device));
  FHESI_TRY(aux32_init(c));
  const i64 N = aux32_row_len(c);
  u32 *da, *db;
  HIP_TRY(hipMalloc(&da, N * 4)); HIP_TRY(hipMalloc(&db, N * 4));
  std::vector<u32> ha(N), hb(N), hr(N);
  int rc = 0;
  const i64 pairs[6][2] = {{0, 0}, {1, 2}, {5, N - 1}, {N - 1, N - 1}, {N / 2, N / 2}, {N / 4 + 1, 3 * (N / 4) + 212}};
  for (int a = 0; a < 4 && !rc; ++a) {
    const u32 p = c->aux32->pr.p[a];
    for (int t = 0; t < 6 && !rc; ++t) {
      std::fill(ha.begin(), ha.end(), 0); std::fill(hb.begin(), hb.end(), 0);
      ha[pairs[t][0]] = 3; hb[pairs[t][1]] = 5;
      HIP_TRY(hipMemcpy(da, ha.data(), N * 4, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
      if (launch_ntt32_fwd(c, da, 1, 1, a) || launch_ntt32_fwd(c, db, 1, 1, a)) { rc = 1; break; }
      aux32_pointwise_kernel<<<(unsigned)(N / 256), 256, 0, c->stream>>>(da, db, p, N);
      if (launch_ntt32_inv(c, da, 1, 1, a, false)) { rc = 1; break; }
      HIP_TRY(hipStreamSynchronize(c->stream));
      HIP_TRY(hipMemcpy(hr.data(), da, N * 4, hipMemcpyDeviceToHost));
      const i64 e = pairs[t][0] + pairs[t][1];
      const i64 pos = e % N;
      const u32 want = e >= N ? p - 15 : 15;
      for (i64 j = 0; j < N; ++j)
        if (hr[j] != (j == pos ? want : 0u)) { fhesi_set_error("aux32 self-test: prime %d, X^%lld * X^%lld: coefficient %lld is %u", a, (long long)pairs[t][0], (long long)pairs[t][1], (long long)j, hr[j]); rc = 1; break; }
    }
    // round trip of a dense vector
    if (!rc) {
      for (i64 j = 0; j < N; ++j) ha[j] = (u32)((1234567u * (u32)j + 89u) % p);
      HIP_TRY(hipMemcpy(da, ha.data(), N * 4, hipMemcpyHostToDevice));
      if (launch_ntt32_fwd(c, da, 1, 1, a) || launch_ntt32_inv(c, da, 1, 1, a, false)) rc = 1;
      HIP_TRY(hipStreamSynchronize(c->stream));
      HIP_TRY(hipMemcpy(hr.data(), da, N * 4, hipMemcpyDeviceToHost));
      for (i64 j = 0; j < N && !rc; ++j) if (hr[j] != ha[j]) { fhesi_set_error("aux32 self-test: prime %d, round trip differs at %lld", a, (long long)j); rc = 1; }
    }
  }
  hipFree(da); hipFree(db);
  return rc;
}
