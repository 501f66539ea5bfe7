// kernels_tensor32.hip -- the tensor half of the fused multiplication (Ciphertext::operator*=, Ciphertext.cpp:167-218) over primes below 2^30.
//
// tProd = (p a) (x) b is an INTEGER polynomial triple: its coefficients are bounded by 2 n (p 2^(64 nl - 1)) 2^(64 nl - 1), far below half the
// chain product, and ScaleDown only keeps round(x / 2^logQ) mod 2^logQ of every coefficient x.  Inside fhesi_ct_mul_relin_batch_dev the
// DoubleCRT of tProd is never visible, so the integers may be computed modulo ANY product of primes that exceeds the bound: here the NP
// largest primes below 2^30 that are 1 mod 2^15 (35 of them at the metric shape, 1050 bits, where the chain has 18 primes of 60 bits).
// Same bytes per row set, a quarter of the multiplier work per butterfly, and the residues are 30-bit numbers, which lets the two
// conversions run on single v_mad_u64_u32 chains:
//   rns32_reduce_kernel   coefficient (two's complement, 2 nl words of 32 bits) -> residues below 4p: 4 + 3 + 3 + ... products of
//                         x_k (2^(32k) s mod p) per 64-bit accumulator, folded through 2^32 mod p, one Barrett step at the end
//   ntt32_fwd_kernel3     the rows' forward transforms (ntt32_core.inc)
//   ntt32_inv_kernel3<.., TENSOR>   element-wise products formed in the loader of the inverse transform (no tensor kernel, no rows
//                         written in between)
//   crt32_scale_kernel    x = sum_i y_i M_i - kappa M with M_i = M/p_i cut into words of 28 bits: y_i M_i[l] < 2^58, so the NP + 1 terms of
//                         a word accumulate in 64 bits without carries; kappa from a 26-bit fixed-point sum; only the words from
//                         28 * 14 = 392 bits upwards are formed -- the dropped part is below 2^429 and can change round(x / 2^512) only
//                         when bits 448..511 of x + 2^511 are all ones: those workgroups are flagged and redone with every word
//                         (EXACT), the pattern of crt_sum_kernel.
// Rows of 2^15 (n = 2^15, the stress shape: 70 primes for logQ = 1024): the head stage of the forward transform is taken inside
// rns32_reduce_kernel<NL, true> (a thread converts coefficients j and j + 2^14 and stores the two butterfly outputs into the two sub-rows),
// the tail of the inverse inside crt32_scale_kernel (its first multiplication takes (A + B) or (A - B) with the tail constant folded into
// the CRT constant), so a row still passes through exactly one forward and one inverse kernel.  With more than 62 primes the CRT words
// are 26 bits wide (71 terms below 2^56 per 64-bit accumulator).
// The reference's safe-prime rings (m = 2q', Bluestein rows in the reference and in the chain path) take the same route as LINEAR
// convolutions: phi(m) coefficients zero-padded into rows of 2^14, products of degree < 2 phi(m) - 1, and the reduction modulo
// X^q' + 1 and Phi_m -- out_j = S_j - S_(j+q') - (-1)^j S_(q'-1), linear, so it is applied to the residues in the loader of the CRT
// kernel (crt32_scale_generic_kernel: any logQ, words at run-time positions through LDS).  Sums of products per group
// (fhesi_ct_mul_sum_relin_dev: Matrix products inside Regression) are formed in evaluation form by tensor_sum32_kernel.
// Padded rows of 2^16 take their second head stage in rns32_reduce_kernel<NL, 2> and their tails as a pass of their own; rows of 2^17 .. 2^20
// (round 6: every m = p - 1 below 2^20) are converted as plain zero-padded rows and take head AND tail stages as passes of their own
// (ntt32_headS_kernel / ntt32_tailS_kernel, ntt32_core.inc); the CRT kernel then folds from whole rows.
// fhesi_ct_mul_dev keeps the reference chain (its rows ARE visible).  Option tensor32 = 0 keeps the chain in the fused pipeline too; option
// tensor_bits = 29 takes the primes below 2^29 (fewer range steps in the row transforms, one or two primes more: measured a tie).
#include "fhesi_internal.h"
#include "ntt32_core.inc"
#include <cmath>

// one (lift, limb count, logQ, prime count) configuration: the conversion tables
struct T32Config {
  u64 lift = 0;
  int nl = 0, logQ = 0, NP = 0, R = 28, WT = 0;
  bool generic = false;              // crt32_scale_generic_kernel (any logQ, fold) instead of the compiled shapes
  T32Primes pr;
  u32* d_rns = nullptr;              // [2][NP][2 nl + 6]: (s 2^(32k) mod p) k < 2 nl, -(s 2^(64 nl)) mod p, 2^32 mod p, floor(2^61 / p), p, head twiddle (w, w');  s = lift (class 0) or 1
  Tw32* d_cinv = nullptr;            // [NP][2] (M / p_i)^-1 mod p_i  [times 1/2 | times psi^-brv(1) / 2: the tail of a 2^15-point inverse]
  u32* d_inv57 = nullptr;            // [NP] floor(2^57 / p_i)
  u32* d_Mw = nullptr;               // [NP + 1][WT]: M_i in words of R bits; row NP = 2^(R WT) - M
  ~T32Config() { hipFree(d_rns); hipFree(d_cinv); hipFree(d_inv57); hipFree(d_Mw); }
};
struct fhesi_tensor32 {
  int S = 0;                         // rows of 2^14 << S
  int bits = 30;                     // 30: the largest primes below 2^30; 29: below 2^29 (option tensor_bits: one or two primes more, 6 of 14 / 7 of 13 range steps per row transform)
  std::vector<u32> primes;           // the primes with transform tables, largest first
  std::vector<Tw32> head, tail;      // per prime: psi^brv(1), psi^-brv(1)   (S = 1)
  std::vector<Tw32> head1;           // per prime: psi^brv(2), psi^brv(3)    (S = 2: the second head stage)
  Tw32* d_fwd = nullptr;             // [primes][2^S][2^14]
  Tw32* d_inv = nullptr;
  Tw32* d_ht = nullptr;              // S = 2: [primes][A32_HT] constants of the stand-alone tail pass (ntt32_tail2_kernel)
  u32* d_p = nullptr;                // S >= 2: the primes
  Tw32* d_hs = nullptr;              // S >= 3: [primes][2][A32_HSN(S)] head / tail twiddles of the stand-alone passes (ntt32_headS_kernel / ntt32_tailS_kernel)
  std::vector<T32Config*> cfgs;
  T32Config* cur = nullptr;          // the configuration of the running sum (tensor32_sum_begin)
};

void tensor32_free(fhesi_ctx* ctx) {
  fhesi_tensor32* x = ctx->tensor32;
  if (!x) return;
  for (T32Config* c : x->cfgs) delete c;
  hipFree(x->d_fwd); hipFree(x->d_inv); hipFree(x->d_ht); hipFree(x->d_p); hipFree(x->d_hs);
  delete x;
  ctx->tensor32 = nullptr;
}

// ---------------------------------------------------------------------------------------------- host big integers (little-endian u64 limbs)
typedef std::vector<u64> Big;
static Big big_mul_small(const Big& a, u64 b) { return hm::bn_mul_small(a, b); }
static u64 big_mod_small(const Big& a, u64 q) { u128 r = 0; for (size_t i = a.size(); i-- > 0;) r = ((r << 64) | a[i]) % q; return (u64)r; }
static u32 big_word(const Big& a, int l, int R) {          // word l of the radix-2^R form
  const int bit = R * l, w = bit >> 6, o = bit & 63;
  u64 v = (size_t)w < a.size() ? a[w] >> o : 0;
  if (o + R > 64 && (size_t)w + 1 < a.size()) v |= a[w + 1] << (64 - o);
  return (u32)(v & (((u64)1 << R) - 1));
}

// the two compiled shapes of crt32_scale_kernel, and the run-time form
static constexpr int T32_WT_512 = 38, T32_R_512 = 28;        // logQ = 512: words of 28 bits (up to 62 primes), 1064 bits per table row
static constexpr int T32_WT_1024 = 82, T32_R_1024 = 26;      // logQ = 1024: words of 26 bits (up to 72 primes), 2132 bits per table row
static constexpr int T32_GEN_NW = 24, T32_GEN_NWX = 40;      // generic kernel: words formed in the first pass / in the exact pass (logQ <= 512)

struct T32Plan { int NP = 0; bool generic = false; };
// The number of primes the tensor half needs (0: this shape does not run through it).
//   |x| < 2^TB with TB = 2 (logQ - 1) + bits(p) + log2(coefficients) + 1 [+ log2(terms per sum)] [+ 2: the three-term fold of the
//   safe-prime rings];  M > 2^(TB + 3) keeps x/M below 1/8 (kappa is then decided by a coarse fixed-point sum), and the chain product must
//   exceed 2^(TB + 1) so that the reference's own centred integers are these same integers.
static T32Plan t32_plan_search(const fhesi_ctx* ctx, u64 p, int nlimbs, int logQ, i64 gmax, std::vector<u32>* primes);
// (the search tests a few hundred candidates for primality: a tenth of a millisecond, which a caller of single multiplications would pay
// twice per call -- so the answer is kept per (lift, limbs, logQ, bits of the group size, option tensor32))
static T32Plan t32_plan(const fhesi_ctx* ctx, u64 p, int nlimbs, int logQ, i64 gmax, std::vector<u32>* primes) {
  int gbits = 0;
  while (((i64)1 << gbits) < gmax) ++gbits;
  auto& memo = const_cast<fhesi_ctx*>(ctx)->t32_memo;          // (a context serves one host thread at a time)
  const std::vector<long long> key{(long long)p, nlimbs, logQ, gbits, ctx->opt.tensor32, ctx->opt.tensor_bits};
  auto it = memo.find(key);
  if (it == memo.end()) {
    std::vector<u32> pr;
    const T32Plan pl = t32_plan_search(ctx, p, nlimbs, logQ, gmax, &pr);
    it = memo.emplace(key, std::make_pair(pl.NP * 2 + (pl.generic ? 1 : 0), std::move(pr))).first;
  }
  if (primes) *primes = it->second.second;
  T32Plan pl;
  pl.NP = it->second.first >> 1; pl.generic = it->second.first & 1;
  return pl;
}
static T32Plan t32_plan_search(const fhesi_ctx* ctx, u64 p, int nlimbs, int logQ, i64 gmax, std::vector<u32>* primes) {
  T32Plan pl;
  if (!ctx->opt.tensor32 || p < 2 || nlimbs < 1 || nlimbs > 16 || 64 * nlimbs < logQ || gmax < 1) return pl;
  const bool lin = ctx->lin_q != 0;
  if (!lin && !(ctx->pow2 && (ctx->logn == A32_LOGN || ctx->logn == A32_LOGN + 1))) return pl;
  const int lg = lin ? ctx->lin_lg : ctx->logn;              // rows of 2^lg
  const bool compiled = !lin && (logQ == 512 || logQ == 1024);
  if (!compiled && (logQ < 64 || logQ > 512)) return pl;     // the generic CRT kernel: logQ <= 512 (rows of 2^14, or two sub-rows of a 2^15-point row)
  int pbits = 0, gbits = 0, cbits = 0;
  while (pbits < 64 && (p >> pbits)) ++pbits;
  while (((i64)1 << gbits) < gmax) ++gbits;
  while (((i64)1 << cbits) < ctx->phim) ++cbits;
  // (operands are centred residues modulo 2^logQ, as every reference Ciphertext holds them: Ciphertext.cpp reduces after each operation)
  const double TB = 2.0 * (logQ - 1) + pbits + cbits + 1 + gbits + (lin ? 2 : 0);
  double chain = 0;
  for (int i = 0; i < ctx->L; ++i) chain += std::log2((double)ctx->q[i]);
  if (chain < TB + 1.5) return pl;
  const int maxp = (compiled && logQ == 1024) ? T32_MAXP : 62;
  // Primes of 30 bits keep M above 2^(TB + 3.5) (rounds 2-5).  Primes below 2^29 (option tensor_bits = 29) make do with 2^(TB + 1.5): |x / M| < 0.36
  // and the fixed-point sum behind kappa is short by less than NP 2^-25, so round(sum y_i / p_i) is still kappa -- 36 primes at the metric ring
  // (1043.5 bits for TB = 1042), 72 at the stress ring.
  const int pb = ctx->opt.tensor_bits == 29 ? 29 : 30;
  const double margin = pb == 29 ? 1.5 : 3.5;
  double have = 0;
  int np = 0;
  for (u64 k = ((u64)1 << (pb - 1 - lg)) - 1; k > ((u64)1 << (pb - 2 - lg)) && have < TB + margin; --k) {
    const u64 cand = (k << (lg + 1)) + 1;
    if (cand > ((u64)1 << pb) - ((u64)1 << 15) + 1 || !hm::is_prime(cand)) continue;
    if (np == maxp) return pl;
    if (primes) primes->push_back((u32)cand);
    have += std::log2((double)cand);
    ++np;
  }
  if (have < TB + margin) return pl;
  // all of M inside the table row
  const double room = compiled ? (logQ == 512 ? (double)T32_R_512 * T32_WT_512 : (double)T32_R_1024 * T32_WT_1024) : 28.0 * ((have + 8) / 28 + 2);
  if (have > room - 8) return pl;
  pl.NP = np;
  pl.generic = !compiled;
  return pl;
}
bool tensor32_applies(const fhesi_ctx* ctx, u64 p, int nlimbs, int logQ) { return t32_plan(ctx, p, nlimbs, logQ, 1, nullptr).NP > 0; }
bool tensor32_sum_applies(const fhesi_ctx* ctx, u64 p, int nlimbs, int logQ, i64 gmax) {
  return t32_plan(ctx, p, nlimbs, logQ, gmax, nullptr).NP > 0;
}

// transform tables of the first `want` primes (grown on demand; a growth waits for the streams)
static int t32_ring(fhesi_ctx* ctx, const std::vector<u32>& primes) {
  fhesi_tensor32* x = ctx->tensor32;
  const int pb = !primes.empty() && primes[0] < (1u << 29) ? 29 : 30;
  if (x && x->bits != pb) {              // (option tensor_bits toggled on a live context: other primes, other tables)
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->lane_stream) HIP_TRY(hipStreamSynchronize(ctx->lane_stream));
    tensor32_free(ctx);
    x = nullptr;
  }
  if (!x) { x = new fhesi_tensor32(); x->S = (ctx->lin_q ? ctx->lin_lg : ctx->logn) - A32_LOGN; x->bits = pb; ctx->tensor32 = x; }
  if (x->primes.size() >= primes.size()) return 0;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (ctx->lane_stream) HIP_TRY(hipStreamSynchronize(ctx->lane_stream));
  const int S = x->S, lg = A32_LOGN + S, NP = (int)primes.size(), NS = 1 << S;
  if (S < 0 || S > 6) FHESI_FAIL("tensor32: rows of 2^%d", lg);
  const int HSN = A32_HSN(S);
  std::vector<Tw32> hs(S >= 3 ? (size_t)NP * 2 * HSN : 0, Tw32{0, 0});
  const i64 n = (i64)A32_N << S;
  const size_t per_prime = (size_t)A32_N << S;
  std::vector<Tw32> hf((size_t)NP * per_prime, Tw32{0, 0}), hi((size_t)NP * per_prime, Tw32{0, 0}), ff((size_t)n), fi((size_t)n), ht((size_t)NP * A32_HT, Tw32{0, 0});
  x->head.assign(NP, Tw32{0, 0}); x->tail.assign(NP, Tw32{0, 0}); x->head1.assign((size_t)NP * 2, Tw32{0, 0});
  for (int a = 0; a < NP; ++a) {
    const u64 p = primes[a];
    u64 psi = 0;
    for (u64 gq = 2; gq < 1000 && !psi; ++gq) {
      const u64 cand = hm::powmod(gq, (p - 1) / (2 * (u64)n), p);
      if (hm::powmod(cand, (u64)n, p) == p - 1) psi = cand;
    }
    if (!psi) FHESI_FAIL("tensor32: no 2n-th root");
    const u64 ipsi = hm::invmod(psi, p);
    auto tw = [&](u64 w) { return Tw32{(u32)w, (u32)((w << 32) / p)}; };
    auto fwd_form = [](Tw32 t) { return Tw32{0u - t.w, t.wp}; };      // what a32_ct<NEGW> takes: (-w mod 2^32, floor(w 2^32 / p)), as aux32_init
    a32_bitrev_powers(p, psi, lg, ff); a32_bitrev_powers(p, ipsi, lg, fi);       // the full table of the n-point transform: psi^brv(idx)
    if (!S) {
      std::transform(ff.begin(), ff.end(), hf.begin() + (size_t)a * per_prime, fwd_form);
      std::copy(fi.begin(), fi.end(), hi.begin() + (size_t)a * per_prime);
    } else {
      // sub-block h runs stage s >= 1 of the row on its groups i = h 2^(s-1) + i':  own index m' + i' (m' = 2^(s-1))  <->  2 m' + h m' + i'  (as aux32_init)
      for (int h = 0; h < NS; ++h)
        for (u64 mp = 1; mp < (u64)A32_N; mp <<= 1)
          for (u64 ip = 0; ip < mp; ++ip) {
            hf[((size_t)a * NS + h) * A32_N + mp + ip] = fwd_form(ff[(mp << S) + h * mp + ip]);
            hi[((size_t)a * NS + h) * A32_N + mp + ip] = fi[(mp << S) + h * mp + ip];
          }
      x->head[a] = ff[1]; x->tail[a] = fi[1];
      if (S == 2) {
        x->head1[(size_t)a * 2] = ff[2]; x->head1[(size_t)a * 2 + 1] = ff[3];
        const u64 inv2 = (p + 1) / 2;
        Tw32* c = &ht[(size_t)a * A32_HT];
        c[0] = ff[1]; c[1] = ff[2]; c[2] = ff[3];
        c[4] = tw(inv2); c[5] = tw(hm::mulmod(fi[2].w, inv2, p)); c[6] = tw(hm::mulmod(fi[3].w, inv2, p)); c[7] = tw(hm::mulmod(fi[1].w, inv2, p));
      }
      if (S >= 3) {          // (as aux32_init)
        Tw32* f = &hs[(size_t)a * 2 * HSN];
        Tw32* b = f + HSN;
        for (int idx = 1; idx < NS; ++idx) { f[idx] = ff[idx]; b[idx] = fi[idx]; }
        const u64 c = hm::invmod((u64)NS % p, p), cm = hm::mulmod(c, ((u64)1 << 32) % p, p);
        b[NS] = tw(c); b[NS + 1] = tw(hm::mulmod(fi[1].w, c, p)); b[NS + 2] = tw(cm); b[NS + 3] = tw(hm::mulmod(fi[1].w, cm, p));
      }
    }
  }
  // the inverse transform's last stage carries the final scaling (ntt32_inv_kernel3): entries 0 and 1 of every (prime, sub-block) table
  // become 1/n and w / n, w = that sub-block's distance-16 twiddle (n = 2^14: the sub-transform's length)
  for (size_t a = 0; a < primes.size(); ++a) {
    const u64 p = primes[a], ninv = hm::invmod((u64)A32_N % p, p);
    for (int h = 0; h < NS; ++h) {
      Tw32* t0 = &hi[(a * NS + h) * A32_N];
      const u64 wn = hm::mulmod(t0[1].w, ninv, p);
      t0[0] = Tw32{(u32)ninv, (u32)((ninv << 32) / p)}; t0[1] = Tw32{(u32)wn, (u32)((wn << 32) / p)};
    }
  }
  hipFree(x->d_fwd); hipFree(x->d_inv); hipFree(x->d_ht); hipFree(x->d_p); hipFree(x->d_hs);
  x->d_fwd = x->d_inv = x->d_ht = x->d_hs = nullptr; x->d_p = nullptr;
  if (S >= 2) {
    if (hipMalloc(&x->d_ht, ht.size() * sizeof(Tw32)) != hipSuccess || hipMalloc(&x->d_p, (size_t)NP * sizeof(u32)) != hipSuccess) FHESI_FAIL("tensor32: hipMalloc failed");
    HIP_TRY(hipMemcpy(x->d_ht, ht.data(), ht.size() * sizeof(Tw32), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(x->d_p, primes.data(), (size_t)NP * sizeof(u32), hipMemcpyHostToDevice));
  }
  if (S >= 3) {
    if (hipMalloc(&x->d_hs, hs.size() * sizeof(Tw32)) != hipSuccess) FHESI_FAIL("tensor32: hipMalloc failed");
    HIP_TRY(hipMemcpy(x->d_hs, hs.data(), hs.size() * sizeof(Tw32), hipMemcpyHostToDevice));
  }
  a32_permute_phase_c(hf); a32_permute_phase_c(hi);      // (the last four stages' twiddles in the order the waves load them: A32_TWC)
  if (hipMalloc(&x->d_fwd, hf.size() * sizeof(Tw32)) != hipSuccess || hipMalloc(&x->d_inv, hi.size() * sizeof(Tw32)) != hipSuccess) FHESI_FAIL("tensor32: hipMalloc failed");
  HIP_TRY(hipMemcpy(x->d_fwd, hf.data(), hf.size() * sizeof(Tw32), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(x->d_inv, hi.data(), hi.size() * sizeof(Tw32), hipMemcpyHostToDevice));
  x->primes = primes;
  return 0;
}

static int t32_config(fhesi_ctx* ctx, u64 lift, int nlimbs, int logQ, i64 gmax, T32Config** out) {
  std::vector<u32> primes;
  const T32Plan pl = t32_plan(ctx, lift, nlimbs, logQ, gmax, &primes);
  if (!pl.NP) FHESI_FAIL("tensor32: shape not supported");
  FHESI_TRY(t32_ring(ctx, primes));
  fhesi_tensor32* x = ctx->tensor32;
  for (T32Config* c : x->cfgs)
    if (c->lift == lift && c->nl == nlimbs && c->logQ == logQ && c->NP == pl.NP && c->generic == pl.generic) { *out = c; return 0; }
  const int NP = pl.NP, S = x->S;
  double have = 0;
  for (int a = 0; a < NP; ++a) have += std::log2((double)primes[a]);
  T32Config* c = new T32Config();
  c->lift = lift; c->nl = nlimbs; c->logQ = logQ; c->NP = NP; c->generic = pl.generic;
  c->R = pl.generic ? 28 : (logQ == 512 ? T32_R_512 : T32_R_1024);
  c->WT = pl.generic ? (int)((have + 8) / 28) + 2 : (logQ == 512 ? T32_WT_512 : T32_WT_1024);
  const int R = c->R, WT = c->WT, stride = (2 * nlimbs + 8 + 7) & ~7;          // (rows of whole 32-byte lines: the kernel reads them with scalar loads)
  std::vector<u32> rns((size_t)2 * NP * stride);
  for (int a = 0; a < NP; ++a) {
    const u64 p = primes[a];
    const u64 ninv = hm::invmod((u64)A32_N % p, p);     // of the 2^14-point (sub-)transform; the tail carries the other 1/2
    c->pr.p[a] = (u32)p;
    c->pr.ninv[a] = (u32)ninv;
    c->pr.ninv_p[a] = (u32)((ninv << 32) / p);
    c->pr.mu61[a] = (u32)(((u64)1 << (32 + t32_shift((u32)p))) / p);
    for (int cls = 0; cls < 2; ++cls) {
      u32* e = &rns[((size_t)cls * NP + a) * stride];
      const u64 b32 = ((u64)1 << 32) % p;
      u64 cur = cls == 0 ? lift % p : 1;
      for (int k = 0; k < 2 * nlimbs; ++k) { e[k] = (u32)cur; cur = hm::mulmod(cur, b32, p); }
      e[2 * nlimbs] = (u32)((p - cur) % p);          // two's complement: value = unsigned - 2^(64 nl)
      e[2 * nlimbs + 1] = (u32)b32;
      e[2 * nlimbs + 2] = c->pr.mu61[a];
      e[2 * nlimbs + 3] = (u32)p;
      // head stage of a 2^15-point row: psi^brv(1); padded rows of 2^16: the second head stage's psi^brv(2), psi^brv(3) (the first is a duplication)
      e[2 * nlimbs + 4] = S == 2 ? x->head1[(size_t)a * 2].w : (S ? x->head[a].w : 0);
      e[2 * nlimbs + 5] = S == 2 ? x->head1[(size_t)a * 2].wp : (S ? x->head[a].wp : 0);
      e[2 * nlimbs + 6] = S == 2 ? x->head1[(size_t)a * 2 + 1].w : 0;
      e[2 * nlimbs + 7] = S == 2 ? x->head1[(size_t)a * 2 + 1].wp : 0;
    }
  }
  // CRT tables
  Big M{1};
  for (int a = 0; a < NP; ++a) M = big_mul_small(M, primes[a]);
  std::vector<Tw32> cinv((size_t)NP * 2);
  std::vector<u32> inv57(NP), Mw((size_t)(NP + 1) * WT + 64);        // (padded: the generic kernel multiplies a fixed number of words per row, whatever the window)
  for (int a = 0; a < NP; ++a) {
    Big Mi{1};
    for (int b = 0; b < NP; ++b) if (b != a) Mi = big_mul_small(Mi, primes[b]);
    const u64 p = primes[a];
    const u64 ci = hm::invmod(big_mod_small(Mi, p), p);
    auto tw = [&](u64 w) { return Tw32{(u32)w, (u32)((w << 32) / p)}; };
    if (S != 1) cinv[(size_t)a * 2] = cinv[(size_t)a * 2 + 1] = tw(ci);      // (rows of 2^16: the tail stages are a pass of their own, ntt32_tail2_kernel)
    else {
      const u64 inv2 = (p + 1) / 2;
      cinv[(size_t)a * 2] = tw(hm::mulmod(ci, inv2, p));
      cinv[(size_t)a * 2 + 1] = tw(hm::mulmod(ci, hm::mulmod(x->tail[a].w, inv2, p), p));
    }
    inv57[a] = (u32)(((u64)1 << 57) / p);
    for (int l = 0; l < WT; ++l) Mw[(size_t)a * WT + l] = big_word(Mi, l, R);
  }
  {
    // 2^(R WT) - M: two's complement of M over enough limbs, read through the same word extraction (masked to R WT bits)
    Big N(M);
    N.resize((R * WT + 63) / 64 + 1, 0);
    u64 carry = 1;
    for (auto& w : N) { const u64 v = ~w + carry; carry = (carry && v == 0) ? 1 : 0; w = v; }
    for (int l = 0; l < WT; ++l) Mw[(size_t)NP * WT + l] = big_word(N, l, R);
  }
  auto up = [&](void** d, const void* h, size_t bytes) -> bool {
    return hipMalloc(d, bytes) == hipSuccess && hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice) == hipSuccess;
  };
  if (!up((void**)&c->d_rns, rns.data(), rns.size() * 4) || !up((void**)&c->d_cinv, cinv.data(), cinv.size() * sizeof(Tw32)) ||
      !up((void**)&c->d_inv57, inv57.data(), inv57.size() * 4) || !up((void**)&c->d_Mw, Mw.data(), Mw.size() * 4)) {
    delete c;
    FHESI_FAIL("tensor32: table upload failed");
  }
  x->cfgs.push_back(c);
  *out = c;
  return 0;
}

// ---------------------------------------------------------------------------------------------- big integer -> residues
// PAIRED: a, b: [count][2][n_src][NL] two's complement coefficients; rows [count][4][NP][nrow] (a0, a1, b0, b1).
// otherwise: na2 polynomials of a (class 0: lifted by p), then those of b; rows [polys][NP][nrow].  Values below 4p; positions from
// n_src upwards (linear-convolution rings) are zero.
// HEAD (rows of 2^15): a thread takes coefficients j and j + 2^14 and stores  x + w y  and  x + 2p - w y  into sub-rows 0 and 1 of the row
// (the head stage of the 2^15-point transform, w = psi^brv(1)).
// a * b_uniform + c in one v_mad_u64_u32, the wave-uniform factor taken from its scalar register
__device__ __forceinline__ u64 mad64s(u32 a, u32 b_uniform, u64 c) {
  u64 r, carry;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(carry) : "v"(a), "s"(b_uniform), "v"(c));
  return r;
}
template <int NL>
__device__ __forceinline__ u32 rns32_one(const u32 (&x)[2 * NL], u32 neg, const u32* __restrict__ t) {
  const u32 r32 = t[2 * NL + 1], mu = t[2 * NL + 2], p = t[2 * NL + 3];
  // products below 2^62 - 2^47: four fit a 64-bit accumulator, three on top of a folded value (below 2^62 + 2^32)
  u64 acc = 0;
  int room = 4;
  // ONE dependency chain of v_mad_u64_u32 (the table words from their scalar registers): written as plain C++ the compiler splits the sum into
  // five independent groups and joins them with 64-bit adds and moves -- 38 instructions per prime where the chain needs 32; the latency
  // of the chain is covered by the other waves (34 registers: full occupancy)
#pragma unroll
  for (int k = 0; k < 2 * NL; ++k) {
    if (room == 0) { acc = mad64s((u32)(acc >> 32), r32, (u64)(u32)acc); room = 3; }
    acc = mad64s(x[k], t[k], acc);
    --room;
  }
  acc = mad64s(neg, t[2 * NL], acc);                          // (neg is 0 or 1: the two's complement correction without a branch)
  acc = mad64s((u32)(acc >> 32), r32, (u64)(u32)acc);       // below 2^32 r32 + 2^32
  // The primes of the tensor half are the largest below 2^30, so 2^32 mod p = 4 (2^30 - p) is a small number (below 2^26 for the 72 largest
  // primes that are 1 mod 2^16): one fold already leaves the total below 2^61.  Any other prime takes the second fold (a wave-uniform branch).
  // (primes of 29 bits: sh = 28 -- one fold leaves the total below 2^60 when 2^32 mod p = 8 (2^29 - p) is below 2^27, the second fold otherwise)
  const u32 sh = t32_shift(p);
  if (r32 >= (1u << (sh - 1))) acc = mad64s((u32)(acc >> 32), r32, (u64)(u32)acc);       // below 2^(32 + sh)
  const u32 q = __umulhi((u32)(acc >> sh), mu);        // at most 2 below floor(acc / p)
  return (u32)acc - q * p;                             // below 3p
}
// HEAD: 0 = plain rows (or, dup, padded rows of 2^15: the value goes to both sub-rows); 1 = rows of 2^15 with the head stage x +- w y of
// coefficients j, j + 2^14; 2 = PADDED rows of 2^16 (fewer than 2^15 coefficients): the first head stage is a duplication, the second the same
// x +- w y with the half's twiddle -- sub-rows 0, 1: psi^brv(2), sub-rows 2, 3: psi^brv(3); coefficients from n_src upwards are zero
template <int NL, int HEAD, bool PAIRED>
__global__ void __launch_bounds__(256) rns32_reduce_kernel(const u64* __restrict__ a, const u64* __restrict__ b, i64 na2, i64 n_src, i64 nrow, u32* __restrict__ rows, int NP,
                                                            const u32* __restrict__ tab, const int* __restrict__ idx_a = nullptr, const int* __restrict__ idx_b = nullptr, int dup = 0) {
  __shared__ __attribute__((aligned(16))) u64 sl[NL * 256];       // [NL][256]
  const i64 poly = blockIdx.y;
  int cls;
  const u64* __restrict__ src;
  if (PAIRED) {      // (idx_a / idx_b: operand ct of the a's / b's is ciphertext idx[ct] of the buffer -- pool indices of a wave of single products)
    const i64 ct = poly >> 2; const int jp = (int)(poly & 3); cls = jp >> 1;
    const i64 cs = idx_a ? (i64)(cls ? idx_b : idx_a)[ct] : ct;
    src = (cls ? b : a) + (cs * 2 + (jp & 1)) * n_src * NL;
  }
  else { cls = poly >= na2; src = cls ? b + (poly - na2) * n_src * NL : a + poly * n_src * NL; }
  const i64 j0 = (i64)blockIdx.x * 256;
  const int tid = threadIdx.x;
  u32* __restrict__ o = rows + poly * NP * nrow + j0 + tid;
  // dup (padded rows of 2^15 on a linear-convolution ring: the polynomial's upper half is zero, so the head stage x +- w 0 leaves x in both
  // sub-rows): every value is written to sub-row 1 as well
  // few polynomials (small batches): the primes are split over gridDim.z so that the launch fills the chip -- with one workgroup per CU
  // every thread walked the 35 primes behind 35 exposed scalar-load latencies (21 us for one ciphertext)
  const int i_first = (int)((i64)blockIdx.z * NP / gridDim.z), i_last = (int)((i64)(blockIdx.z + 1) * NP / gridDim.z);
  if (j0 >= n_src) {                                             // (whole block in the zero padding)
    // (every sub-row this block owns: a ring needing rows of 2^16 has more than 2^14 coefficients and never comes here with HEAD = 2, the
    // FHESI_LIN_LG test hook on a small ring does)
    for (int i = i_first; i < i_last; ++i) {
      o[(i64)i * nrow] = 0;
      if (dup || HEAD != 0) o[(i64)i * nrow + A32_N] = 0;
      if constexpr (HEAD == 2) { o[(i64)i * nrow + 2 * A32_N] = 0; o[(i64)i * nrow + 3 * A32_N] = 0; }
    }
    return;
  }
  const i64 avail = (n_src - j0) * NL;                           // words of this block's 256 coefficients that exist
  u32 x[2 * NL], x1[HEAD != 0 ? 2 * NL : 1];
  const u64* __restrict__ sblk = src + j0 * NL;
  if (avail >= 256 * NL) {                                       // (uniform: every block but the last one of a padded row)
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      const int e = it * 256 + tid;
      sl[(e % NL) * 256 + e / NL] = sblk[e];
    }
  } else {
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      const int e = it * 256 + tid;
      sl[(e % NL) * 256 + e / NL] = e < avail ? sblk[e] : 0;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NL; ++k) { const u64 v = sl[k * 256 + tid]; x[2 * k] = (u32)v; x[2 * k + 1] = (u32)(v >> 32); }
  if constexpr (HEAD != 0) {
    __syncthreads();
    const i64 avail1 = HEAD == 2 ? (n_src - (j0 + A32_N)) * NL : (i64)256 * NL;      // (padded rows: the second block may lie partly or wholly in the padding)
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      const int e = it * 256 + tid;
      sl[(e % NL) * 256 + e / NL] = e < avail1 ? src[(j0 + A32_N) * NL + e] : 0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NL; ++k) { const u64 v = sl[k * 256 + tid]; x1[2 * k] = (u32)v; x1[2 * k + 1] = (u32)(v >> 32); }
  }
  const u32 neg = x[2 * NL - 1] >> 31;
  u32 neg1 = 0;
  if constexpr (HEAD != 0) neg1 = x1[2 * NL - 1] >> 31;
  constexpr int STRIDE = (2 * NL + 8 + 7) & ~7;
#pragma unroll 2
  for (int i = i_first; i < i_last; ++i) {
    const u32* __restrict__ t = tab + ((i64)cls * NP + i) * STRIDE;
    const u32 r0 = rns32_one<NL>(x, neg, t);
    if constexpr (HEAD == 0) {
      __builtin_nontemporal_store(r0, &o[(i64)i * nrow]);                  // (a zero coefficient gives 0: the padding inside a partial block)
      if (dup) __builtin_nontemporal_store(r0, &o[(i64)i * nrow + A32_N]);
    }
    else {
      const u32 p = t[2 * NL + 3], twop = 2 * p;
      const u32 r1 = rns32_one<NL>(x1, neg1, t);
      const u32 X = r0 >= twop ? r0 - twop : r0;
      const u32 T = mul_lazy32(r1, Tw32{t[2 * NL + 4], t[2 * NL + 5]}, p);
      o[(i64)i * nrow] = X + T;
      o[(i64)i * nrow + A32_N] = X + twop - T;
      if constexpr (HEAD == 2) {
        const u32 Tb = mul_lazy32(r1, Tw32{t[2 * NL + 6], t[2 * NL + 7]}, p);
        o[(i64)i * nrow + 2 * A32_N] = X + Tb;
        o[(i64)i * nrow + 3 * A32_N] = X + twop - Tb;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- sums of tensor products, evaluation form
//   out[g][0..2][prime][:] = sum_{t in [seg[g], seg[g+1])} (a0 b0, a0 b1 + a1 b0, a1 b1)  mod p   with a = ra[slot_a[t]], b = rb[slot_b[t]]
// ra [nua][2][NP][nrow], rb [nub][2][NP][nrow]: forward transforms (below p); out below 2p (what the inverse transform takes)
__global__ void __launch_bounds__(256) tensor_sum32_kernel(const u32* __restrict__ ra, const u32* __restrict__ rb, const int* __restrict__ slot_a,
                                                           const int* __restrict__ slot_b, const int* __restrict__ seg, int accumulate, u32* __restrict__ out,
                                                           i64 nrow, int NP, T32Primes pr) {
  const int g = blockIdx.z, l = blockIdx.y;
  const u32 p = pr.p[l], mu = pr.mu61[l], twop = 2 * p, sh = t32_shift(p);
  const u32 r32 = (u32)((((u64)1) << 32) % p);
  const i64 rs = (i64)NP * nrow;
  const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  u32* o0 = out + (((i64)g * 3 + 0) * NP + l) * nrow + j;
  const int t0 = seg[g], t1 = seg[g + 1];
  // a 64-bit total takes 8 products below 2^60; reduced to below 2p every 4 terms (t1 adds two products per term)
  auto red = [&](u64 v) -> u32 {
    v = (u64)(u32)(v >> 32) * r32 + (u32)v;          // below 2^62 + 2^32
    v = (u64)(u32)(v >> 32) * r32 + (u32)v;          // below 2^61 (primes of 29 bits: below 2^59)
    const u32 q = __umulhi((u32)(v >> sh), mu);
    const u32 r = (u32)v - q * p;                     // below 3p
    return r >= twop ? r - twop : r;
  };
  u64 r0 = 0, r1 = 0, r2 = 0;
  if (accumulate) { r0 = o0[0]; r1 = o0[rs]; r2 = o0[2 * rs]; }
  // four terms per round: their sixteen loads are issued together (the kernel waits on its loads), then one reduction
  int t = t0;
  for (; t + 4 <= t1; t += 4) {
    u32 a0[4], a1[4], b0[4], b1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const u32* a = ra + (((i64)slot_a[t + u] * 2) * NP + l) * nrow + j;
      const u32* b = rb + (((i64)slot_b[t + u] * 2) * NP + l) * nrow + j;
      a0[u] = a[0]; a1[u] = a[rs]; b0[u] = b[0]; b1[u] = b[rs];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      r0 += (u64)a0[u] * b0[u];
      r1 += (u64)a0[u] * b1[u] + (u64)a1[u] * b0[u];
      r2 += (u64)a1[u] * b1[u];
    }
    r0 = red(r0); r1 = red(r1); r2 = red(r2);
  }
  for (; t < t1; ++t) {                               // at most three more
    const u32* a = ra + (((i64)slot_a[t] * 2) * NP + l) * nrow + j;
    const u32* b = rb + (((i64)slot_b[t] * 2) * NP + l) * nrow + j;
    const u32 a0 = a[0], a1 = a[rs], b0 = b[0], b1 = b[rs];
    r0 += (u64)a0 * b0;
    r1 += (u64)a0 * b1 + (u64)a1 * b0;
    r2 += (u64)a1 * b1;
  }
  o0[0] = red(r0);
  o0[rs] = red(r1);
  o0[2 * rs] = red(r2);
}

// ---------------------------------------------------------------------------------------------- residues -> round(x / 2^logQ) mod 2^logQ
// rows [npolys][NP][n] (below p) -> out [npolys][LQ/64][n] limb-major positive residues (crt mode 1, kernels_crt.hip).
// S = 1: the rows are the two sub-inverses of 2^15-point rows; coefficient j < 2^14 is (A_j + B_j) / 2, coefficient j + 2^14 is
// (A_j - B_j) psi^-brv(1) / 2 -- the constants are folded into the CRT constant of the first multiplication.
template <int LQ, bool EXACT, int R, int WT, int S>
__global__ void __launch_bounds__(128) crt32_scale_kernel(const u32* __restrict__ rows, i64 n, int NP, T32Primes pr, const Tw32* __restrict__ cinv,
                                                           const u32* __restrict__ inv57, const u32* __restrict__ Mw, u64* __restrict__ out,
                                                           unsigned char* __restrict__ flags, int wm /* 1: out as 32-bit WORD rows [npolys][LQ/32][n] (what the 32-bit digit loader reads whole lines of) */) {
  static_assert((LQ & 63) == 0, "logQ a multiple of 64");
  constexpr int WU = (2 * LQ + R - 1) / R;            // words that reach below bit 2 logQ
  constexpr int J0 = EXACT ? 0 : (LQ - 64 - 30 - 8) / R;     // first word formed: R J0 + 30 + log2(NP + 1) + 1 <= logQ - 64
  constexpr int NW = WU - J0;
  static_assert(WU <= WT && J0 >= 0, "window");
  const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  if (EXACT && !flags[wg]) return;
  const i64 poly = blockIdx.y;
  const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  const int hi = S ? (int)(j >> A32_LOGN) : 0;        // (uniform per workgroup)
  const u32* __restrict__ src = rows + poly * NP * n + (S ? (j & (A32_N - 1)) : j);
  u64 acc[NW];
#pragma unroll
  for (int l = 0; l < NW; ++l) acc[l] = 0;
  u32 fsum = 0;
#pragma unroll 2
  for (int i = 0; i < NP; ++i) {
    const u32 p = pr.p[i];
    u32 r = src[(i64)i * n];
    if (S) { const u32 B = src[(i64)i * n + A32_N]; r = hi ? r + p - B : r + B; }
    u32 y = mul_lazy32(r, cinv[2 * i + hi], p);
    y = y >= p ? y - p : y;
    fsum += __umulhi(y, inv57[i]);                    // (y / p) 2^25, low by less than one unit
    const u32* __restrict__ Mi = Mw + (i64)i * WT + J0;
#pragma unroll
    for (int l = 0; l < NW; ++l) acc[l] += (u64)y * Mi[l];
  }
  // x/M is within 1/8 of an integer: kappa = round(sum y_i / p_i); x = sum - kappa M = sum + kappa (2^(R WT) - M)  (mod 2^(R WT))
  const u32 kappa = (fsum + (1u << 24)) >> 25;
  {
    const u32* __restrict__ Nm = Mw + (i64)NP * WT + J0;
#pragma unroll
    for (int l = 0; l < NW; ++l) acc[l] += (u64)kappa * Nm[l];
  }
  // carries: words of R bits
  u64 carry = 0;
#pragma unroll
  for (int l = 0; l < NW; ++l) { const u64 v = acc[l] + carry; acc[l] = v & (((u64)1 << R) - 1); carry = v >> R; }
  // 64 bits from bit B of x (two's complement, bits above R WU dropped)
  auto limb = [&](int B) -> u64 {
    const int l0 = B / R - J0, o = B % R;
    u64 v = acc[l0] >> o;
    if (l0 + 1 < NW) v |= acc[l0 + 1] << (R - o);
    if (l0 + 2 < NW) v |= acc[l0 + 2] << (2 * R - o);
    if (l0 + 3 < NW && 3 * R - o < 64) v |= acc[l0 + 3] << (3 * R - o);
    return v;
  };
  const u64 G = limb(LQ - 64);                         // bits logQ-64 .. logQ-1
  int undecided = 0;
  if (!EXACT) undecided = (G == 0x7fffffffffffffffull) ? 1 : 0;
  u64 c = G >> 63;                                     // round half up: + bit logQ-1
  u64* __restrict__ o = out + poly * (LQ / 64) * n + j;
  u32* __restrict__ o32 = reinterpret_cast<u32*>(out) + poly * (LQ / 32) * n + j;
#pragma unroll
  for (int i = 0; i < LQ / 64; ++i) {
    u64 v = limb(LQ + 64 * i);
    v += c;
    c = (c && v == 0) ? 1 : 0;
    if (wm) { o32[(i64)(2 * i) * n] = (u32)v; o32[(i64)(2 * i + 1) * n] = (u32)(v >> 32); }
    else o[(i64)i * n] = v;
  }
  if (!EXACT) {
    const int any = __syncthreads_or(undecided);
    if (threadIdx.x == 0) flags[wg] = any ? 1 : 0;
  }
}

// The same conversion for any logQ <= 512 (words of 28 bits, rows of 2^14) and for the linear-convolution rings: the window [J0, J0 + NW)
// of words and the bit positions are run-time values, so after the (compile-time indexed) accumulation and the carry pass the words go
// through LDS, one column per thread, and the 64-bit limbs are cut from there.  fold_q = q' (m = 2q'): the residue of output coefficient j is
//   r_j - r_(j+q') - (-1)^j r_(q'-1)   of the linear product's residues (modulo X^q' + 1, then modulo Phi_m = sum (-X)^i).
// NWMAX = T32_GEN_NW with J0 from the host, or T32_GEN_NWX with J0 = 0 for the exact pass over flagged workgroups.
// S = 1 (rows of 2^15 = the two sub-inverses A, B of ntt32_inv_kernel3): the coefficient at position e < 2^14 is (A_e + B_e) / 2, at e + 2^14
// (A_e - B_e) psi^-brv(1) / 2; the two constants sit in cinv[2i], cinv[2i + 1] (times the CRT constant), so the positions of the lower and of
// the upper half are summed separately and the residue is  lo c_lo + hi c_hi.
// S and FOLD (0: none, 1: m = 2q', 2: m prime) are compile-time: the (up to) three positions a coefficient is folded from do not depend on the
// prime, so their offsets, halves and signs are worked out once, the loads of a prime are unconditional (clamped index, masked value) and
// independent of each other, and the multiply-adds run over all NWMAX words without a branch (words from NW upwards are never looked at; the
// table is padded so that the reads stay inside it).  With the positions behind run-time branches and `if (l < NW)` around every word the
// kernel waited on one scalar and one vector load after the other: 13.5 ms per 1024 multiplications on the reference's ring at the metric's
// size against 2.5 ms for the compiled shape on half as long rows.
template <int NWMAX, bool EXACT, int S, int FOLD>
__global__ void __launch_bounds__(128) crt32_scale_generic_kernel(const u32* __restrict__ rows, i64 nrow, i64 n_out, i64 fold_off, int NP, int LQ, int J0, int NW, int WT,
                                                                   T32Primes pr, const Tw32* __restrict__ cinv, const u32* __restrict__ inv57, const u32* __restrict__ Mw,
                                                                   u64* __restrict__ out, unsigned char* __restrict__ flags, int wm) {
  constexpr int R = 28;
  __shared__ u32 xs[NWMAX * 128];
  const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  if (EXACT && !flags[wg]) return;
  const i64 poly = blockIdx.y;
  const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = j < n_out;
  const u32* __restrict__ src = rows + poly * NP * nrow;
  u64 acc[NWMAX];
#pragma unroll
  for (int l = 0; l < NWMAX; ++l) acc[l] = 0;
  u32 fsum = 0;
  if (active) {
    // term 0: position j, +;  term 1: position j + off, - (m = 2q') or + (m prime);  term 2: position off - 1, -(-1)^j (m = 2q') or - (m prime)
    constexpr int NT = FOLD ? 3 : 1;
    u32 eb[NT];
    bool up[NT], ok[NT], neg[NT];
    {
      const i64 e[3] = {j, j + fold_off, fold_off - 1};
      const bool ng[3] = {false, FOLD == 1, FOLD == 2 || !(j & 1)};
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        ok[k] = e[k] < nrow;
        const u32 idx = ok[k] ? (u32)e[k] : 0u;
        up[k] = S && idx >= (u32)A32_N;
        eb[k] = S ? (idx & (u32)(A32_N - 1)) : idx;
        neg[k] = ng[k];
      }
    }
#pragma unroll 2
    for (int i = 0; i < NP; ++i) {
      const u32 p = pr.p[i], twop = 2 * p;
      const u32* __restrict__ ri = src + (i64)i * nrow;
      u32 y;
      if (S) {
        // position e of the row: lower half (A_e + B_e), upper half (A - B)(e - 2^14), each below 2p, summed per half with its sign
        u32 A[NT], B[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k) { A[k] = ri[eb[k]]; B[k] = ri[eb[k] + A32_N]; }
        auto red2 = [&](u32 v) -> u32 { return min(v, v - twop); };                 // [0, 4p) -> [0, 2p)
        u32 lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < NT; ++k) {
          const u32 t = up[k] ? A[k] + p - B[k] : A[k] + B[k];                       // at most 2p
          const u32 v = ok[k] ? (neg[k] ? twop - t : t) : 0u;
          if (k == 0) { lo = up[0] ? 0u : v; hi = up[0] ? v : 0u; }
          else { lo = red2(lo + (up[k] ? 0u : v)); hi = red2(hi + (up[k] ? v : 0u)); }
        }
        if (NT == 1) { lo = red2(lo); hi = red2(hi); }
        y = mul_lazy32(lo, cinv[2 * i], p) + mul_lazy32(hi, cinv[2 * i + 1], p);    // below 4p
        y = min(y, y - twop);
        y = min(y, y - p);
      } else {
        u32 r = ri[eb[0]];
        if (FOLD) {
          const u32 b = ok[1] ? ri[eb[1]] : 0u, c = ri[eb[2]];
          if (FOLD == 1) r = r + (p - b) + (neg[2] ? p - c : c);                    // r_j - r_(j+q') - (-1)^j r_(q'-1): below 4p
          else r = r + b + (p - c);                                                  // r_j + r_(j+m) - r_(m-1): below 3p
        }
        y = mul_lazy32(r, cinv[2 * i], p);
        y = y >= p ? y - p : y;
      }
      fsum += __umulhi(y, inv57[i]);
      const u32* __restrict__ Mi = Mw + (i64)i * WT + J0;
#pragma unroll
      for (int l = 0; l < NWMAX; ++l) acc[l] += (u64)y * Mi[l];
    }
    const u32 kappa = (fsum + (1u << 24)) >> 25;
    const u32* __restrict__ Nm = Mw + (i64)NP * WT + J0;
    u64 carry = 0;
#pragma unroll
    for (int l = 0; l < NWMAX; ++l)
      if (l < NW) {
        const u64 v = acc[l] + (u64)kappa * Nm[l] + carry;
        xs[l * 128 + threadIdx.x] = (u32)(v & 0xfffffffull);
        carry = v >> R;
      }
  }
  int undecided = 0;
  if (active) {
    // 64 bits from bit B of x (each thread reads back its own column)
    auto word = [&](int l) -> u64 { return (l >= 0 && l < NW) ? (u64)xs[l * 128 + threadIdx.x] : 0ull; };
    auto limb = [&](int B) -> u64 {
      const int l0 = B / R - J0, o = B % R;
      u64 v = word(l0) >> o;
      v |= word(l0 + 1) << (R - o);
      v |= word(l0 + 2) << (2 * R - o);
      if (3 * R - o < 64) v |= word(l0 + 3) << (3 * R - o);
      return v;
    };
    const u64 G = limb(LQ - 64);                         // bits logQ-64 .. logQ-1
    if (!EXACT) undecided = (G == 0x7fffffffffffffffull) ? 1 : 0;
    u64 c = G >> 63;                                     // round half up: + bit logQ-1
    const int nlq = (LQ + 63) >> 6;
    u64* __restrict__ o = out + poly * nlq * n_out + j;
    u32* __restrict__ o32 = reinterpret_cast<u32*>(out) + poly * (2 * nlq) * n_out + j;
    for (int i = 0; i < nlq; ++i) {
      u64 v = limb(LQ + 64 * i);
      v += c;
      c = (c && v == 0) ? 1 : 0;
      const int bits_left = LQ - 64 * i;
      if (bits_left < 64) v &= ((u64)1 << bits_left) - 1;
      if (wm) { o32[(i64)(2 * i) * n_out] = (u32)v; o32[(i64)(2 * i + 1) * n_out] = (u32)(v >> 32); }
      else o[(i64)i * n_out] = v;
    }
  }
  if (!EXACT) {
    const int any = __syncthreads_or(undecided);
    if (threadIdx.x == 0) flags[wg] = any ? 1 : 0;
  }
}

// ---------------------------------------------------------------------------------------------- launchers
static i64 t32_nrow(const fhesi_ctx* ctx) { return (i64)A32_N << ctx->tensor32->S; }
template <int NL>
static int t32_launch_rns(fhesi_ctx* ctx, const T32Config* c, const u64* d_a, const u64* d_b, i64 npolys, i64 na2, bool paired, u32* d_r) {
  const int S = ctx->tensor32->S;
  const i64 nrow = t32_nrow(ctx), n_src = ctx->phim;
  const unsigned zs = npolys * (A32_N / 256) >= 2048 ? 1u : (npolys * (A32_N / 256) >= 1024 ? 2u : 5u);      // (primes split over z for small launches)
  // (rows of 2^17 and longer -- S >= 3, the simple path -- are converted as plain zero-padded rows, one block per 256 positions of the whole row;
  // their head stages are a pass of their own in t32_fwd)
  const dim3 grid((unsigned)((S >= 3 ? nrow : (i64)A32_N) / 256), (unsigned)npolys, zs);
  const int* ia = paired && ctx->op_idx ? ctx->op_idx + ctx->op_idx_done : nullptr;
  const int* ib = ia ? ia + ctx->op_idx_n : nullptr;
  const int dup = S == 1 && ctx->lin_q ? 1 : 0;
  if (S >= 2 && !ctx->lin_q) FHESI_FAIL("tensor32: rows of 2^16 and longer exist for the padded linear-convolution rings only");
#define T32_GO(HEAD, PAIRED) do { PROF_KERNEL(ctx, PROF_RNS, rns32_reduce_kernel<NL, HEAD, PAIRED>); \
    rns32_reduce_kernel<NL, HEAD, PAIRED><<<grid, 256, 0, ctx->stream>>>(d_a, d_b, na2, n_src, nrow, d_r, c->NP, c->d_rns, ia, ib, dup); } while (0)
  if (S >= 3) { if (paired) T32_GO(0, true); else T32_GO(0, false); }
  else if (S == 2) { if (paired) T32_GO(2, true); else T32_GO(2, false); }
  else if (S && !dup) { if (paired) T32_GO(1, true); else T32_GO(1, false); }
  else { if (paired) T32_GO(0, true); else T32_GO(0, false); }
#undef T32_GO
  HIP_TRY(hipGetLastError());
  return 0;
}
static int t32_rns(fhesi_ctx* ctx, const T32Config* c, const u64* d_a, const u64* d_b, i64 npolys, i64 na2, bool paired, u32* d_r) {
  ProfScope prof(ctx, PROF_RNS, (double)npolys);
#define T32_RNS(NL) case NL: return t32_launch_rns<NL>(ctx, c, d_a, d_b, npolys, na2, paired, d_r);
  switch (c->nl) {
    T32_RNS(1) T32_RNS(2) T32_RNS(3) T32_RNS(4) T32_RNS(5) T32_RNS(6) T32_RNS(7) T32_RNS(8) T32_RNS(9) T32_RNS(10) T32_RNS(11) T32_RNS(12)
    T32_RNS(13) T32_RNS(14) T32_RNS(15) T32_RNS(16)
    default: break;
  }
#undef T32_RNS
  FHESI_FAIL("tensor32: %d limbs", c->nl);
}
static int t32_fwd(fhesi_ctx* ctx, const T32Config* c, u32* d_r, i64 npolys) {
  const fhesi_tensor32* x = ctx->tensor32;
  ProfScope prof(ctx, PROF_NTT_FWD, (double)(npolys * c->NP));
  if (npolys > 0x7fffffff) FHESI_FAIL("tensor32: too many rows per launch");
  const dim3 grid((unsigned)npolys, (unsigned)(c->NP << x->S));
#define T32_FWD_GO(SS, PB) do { PROF_KERNEL(ctx, PROF_NTT_FWD, (ntt32_fwd_kernel3<false, SS, false, T32Primes, true, false, PB>)); \
    ntt32_fwd_kernel3<false, SS, false, T32Primes, true, false, PB><<<grid, A32_T, 0, ctx->stream>>>(d_r, npolys, c->NP, 0, c->pr, x->d_fwd, Dig32Src{}, Aux32Head{}); } while (0)
  if (x->S >= 3) {       // head stages of the plain rows as a pass of their own (at most 65535 rows, a multiple of NP, per launch)
    const i64 nr = npolys * c->NP, step = (65535 / c->NP) * (i64)c->NP;
    for (i64 r0 = 0; r0 < nr; r0 += step) {
      const dim3 hg(A32_N / 256, (unsigned)std::min(step, nr - r0));
      u32* rp = d_r + (r0 << (A32_LOGN + x->S));
      if (x->S == 3) ntt32_headS_kernel<3><<<hg, 256, 0, ctx->stream>>>(rp, c->NP, 0, x->d_p, x->d_hs);
      else if (x->S == 4) ntt32_headS_kernel<4><<<hg, 256, 0, ctx->stream>>>(rp, c->NP, 0, x->d_p, x->d_hs);
      else if (x->S == 5) ntt32_headS_kernel<5><<<hg, 256, 0, ctx->stream>>>(rp, c->NP, 0, x->d_p, x->d_hs);
      else ntt32_headS_kernel<6><<<hg, 256, 0, ctx->stream>>>(rp, c->NP, 0, x->d_p, x->d_hs);
    }
    HIP_TRY(hipGetLastError());
  }
#define T32_FWD_S(PB) do { switch (x->S) { case 0: T32_FWD_GO(0, PB); break; case 1: T32_FWD_GO(1, PB); break; case 2: T32_FWD_GO(2, PB); break; case 3: T32_FWD_GO(3, PB); break; \
    case 4: T32_FWD_GO(4, PB); break; case 5: T32_FWD_GO(5, PB); break; default: T32_FWD_GO(6, PB); break; } } while (0)
  if (x->bits == 29) T32_FWD_S(29); else T32_FWD_S(30);
#undef T32_FWD_S
#undef T32_FWD_GO
  HIP_TRY(hipGetLastError());
  return 0;
}
template <int LQ, int R, int WT, int S>
static int t32_launch_crt(fhesi_ctx* ctx, const T32Config* c, const u32* d_t, i64 npolys, u64* d_parts, bool wm) {
  const i64 n = (i64)A32_N << S;
  const dim3 grid((unsigned)(n / 128), (unsigned)npolys);
  void* d_fl;
  FHESI_TRY(ws_reserve(ctx, 6, (size_t)grid.x * grid.y, &d_fl));          // (per lane, like every workspace slot)
  unsigned char* fl = (unsigned char*)d_fl;
  PROF_KERNEL(ctx, PROF_CRT, (crt32_scale_kernel<LQ, false, R, WT, S>));
  crt32_scale_kernel<LQ, false, R, WT, S><<<grid, 128, 0, ctx->stream>>>(d_t, n, c->NP, c->pr, c->d_cinv, c->d_inv57, c->d_Mw, d_parts, fl, wm ? 1 : 0);
  HIP_TRY(hipGetLastError());
  if (!ctx->opt.crt_skip_cleanup) {
    crt32_scale_kernel<LQ, true, R, WT, S><<<grid, 128, 0, ctx->stream>>>(d_t, n, c->NP, c->pr, c->d_cinv, c->d_inv57, c->d_Mw, d_parts, fl, wm ? 1 : 0);
    HIP_TRY(hipGetLastError());
  }
  return 0;
}
// d_t [npolys][NP][nrow] coefficient-form residues -> d_parts [npolys][ceil(logQ/64)][phi(m)]
static int t32_crt(fhesi_ctx* ctx, const T32Config* c, const u32* d_t, i64 npolys, u64* d_parts, bool wm) {
  const int S = ctx->tensor32->S, logQ = c->logQ;
  ProfScope prof(ctx, PROF_CRT, (double)npolys);
  if (!c->generic) {
    if (logQ == 512 && !S) return t32_launch_crt<512, T32_R_512, T32_WT_512, 0>(ctx, c, d_t, npolys, d_parts, wm);
    if (logQ == 512 && S) return t32_launch_crt<512, T32_R_512, T32_WT_512, 1>(ctx, c, d_t, npolys, d_parts, wm);
    if (logQ == 1024 && !S) return t32_launch_crt<1024, T32_R_1024, T32_WT_1024, 0>(ctx, c, d_t, npolys, d_parts, wm);
    return t32_launch_crt<1024, T32_R_1024, T32_WT_1024, 1>(ctx, c, d_t, npolys, d_parts, wm);
  }
  // the generic form: window from R J0 + 30 + log2(NP + 1) + 1 <= logQ - 64 up to bit 2 logQ
  const i64 n_out = ctx->phim, nrow = t32_nrow(ctx);
  const int WU = (2 * logQ + 27) / 28;
  int J0 = (logQ - 64 - 38) / 28;
  if (logQ < 64 + 38) J0 = 0;
  const int NW = WU - J0;
  if (NW > T32_GEN_NW || WU > T32_GEN_NWX || WU > c->WT) FHESI_FAIL("tensor32: logQ=%d outside the generic CRT window", logQ);
  const dim3 grid((unsigned)((n_out + 127) / 128), (unsigned)npolys);
  void* d_fl;
  FHESI_TRY(ws_reserve(ctx, 6, (size_t)grid.x * grid.y, &d_fl));
  unsigned char* fl = (unsigned char*)d_fl;
  const int fold = !ctx->lin_q ? 0 : (ctx->lin_prime ? 2 : 1);
  const i64 off = ctx->lin_q;
  // (S, FOLD) compile-time; the first pass over 16 or 24 words, whichever holds the window
#define T32_GEN_GO(NWM, SS, FF) do { \
    PROF_KERNEL(ctx, PROF_CRT, (crt32_scale_generic_kernel<NWM, false, SS, FF>)); \
    crt32_scale_generic_kernel<NWM, false, SS, FF><<<grid, 128, 0, ctx->stream>>>(d_t, nrow, n_out, off, c->NP, logQ, J0, NW, c->WT, c->pr, c->d_cinv, c->d_inv57, c->d_Mw, d_parts, fl, wm ? 1 : 0); \
    HIP_TRY(hipGetLastError()); \
    if (!ctx->opt.crt_skip_cleanup) { \
      crt32_scale_generic_kernel<T32_GEN_NWX, true, SS, FF><<<grid, 128, 0, ctx->stream>>>(d_t, nrow, n_out, off, c->NP, logQ, 0, WU, c->WT, c->pr, c->d_cinv, c->d_inv57, c->d_Mw, d_parts, fl, wm ? 1 : 0); \
      HIP_TRY(hipGetLastError()); \
    } } while (0)
  if (S >= 2) {      // rows of 2^16 and longer: the tail stages as a pass of their own over the sub-inverses (in place), then the fold from whole rows
    const fhesi_tensor32* x = ctx->tensor32;
    const i64 nr = npolys * c->NP, step = (65535 / c->NP) * (i64)c->NP;
    for (i64 r0 = 0; r0 < nr; r0 += step) {
      const unsigned ny = (unsigned)std::min(step, nr - r0);
      u32* rp = const_cast<u32*>(d_t) + (r0 << (A32_LOGN + S));
      if (S == 2) ntt32_tail2_kernel<<<dim3(16, ny), 256, 0, ctx->stream>>>(rp, c->NP, 0, x->d_p, x->d_ht);
      else if (S == 3) ntt32_tailS_kernel<3><<<dim3(A32_N / 256, ny), 256, 0, ctx->stream>>>(rp, c->NP, 0, x->d_p, x->d_hs, 0);
      else if (S == 4) ntt32_tailS_kernel<4><<<dim3(A32_N / 256, ny), 256, 0, ctx->stream>>>(rp, c->NP, 0, x->d_p, x->d_hs, 0);
      else if (S == 5) ntt32_tailS_kernel<5><<<dim3(A32_N / 256, ny), 256, 0, ctx->stream>>>(rp, c->NP, 0, x->d_p, x->d_hs, 0);
      else ntt32_tailS_kernel<6><<<dim3(A32_N / 256, ny), 256, 0, ctx->stream>>>(rp, c->NP, 0, x->d_p, x->d_hs, 0);
    }
    HIP_TRY(hipGetLastError());
  }
  // (A compiled form of this kernel for the reference drivers' own shapes -- logQ, word window and limb positions as constants, the words in
  // registers from the multiply-adds to the limbs -- was measured in round 6 and removed: crt class 8.80-8.85 ms per 1024 multiplications at
  // m = 32602 with this run-time form, 8.92-8.99 with the compiled one (profiles/r06_ab_crt_compiled.txt).  The kernel's time is its loads -- two
  // sub-rows at three folded positions per prime -- and its multiply-adds, twice the metric ring's for rows twice as long, not its indexing.)
#define T32_GEN_SF(NWM) do { \
    if (S != 1) { if (fold == 0) T32_GEN_GO(NWM, 0, 0); else if (fold == 1) T32_GEN_GO(NWM, 0, 1); else T32_GEN_GO(NWM, 0, 2); } \
    else { if (fold == 0) T32_GEN_GO(NWM, 1, 0); else if (fold == 1) T32_GEN_GO(NWM, 1, 1); else T32_GEN_GO(NWM, 1, 2); } } while (0)
  if (NW <= 16) T32_GEN_SF(16); else T32_GEN_SF(T32_GEN_NW);
#undef T32_GEN_SF
#undef T32_GEN_GO
  return 0;
}

// ---- pairs (fhesi_ct_mul_relin_batch_dev): a, b [count][2][phi(m)][nlimbs] -> d_parts [count * 3][ceil(logQ/64)][phi(m)]: the scaled-down tProd as ByteDecomp takes it
int launch_tensor32(fhesi_ctx* ctx, u64 p, const u64* d_a, const u64* d_b, int nlimbs, int logQ, i64 count, u64* d_parts, bool parts_wm) {
  T32Config* c;
  FHESI_TRY(t32_config(ctx, p, nlimbs, logQ, 1, &c));
  if (!count) return 0;
  const fhesi_tensor32* x = ctx->tensor32;
  const int NP = c->NP, S = x->S;
  const i64 nrow = t32_nrow(ctx);
  void *d_r, *d_t;
  FHESI_TRY(ws_reserve(ctx, 5, (size_t)count * 4 * NP * nrow * 4, &d_r));
  FHESI_TRY(ws_reserve(ctx, 1, (size_t)count * 3 * NP * nrow * 4, &d_t));
  FHESI_TRY(t32_rns(ctx, c, d_a, d_b, count * 4, 0, true, (u32*)d_r));
  FHESI_TRY(t32_fwd(ctx, c, (u32*)d_r, count * 4));
  {
    ProfScope prof(ctx, PROF_NTT_INV, (double)(count * 3 * NP));
    const dim3 grid((unsigned)(count >= 8 ? ((count + 7) / 8) * 24 : count * 3), (unsigned)(NP << S));      // (groups of 8 ciphertexts x 3 rows, or the rows themselves: see the kernel)
    if (x->bits == 29) {
      PROF_KERNEL(ctx, PROF_NTT_INV, (ntt32_inv_kernel3<false, true, T32Primes, 29>));
      ntt32_inv_kernel3<false, true, T32Primes, 29><<<grid, A32_T, 0, ctx->stream>>>((u32*)d_t, count * 3, NP, 0, c->pr, x->d_inv, S, (const u32*)d_r);
    } else {
      PROF_KERNEL(ctx, PROF_NTT_INV, (ntt32_inv_kernel3<false, true, T32Primes>));
      ntt32_inv_kernel3<false, true, T32Primes><<<grid, A32_T, 0, ctx->stream>>>((u32*)d_t, count * 3, NP, 0, c->pr, x->d_inv, S, (const u32*)d_r);
    }
    HIP_TRY(hipGetLastError());
  }
  return t32_crt(ctx, c, (const u32*)d_t, count * 3, d_parts, parts_wm);
}

// ---- sums of products per group (fhesi_ct_mul_sum_relin_dev)
int tensor32_sum_begin(fhesi_ctx* ctx, u64 p, int nlimbs, int logQ, i64 gmax) {
  T32Config* c;
  FHESI_TRY(t32_config(ctx, p, nlimbs, logQ, gmax, &c));
  ctx->tensor32->cur = c;
  return 0;
}
size_t tensor32_sum_bytes(const fhesi_ctx* ctx, i64 ngroups) { return (size_t)ngroups * 3 * ctx->tensor32->cur->NP * t32_nrow(ctx) * 4; }
// d_ops: [nua + nub][2][phi(m)][nlimbs] (the a operands first); slots / segments on the device; d_sum [ng][3][NP][nrow]
int tensor32_sum_pass(fhesi_ctx* ctx, const u64* d_ops, i64 nua, i64 nub, const int* d_slot_a, const int* d_slot_b, const int* d_seg, i64 ng, i64 nterms, bool accumulate, void* d_sum) {
  const T32Config* c = ctx->tensor32->cur;
  const i64 nrow = t32_nrow(ctx), n = ctx->phim;
  void* d_r;
  FHESI_TRY(ws_reserve(ctx, 0, (size_t)(nua + nub) * 2 * c->NP * nrow * 4, &d_r));
  const u64* d_b = d_ops + (size_t)nua * 2 * n * c->nl;
  FHESI_TRY(t32_rns(ctx, c, d_ops, d_b, (nua + nub) * 2, nua * 2, false, (u32*)d_r));
  FHESI_TRY(t32_fwd(ctx, c, (u32*)d_r, (nua + nub) * 2));
  const u32* ra = (const u32*)d_r;
  const u32* rb = ra + (size_t)nua * 2 * c->NP * nrow;
  ProfScope prof(ctx, PROF_TENSOR, (double)nterms);
  PROF_KERNEL(ctx, PROF_TENSOR, tensor_sum32_kernel);
  for (i64 done = 0; done < ng; done += 65535) {
    const i64 cnt = ng - done < 65535 ? ng - done : 65535;
    dim3 grid((unsigned)(nrow / 256), (unsigned)c->NP, (unsigned)cnt);
    tensor_sum32_kernel<<<grid, 256, 0, ctx->stream>>>(ra, rb, d_slot_a, d_slot_b, d_seg + done, accumulate ? 1 : 0, (u32*)d_sum + (size_t)done * 3 * c->NP * nrow, nrow, c->NP, c->pr);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}
int tensor32_sum_finish(fhesi_ctx* ctx, void* d_sum, i64 ng, u64* d_parts, bool parts_wm) {
  const T32Config* c = ctx->tensor32->cur;
  const fhesi_tensor32* x = ctx->tensor32;
  {
    ProfScope prof(ctx, PROF_NTT_INV, (double)(ng * 3 * c->NP));
    const dim3 grid((unsigned)(ng * 3), (unsigned)(c->NP << x->S));
    if (x->bits == 29) {
      PROF_KERNEL(ctx, PROF_NTT_INV, (ntt32_inv_kernel3<false, false, T32Primes, 29>));
      ntt32_inv_kernel3<false, false, T32Primes, 29><<<grid, A32_T, 0, ctx->stream>>>((u32*)d_sum, ng * 3, c->NP, 0, c->pr, x->d_inv, x->S, nullptr);
    } else {
      PROF_KERNEL(ctx, PROF_NTT_INV, (ntt32_inv_kernel3<false, false, T32Primes>));
      ntt32_inv_kernel3<false, false, T32Primes><<<grid, A32_T, 0, ctx->stream>>>((u32*)d_sum, ng * 3, c->NP, 0, c->pr, x->d_inv, x->S, nullptr);
    }
    HIP_TRY(hipGetLastError());
  }
  return t32_crt(ctx, c, (const u32*)d_sum, ng * 3, d_parts, parts_wm);
}
