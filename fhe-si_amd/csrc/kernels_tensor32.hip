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
// fhesi_ct_mul_dev keeps the reference chain (its rows ARE visible).  Option tensor32 = 0 keeps the chain in the fused pipeline too.
#include "fhesi_internal.h"
#include "ntt32_core.inc"
#include <cmath>

struct fhesi_tensor32 {
  int NP = 0, nl = 0, logQ = 0;
  u64 lift = 0;
  T32Primes pr;
  Tw32* d_fwd = nullptr;             // [NP][2^14]
  Tw32* d_inv = nullptr;
  u32* d_rns = nullptr;              // [2][NP][2 nl + 4]: (s 2^(32k) mod p) k < 2 nl, -(s 2^(64 nl)) mod p, 2^32 mod p, floor(2^61 / p), p;  s = lift (class 0) or 1
  Tw32* d_cinv = nullptr;            // [NP] (M / p_i)^-1 mod p_i
  u32* d_inv58 = nullptr;            // [NP] floor(2^58 / p_i)
  u32* d_M28 = nullptr;              // [NP + 1][WT]: M_i in words of 28 bits; row NP = 2^(28 WT) - M
  int WT = 0;
};

static void t32_release(fhesi_tensor32* x) {
  if (!x) return;
  hipFree(x->d_fwd); hipFree(x->d_inv); hipFree(x->d_rns); hipFree(x->d_cinv); hipFree(x->d_inv58); hipFree(x->d_M28);
  delete x;
}
void tensor32_free(fhesi_ctx* ctx) { t32_release(ctx->tensor32); ctx->tensor32 = nullptr; }

// ---------------------------------------------------------------------------------------------- host big integers (little-endian u64 limbs)
typedef std::vector<u64> Big;
static Big big_mul_small(const Big& a, u64 b) { return hm::bn_mul_small(a, b); }
static u64 big_mod_small(const Big& a, u64 q) { u128 r = 0; for (size_t i = a.size(); i-- > 0;) r = ((r << 64) | a[i]) % q; return (u64)r; }
static u32 big_bits28(const Big& a, int l) {          // word l of the radix-2^28 form
  const int bit = 28 * l, w = bit >> 6, o = bit & 63;
  u64 v = (size_t)w < a.size() ? a[w] >> o : 0;
  if (o > 36 && (size_t)w + 1 < a.size()) v |= a[w + 1] << (64 - o);
  return (u32)(v & 0xfffffffu);
}

// The number of primes the tensor half needs, 0 if this shape does not run through it.
//   |x| < 2^TB with TB = 2 (64 nl - 1) + bits(p) + log2(n) + 1;  M > 2^(TB + 3) keeps x/M below 1/8 (kappa is then decided by a coarse
//   fixed-point sum), and the chain product must exceed 2^(TB + 1) so that the reference's own centred integers are these same integers.
static int t32_plan(const fhesi_ctx* ctx, u64 p, int nlimbs, int logQ, std::vector<u32>* primes) {
  if (!ctx->pow2 || ctx->logn != A32_LOGN || !ctx->opt.tensor32 || !ctx->opt.ntt32_v3) return 0;
  if (logQ != 512 || nlimbs < 1 || nlimbs > 12 || 64 * nlimbs < logQ || p < 2) return 0;       // (crt32_scale_kernel is instantiated for logQ = 512)
  int pbits = 0;
  while (pbits < 64 && (p >> pbits)) ++pbits;
  const double TB = 2.0 * (64 * nlimbs - 1) + pbits + A32_LOGN + 1;
  double chain = 0;
  for (int i = 0; i < ctx->L; ++i) chain += std::log2((double)ctx->q[i]);
  if (chain < TB + 1.5) return 0;
  double have = 0;
  int np = 0;
  for (u64 k = ((u64)1 << (29 - A32_LOGN)) - 1; k > ((u64)1 << (28 - A32_LOGN)) && have < TB + 3.5; --k) {
    const u64 cand = (k << (A32_LOGN + 1)) + 1;
    if (cand > ((u64)1 << 30) - ((u64)1 << 15) + 1 || !hm::is_prime(cand)) continue;
    if (np == T32_MAXP) return 0;
    if (primes) primes->push_back((u32)cand);
    have += std::log2((double)cand);
    ++np;
  }
  if (have < TB + 3.5) return 0;
  // the window of crt32_scale_kernel: words up to bit 2 logQ, all of M inside 28 * 38 bits
  if (have > 28.0 * 38 - 8) return 0;
  return np;
}
bool tensor32_applies(const fhesi_ctx* ctx, u64 p, int nlimbs, int logQ) { return t32_plan(ctx, p, nlimbs, logQ, nullptr) > 0; }

static constexpr int T32_WT = 38;          // words of 28 bits per table row (1064 bits)

static int t32_init(fhesi_ctx* ctx, u64 lift, int nlimbs, int logQ) {
  fhesi_tensor32* x = ctx->tensor32;
  if (x && x->lift == lift && x->nl == nlimbs && x->logQ == logQ) return 0;
  std::vector<u32> primes;
  const int NP = t32_plan(ctx, lift, nlimbs, logQ, &primes);
  if (!NP) FHESI_FAIL("tensor32: shape not supported");
  if (x) { HIP_TRY(hipStreamSynchronize(ctx->stream)); if (ctx->lane_stream) HIP_TRY(hipStreamSynchronize(ctx->lane_stream)); tensor32_free(ctx); }
  x = new fhesi_tensor32();
  x->NP = NP; x->nl = nlimbs; x->logQ = logQ; x->lift = lift; x->WT = T32_WT;
  const i64 n = A32_N;
  std::vector<Tw32> hf((size_t)NP * n), hi((size_t)NP * n);
  const int stride = 2 * nlimbs + 4;
  std::vector<u32> rns((size_t)2 * NP * stride);
  for (int a = 0; a < NP; ++a) {
    const u64 p = primes[a];
    u64 psi = 0;
    for (u64 gq = 2; gq < 1000 && !psi; ++gq) {
      const u64 cand = hm::powmod(gq, (p - 1) / (2 * (u64)n), p);
      if (hm::powmod(cand, (u64)n, p) == p - 1) psi = cand;
    }
    if (!psi) { t32_release(x); FHESI_FAIL("tensor32: no 2n-th root"); }
    const u64 ipsi = hm::invmod(psi, p);
    auto tw = [&](u64 w) { return Tw32{(u32)w, (u32)((w << 32) / p)}; };
    for (u64 idx = 0; idx < (u64)n; ++idx) {
      const u64 e = hm::brv(idx, A32_LOGN);
      hf[(size_t)a * n + idx] = tw(hm::powmod(psi, e, p));       // (w itself: the plain-row kernels take a32_ct<false>)
      hi[(size_t)a * n + idx] = tw(hm::powmod(ipsi, e, p));
    }
    const u64 ninv = hm::invmod((u64)n % p, p);
    x->pr.p[a] = (u32)p;
    x->pr.ninv[a] = (u32)ninv;
    x->pr.ninv_p[a] = (u32)((ninv << 32) / p);
    x->pr.mu61[a] = (u32)(((u64)1 << 61) / p);
    for (int cls = 0; cls < 2; ++cls) {
      u32* e = &rns[((size_t)cls * NP + a) * stride];
      const u64 b32 = ((u64)1 << 32) % p;
      u64 cur = cls == 0 ? lift % p : 1;
      for (int k = 0; k < 2 * nlimbs; ++k) { e[k] = (u32)cur; cur = hm::mulmod(cur, b32, p); }
      e[2 * nlimbs] = (u32)((p - cur) % p);          // two's complement: value = unsigned - 2^(64 nl)
      e[2 * nlimbs + 1] = (u32)b32;
      e[2 * nlimbs + 2] = x->pr.mu61[a];
      e[2 * nlimbs + 3] = (u32)p;
    }
  }
  // CRT tables
  Big M{1};
  for (int a = 0; a < NP; ++a) M = big_mul_small(M, primes[a]);
  std::vector<Tw32> cinv(NP);
  std::vector<u32> inv58(NP), M28((size_t)(NP + 1) * T32_WT);
  for (int a = 0; a < NP; ++a) {
    Big Mi{1};
    for (int b = 0; b < NP; ++b) if (b != a) Mi = big_mul_small(Mi, primes[b]);
    const u64 p = primes[a];
    const u64 c = hm::invmod(big_mod_small(Mi, p), p);
    cinv[a] = Tw32{(u32)c, (u32)((c << 32) / p)};
    inv58[a] = (u32)(((u64)1 << 58) / p);
    for (int l = 0; l < T32_WT; ++l) M28[(size_t)a * T32_WT + l] = big_bits28(Mi, l);
  }
  {
    // 2^(28 WT) - M: two's complement of M over enough limbs, read through the same word extraction (masked to 28 WT bits)
    Big N(M);
    N.resize((28 * T32_WT + 63) / 64 + 1, 0);
    u64 carry = 1;
    for (auto& w : N) { const u64 v = ~w + carry; carry = (carry && v == 0) ? 1 : 0; w = v; }
    for (int l = 0; l < T32_WT; ++l) M28[(size_t)NP * T32_WT + l] = big_bits28(N, l);
  }
  auto up = [&](void** d, const void* h, size_t bytes) -> bool {
    return hipMalloc(d, bytes) == hipSuccess && hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice) == hipSuccess;
  };
  if (!up((void**)&x->d_fwd, hf.data(), hf.size() * sizeof(Tw32)) || !up((void**)&x->d_inv, hi.data(), hi.size() * sizeof(Tw32)) ||
      !up((void**)&x->d_rns, rns.data(), rns.size() * 4) || !up((void**)&x->d_cinv, cinv.data(), cinv.size() * sizeof(Tw32)) ||
      !up((void**)&x->d_inv58, inv58.data(), inv58.size() * 4) || !up((void**)&x->d_M28, M28.data(), M28.size() * 4)) {
    t32_release(x);
    FHESI_FAIL("tensor32: table upload failed");
  }
  ctx->tensor32 = x;
  return 0;
}

// ---------------------------------------------------------------------------------------------- big integer -> residues
// a, b: [count][2][n][NL] two's complement coefficients; rows [count][4][NP][n] (a0, a1, b0, b1), values below 3p
template <int NL>
__global__ void __launch_bounds__(256) rns32_reduce_kernel(const u64* __restrict__ a, const u64* __restrict__ b, i64 n, u32* __restrict__ rows, int NP,
                                                            const u32* __restrict__ tab) {
  __shared__ __attribute__((aligned(16))) u64 sl[NL * 256];       // [NL][256]
  const i64 poly = blockIdx.y;                                   // ct * 4 + j
  const i64 ct = poly >> 2;
  const int jp = (int)(poly & 3), cls = jp >> 1;
  const i64 j0 = (i64)blockIdx.x * 256;
  const int tid = threadIdx.x;
  const u64* __restrict__ src = (cls ? b : a) + (ct * 2 + (jp & 1)) * n * NL + j0 * NL;
#pragma unroll
  for (int it = 0; it < NL; ++it) {
    const int e = it * 256 + tid;
    sl[(e % NL) * 256 + e / NL] = src[e];
  }
  __syncthreads();
  u32 x[2 * NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) { const u64 v = sl[k * 256 + tid]; x[2 * k] = (u32)v; x[2 * k + 1] = (u32)(v >> 32); }
  const u32 neg = x[2 * NL - 1] >> 31;
  constexpr int STRIDE = 2 * NL + 4;
  u32* __restrict__ o = rows + poly * NP * n + j0 + tid;
#pragma unroll 2
  for (int i = 0; i < NP; ++i) {
    const u32* __restrict__ t = tab + ((i64)cls * NP + i) * STRIDE;
    const u32 r32 = t[2 * NL + 1], mu = t[2 * NL + 2], p = t[2 * NL + 3];
    // products below 2^62 - 2^47: four fit a 64-bit accumulator, three on top of a folded value (below 2^62 + 2^32)
    u64 acc = 0;
    int room = 4;
#pragma unroll
    for (int k = 0; k < 2 * NL; ++k) {
      if (room == 0) { acc = (u64)(u32)(acc >> 32) * r32 + (u32)acc; room = 3; }
      acc += (u64)x[k] * t[k];
      --room;
    }
    acc += neg ? t[2 * NL] : 0u;
    acc = (u64)(u32)(acc >> 32) * r32 + (u32)acc;       // below 2^62 + 2^32
    acc = (u64)(u32)(acc >> 32) * r32 + (u32)acc;       // below 2^61
    const u32 q = __umulhi((u32)(acc >> 29), mu);        // at most 2 below floor(acc / p)
    o[(i64)i * n] = (u32)acc - q * p;
  }
}

// ---------------------------------------------------------------------------------------------- residues -> round(x / 2^logQ) mod 2^logQ
// rows [npolys][NP][n] (below p) -> out [npolys][LQ/64][n] limb-major positive residues (crt mode 1, kernels_crt.hip)
template <int LQ, bool EXACT>
__global__ void __launch_bounds__(128) crt32_scale_kernel(const u32* __restrict__ rows, i64 n, int NP, T32Primes pr, const Tw32* __restrict__ cinv,
                                                           const u32* __restrict__ inv58, const u32* __restrict__ M28, u64* __restrict__ out,
                                                           unsigned char* __restrict__ flags) {
  static_assert((LQ & 63) == 0, "logQ a multiple of 64");
  constexpr int WT = T32_WT;                          // table words
  constexpr int WU = (2 * LQ + 27) / 28;              // words that reach below bit 2 logQ
  constexpr int J0 = EXACT ? 0 : (LQ - 101) / 28;     // first word formed: 28 J0 + 37 <= logQ - 64
  constexpr int NW = WU - J0;
  static_assert(WU <= WT && J0 >= 0, "window");
  const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  if (EXACT && !flags[wg]) return;
  const i64 poly = blockIdx.y;
  const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  const u32* __restrict__ src = rows + poly * NP * n + j;
  u64 acc[NW];
#pragma unroll
  for (int l = 0; l < NW; ++l) acc[l] = 0;
  u32 fsum = 0;
#pragma unroll 2
  for (int i = 0; i < NP; ++i) {
    const u32 p = pr.p[i];
    u32 y = mul_lazy32(src[(i64)i * n], cinv[i], p);
    y = y >= p ? y - p : y;
    fsum += __umulhi(y, inv58[i]);                    // (y / p) 2^26, low by less than one unit
    const u32* __restrict__ Mi = M28 + (i64)i * WT + J0;
#pragma unroll
    for (int l = 0; l < NW; ++l) acc[l] += (u64)y * Mi[l];
  }
  // x/M is within 1/8 of an integer: kappa = round(sum y_i / p_i); x = sum - kappa M = sum + kappa (2^(28 WT) - M)  (mod 2^(28 WT))
  const u32 kappa = (fsum + (1u << 25)) >> 26;
  {
    const u32* __restrict__ Nm = M28 + (i64)NP * WT + J0;
#pragma unroll
    for (int l = 0; l < NW; ++l) acc[l] += (u64)kappa * Nm[l];
  }
  // carries: words of 28 bits
  u64 carry = 0;
#pragma unroll
  for (int l = 0; l < NW; ++l) { const u64 v = acc[l] + carry; acc[l] = v & 0xfffffffull; carry = v >> 28; }
  // 64 bits from bit B of x (two's complement, bits above 28 WU dropped)
  auto limb = [&](int B) -> u64 {
    const int l0 = B / 28 - J0, o = B % 28;
    u64 v = acc[l0] >> o;
    if (l0 + 1 < NW) v |= acc[l0 + 1] << (28 - o);
    if (l0 + 2 < NW) v |= acc[l0 + 2] << (56 - o);
    if (l0 + 3 < NW && 84 - o < 64) v |= acc[l0 + 3] << (84 - o);
    return v;
  };
  const u64 G = limb(LQ - 64);                         // bits logQ-64 .. logQ-1
  int undecided = 0;
  if (!EXACT) undecided = (G == 0x7fffffffffffffffull) ? 1 : 0;
  u64 c = G >> 63;                                     // round half up: + bit logQ-1
  u64* __restrict__ o = out + poly * (LQ / 64) * n + j;
#pragma unroll
  for (int i = 0; i < LQ / 64; ++i) {
    u64 v = limb(LQ + 64 * i);
    v += c;
    c = (c && v == 0) ? 1 : 0;
    o[(i64)i * n] = v;
  }
  if (!EXACT) {
    const int any = __syncthreads_or(undecided);
    if (threadIdx.x == 0) flags[wg] = any ? 1 : 0;
  }
}

// ---------------------------------------------------------------------------------------------- the tensor half
// a, b: [count][2][n][nlimbs] coefficients -> d_parts [count * 3][logQ/64][n]: the scaled-down tProd as ByteDecomp takes it
int launch_tensor32(fhesi_ctx* ctx, u64 p, const u64* d_a, const u64* d_b, int nlimbs, int logQ, i64 count, u64* d_parts) {
  FHESI_TRY(t32_init(ctx, p, nlimbs, logQ));
  if (!count) return 0;
  fhesi_tensor32* x = ctx->tensor32;
  const int NP = x->NP;
  const i64 n = A32_N;
  void *d_r, *d_t;
  FHESI_TRY(ws_reserve(ctx, 5, (size_t)count * 4 * NP * n * 4, &d_r));
  FHESI_TRY(ws_reserve(ctx, 1, (size_t)count * 3 * NP * n * 4, &d_t));
  {
    ProfScope prof(ctx, PROF_RNS, (double)(count * 4));
    const dim3 grid((unsigned)(n / 256), (unsigned)(count * 4));
#define T32_RNS(NL) case NL: PROF_KERNEL(ctx, PROF_RNS, rns32_reduce_kernel<NL>); rns32_reduce_kernel<NL><<<grid, 256, 0, ctx->stream>>>(d_a, d_b, n, (u32*)d_r, NP, x->d_rns); break;
    switch (nlimbs) {
      T32_RNS(1) T32_RNS(2) T32_RNS(3) T32_RNS(4) T32_RNS(5) T32_RNS(6) T32_RNS(7) T32_RNS(8) T32_RNS(9) T32_RNS(10) T32_RNS(11) T32_RNS(12)
      default: FHESI_FAIL("tensor32: %d limbs", nlimbs);
    }
#undef T32_RNS
    HIP_TRY(hipGetLastError());
  }
  {
    ProfScope prof(ctx, PROF_NTT_FWD, (double)(count * 4 * NP));
    PROF_KERNEL(ctx, PROF_NTT_FWD, (ntt32_fwd_kernel3<false, 0, false, T32Primes, false>));
    ntt32_fwd_kernel3<false, 0, false, T32Primes, false><<<(unsigned)(count * 4 * NP), A32_T, 0, ctx->stream>>>((u32*)d_r, count * 4, NP, 0, x->pr, x->d_fwd, Dig32Src{}, Aux32Head{});
    HIP_TRY(hipGetLastError());
  }
  {
    ProfScope prof(ctx, PROF_NTT_INV, (double)(count * 3 * NP));
    PROF_KERNEL(ctx, PROF_NTT_INV, (ntt32_inv_kernel3<false, true, T32Primes>));
    const unsigned grid = (unsigned)(((count + 7) / 8) * 24 * NP);
    ntt32_inv_kernel3<false, true, T32Primes><<<grid, A32_T, 0, ctx->stream>>>((u32*)d_t, count * 3, NP, 0, x->pr, x->d_inv, 0, (const u32*)d_r);
    HIP_TRY(hipGetLastError());
  }
  {
    ProfScope prof(ctx, PROF_CRT, (double)(count * 3));
    const dim3 grid((unsigned)(n / 128), (unsigned)(count * 3));
    void* d_fl;
    FHESI_TRY(ws_reserve(ctx, 6, (size_t)grid.x * grid.y, &d_fl));          // (per lane, like every workspace slot)
    unsigned char* fl = (unsigned char*)d_fl;
    PROF_KERNEL(ctx, PROF_CRT, (crt32_scale_kernel<512, false>));
    crt32_scale_kernel<512, false><<<grid, 128, 0, ctx->stream>>>((const u32*)d_t, n, NP, x->pr, x->d_cinv, x->d_inv58, x->d_M28, d_parts, fl);
    HIP_TRY(hipGetLastError());
    if (!ctx->opt.crt_skip_cleanup) {
      crt32_scale_kernel<512, true><<<grid, 128, 0, ctx->stream>>>((const u32*)d_t, n, NP, x->pr, x->d_cinv, x->d_inv58, x->d_M28, d_parts, fl);
      HIP_TRY(hipGetLastError());
    }
  }
  return 0;
}
