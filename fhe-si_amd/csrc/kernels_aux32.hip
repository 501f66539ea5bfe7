// kernels_aux32.hip -- the key-switch dot product through FOUR 30-bit auxiliary primes (n = 2^14).
//
// kernels_ksaux.hip computes the key switch's integer dot product S_l = sum_k digit_k (*) K_{k,l} modulo two 60-bit chain primes.
// The same integers (|S_l| < 2^119) are determined just as well by their residues modulo four primes below 2^30, and there a
// multiply-accumulate is ONE v_mad_u64_u32 (30 x 30 -> 60 bits, sixteen of them fit a 64-bit accumulator) instead of the four that
// a 60 x 60-bit product needs -- half the multiplies for the same information, and half the key bytes per multiply through L2.
// The price is an own set of transforms over those primes: 32-bit negacyclic NTTs (Harvey butterflies on lazy values below 4p),
// written for this one size.  Evaluation order is whatever the forward transform produces (no bit reversal anywhere): the dot
// product is element-wise and the inverse transform is the exact mirror.
//
//   p_a = the four largest primes below 2^30 with p = 1 mod 2^15  (aux32_init; roots and tables per context, on first use)
//   ntt32_fwd_kernel<DIGITS>   32 values per thread, 512 threads per row: 5 stages in registers, LDS exchange, 5 stages, exchange, 4 stages
//   ntt32_inv_kernel           the mirror (Gentleman-Sande), 1/n folded into a final multiplication
//   dot32_kernel               O[ct][r][l][a] = sum_k D[ct][k][a] * K[a][l][r][k]  mod p_a
// The recombination (Garner over the four residues, then exactly as ks_recombine_kernel) is in kernels_crt.hip.
#include "fhesi_internal.h"

struct Tw32 { u32 w, wp; };          // constant and floor(w 2^32 / p)
struct Aux32Primes { u32 p[4]; u32 ninv[4], ninv_p[4]; };

struct fhesi_aux32 {
  Aux32Primes pr;
  Tw32* d_fwd = nullptr;               // [4][n]  psi^brv(idx), idx = m + i (stage with m groups, group i)
  Tw32* d_inv = nullptr;               // [4][n]  the inverses
};

static constexpr int A32_LOGN = 14, A32_N = 1 << A32_LOGN, A32_T = 512, A32_P = 592;      // LDS stride of a 512-element sub-problem (padded)
__device__ __forceinline__ u32 a32_f(u32 t_id) { return t_id + ((t_id >> 5) << 2); }       // position inside a sub-problem (4 pad words per 32)
__device__ __forceinline__ u32 mul_lazy32(u32 y, Tw32 t, u32 p) { return y * t.w - __umulhi(y, t.wp) * p; }     // y any 32-bit value -> [0, 2p)
__device__ __forceinline__ int a32_bfly_k(int b, int h) { return ((b / h) * 2 * h) + (b % h); }
// Cooley-Tukey butterfly on lazy values (below 4p < 2^32):  X' = X + w Y,  Y' = X - w Y
__device__ __forceinline__ void a32_ct(u32& x, u32& y, Tw32 t, u32 p) {
  const u32 twop = 2 * p;
  const u32 X = x >= twop ? x - twop : x;
  const u32 T = mul_lazy32(y, t, p);
  x = X + T;
  y = X - T + twop;
}
// Gentleman-Sande butterfly on values below 2p:  X' = X + Y,  Y' = (X - Y) w
__device__ __forceinline__ void a32_gs(u32& x, u32& y, Tw32 t, u32 p) {
  const u32 twop = 2 * p;
  const u32 s = x + y, d = x - y + twop;
  x = s >= twop ? s - twop : s;
  y = mul_lazy32(d, t, p);
}

struct Dig32Src { const u64* parts; int nl, digit_bits, nd; };     // limb-major scaled-down parts [npolys][nl][n]

// rows: [count][nslots][n] u32; block rb -> (unit c = rb % count, slot = rb / count), prime a0 + slot.
// DIGITS: unit c = poly * nd + digit, the values are cut out of the parts (ByteDecomp, Ciphertext.cpp:82-105); output lazy (below 4p).
// otherwise: in place on the row, output reduced (below p).
template <bool DIGITS>
__global__ void __launch_bounds__(A32_T, 2) ntt32_fwd_kernel(u32* __restrict__ rows, i64 count, int nslots, int a0, Aux32Primes pr, const Tw32* __restrict__ tabs,
                                                             Dig32Src ds) {
  __shared__ u32 lds[32 * A32_P];
  const u32 tid = threadIdx.x;
  const i64 c = blockIdx.x % count;
  const int slot = (int)(blockIdx.x / count), a = a0 + slot;
  const u32 p = pr.p[a];
  const Tw32* __restrict__ tab = tabs + (i64)a * A32_N;
  u32* __restrict__ g = rows + (c * nslots + slot) * A32_N;
  u32 r[32];
  if (DIGITS) {
    const u32 d = (u32)(c % ds.nd);
    const i64 poly = c / ds.nd;
    const u32 bit = d * (u32)ds.digit_bits, g0 = bit >> 5, sh = bit & 31;
    const u32 mask = (1u << ds.digit_bits) - 1;
    const u32* __restrict__ p32 = reinterpret_cast<const u32*>(ds.parts);
    const u32* __restrict__ w0 = p32 + (((poly * ds.nl + (g0 >> 1)) << A32_LOGN) << 1) + (g0 & 1);
    const bool two = (sh + ds.digit_bits > 32) && (int)((g0 + 1) >> 1) < ds.nl;
    if (two) {
      const u32 g1 = g0 + 1;
      const u32* __restrict__ w1 = p32 + (((poly * ds.nl + (g1 >> 1)) << A32_LOGN) << 1) + (g1 & 1);
#pragma unroll
      for (int k = 0; k < 32; ++k) { const u32 e = 2 * (k * A32_T + tid); r[k] = ((w0[e] >> sh) | (w1[e] << (32 - sh))) & mask; }
    } else {
#pragma unroll
      for (int k = 0; k < 32; ++k) r[k] = (w0[2 * (k * A32_T + tid)] >> sh) & mask;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 32; ++k) r[k] = g[k * A32_T + tid];
  }
  // phase A: element e = k * 512 + tid; distances 16, 8, 4, 2, 1 in k; twiddles depend on the register index only
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int h = 16 >> s;
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      const int k = a32_bfly_k(b, h);
      a32_ct(r[k], r[k + h], tab[(1 << s) + (k >> (5 - s))], p);
    }
  }
#pragma unroll
  for (int k = 0; k < 32; ++k) lds[k * A32_P + a32_f(tid)] = r[k];
  __syncthreads();
  const u32 kq = tid >> 4, lo = tid & 15;          // sub-problem and position inside a group of 16
#pragma unroll
  for (int k2 = 0; k2 < 32; ++k2) r[k2] = lds[kq * A32_P + a32_f(k2 * 16 + lo)];
  // phase B: sub-problem kq (512 elements t = k2 * 16 + lo); distances 16 .. 1 in k2
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const int h = 16 >> u;
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      const int k2 = a32_bfly_k(b, h);
      a32_ct(r[k2], r[k2 + h], tab[(32 << u) + (kq << u) + (k2 >> (5 - u))], p);
    }
  }
  __syncthreads();
#pragma unroll
  for (int k2 = 0; k2 < 32; ++k2) lds[kq * A32_P + a32_f(k2 * 16 + lo)] = r[k2];
  __syncthreads();
  // phase C: this thread takes the groups k2 = 2 lo, 2 lo + 1 (32 consecutive elements); distances 8 .. 1 inside a group
#pragma unroll
  for (int i = 0; i < 32; ++i) r[i] = lds[kq * A32_P + lo * 36 + i];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int h = 8 >> v;
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int x = a32_bfly_k(b, h);
        a32_ct(r[gq * 16 + x], r[gq * 16 + x + h], tab[(1024 << v) + ((kq * 32 + 2 * lo + gq) << v) + (x >> (4 - v))], p);
      }
    }
  }
  u32* __restrict__ o = g + kq * 512 + lo * 32;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    u32 v = r[i];
    if (!DIGITS) { const u32 twop = 2 * p; v = v >= twop ? v - twop : v; v = v >= p ? v - p : v; }
    o[i] = v;
  }
}

// the mirror: input in the forward transform's output order (values below 2p), output natural order, scaled by 1/n, reduced
__global__ void __launch_bounds__(A32_T, 2) ntt32_inv_kernel(u32* __restrict__ rows, i64 count, int nslots, int a0, Aux32Primes pr, const Tw32* __restrict__ tabs) {
  __shared__ u32 lds[32 * A32_P];
  const u32 tid = threadIdx.x;
  const i64 c = blockIdx.x % count;
  const int slot = (int)(blockIdx.x / count), a = a0 + slot;
  const u32 p = pr.p[a];
  const Tw32* __restrict__ tab = tabs + (i64)a * A32_N;
  u32* __restrict__ g = rows + (c * nslots + slot) * A32_N;
  const u32 kq = tid >> 4, lo = tid & 15;
  u32 r[32];
  const u32* __restrict__ in = g + kq * 512 + lo * 32;
#pragma unroll
  for (int i = 0; i < 32; ++i) r[i] = in[i];
#pragma unroll
  for (int v = 3; v >= 0; --v) {
    const int h = 8 >> v;
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int x = a32_bfly_k(b, h);
        a32_gs(r[gq * 16 + x], r[gq * 16 + x + h], tab[(1024 << v) + ((kq * 32 + 2 * lo + gq) << v) + (x >> (4 - v))], p);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 32; ++i) lds[kq * A32_P + lo * 36 + i] = r[i];
  __syncthreads();
#pragma unroll
  for (int k2 = 0; k2 < 32; ++k2) r[k2] = lds[kq * A32_P + a32_f(k2 * 16 + lo)];
#pragma unroll
  for (int u = 4; u >= 0; --u) {
    const int h = 16 >> u;
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      const int k2 = a32_bfly_k(b, h);
      a32_gs(r[k2], r[k2 + h], tab[(32 << u) + (kq << u) + (k2 >> (5 - u))], p);
    }
  }
  __syncthreads();
#pragma unroll
  for (int k2 = 0; k2 < 32; ++k2) lds[kq * A32_P + a32_f(k2 * 16 + lo)] = r[k2];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 32; ++k) r[k] = lds[k * A32_P + a32_f(tid)];
#pragma unroll
  for (int s = 4; s >= 0; --s) {
    const int h = 16 >> s;
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      const int k = a32_bfly_k(b, h);
      a32_gs(r[k], r[k + h], tab[(1 << s) + (k >> (5 - s))], p);
    }
  }
  const Tw32 tn{pr.ninv[a], pr.ninv_p[a]};
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    u32 v = mul_lazy32(r[k], tn, p);
    v = v >= p ? v - p : v;
    g[k * A32_T + tid] = v;
  }
}

// ---------------------------------------------------------------------------------------------- host side
static int aux32_init(fhesi_ctx* ctx) {
  if (ctx->aux32) return 0;
  if (!ctx->pow2 || ctx->logn != A32_LOGN) FHESI_FAIL("aux32: only for n = 2^14");
  fhesi_aux32* x = new fhesi_aux32();
  // the four largest primes below 2^30 that are 1 mod 2^15 (= 2n)
  int found = 0;
  for (u64 k = ((u64)1 << 15) - 1; k > 0 && found < 4; --k) {
    const u64 cand = (k << 15) + 1;
    if (cand < ((u64)1 << 30) && hm::is_prime(cand)) x->pr.p[found++] = (u32)cand;
  }
  if (found < 4) { delete x; FHESI_FAIL("aux32: no primes"); }
  std::vector<Tw32> hf((size_t)4 * A32_N), hi((size_t)4 * A32_N);
  for (int a = 0; a < 4; ++a) {
    const u64 p = x->pr.p[a];
    u64 psi = 0;
    for (u64 gq = 2; gq < 1000 && !psi; ++gq) {
      const u64 cand = hm::powmod(gq, (p - 1) / (2 * A32_N), p);
      if (hm::powmod(cand, A32_N, p) == p - 1) psi = cand;
    }
    if (!psi) { delete x; FHESI_FAIL("aux32: no 2n-th root"); }
    const u64 ipsi = hm::invmod(psi, p);
    for (u64 idx = 0; idx < (u64)A32_N; ++idx) {
      const u64 e = hm::brv(idx, A32_LOGN);
      const u64 w = hm::powmod(psi, e, p), wi = hm::powmod(ipsi, e, p);
      hf[(size_t)a * A32_N + idx] = Tw32{(u32)w, (u32)((w << 32) / p)};
      hi[(size_t)a * A32_N + idx] = Tw32{(u32)wi, (u32)((wi << 32) / p)};
    }
    const u64 ninv = hm::invmod(A32_N % p, p);
    x->pr.ninv[a] = (u32)ninv;
    x->pr.ninv_p[a] = (u32)((ninv << 32) / p);
  }
  if (hipMalloc(&x->d_fwd, hf.size() * sizeof(Tw32)) != hipSuccess || hipMalloc(&x->d_inv, hi.size() * sizeof(Tw32)) != hipSuccess) { delete x; FHESI_FAIL("aux32: hipMalloc failed"); }
  HIP_TRY(hipMemcpy(x->d_fwd, hf.data(), hf.size() * sizeof(Tw32), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(x->d_inv, hi.data(), hi.size() * sizeof(Tw32), hipMemcpyHostToDevice));
  ctx->aux32 = x;
  return 0;
}
void aux32_free(fhesi_ctx* ctx) {
  if (!ctx->aux32) return;
  hipFree(ctx->aux32->d_fwd); hipFree(ctx->aux32->d_inv);
  delete ctx->aux32;
  ctx->aux32 = nullptr;
}

int launch_ntt32_fwd(fhesi_ctx* ctx, u32* d_rows, i64 count, int nslots, int a0) {
  FHESI_TRY(aux32_init(ctx));
  if (!count) return 0;
  ntt32_fwd_kernel<false><<<(unsigned)(count * nslots), A32_T, 0, ctx->stream>>>(d_rows, count, nslots, a0, ctx->aux32->pr, ctx->aux32->d_fwd, Dig32Src{});
  HIP_TRY(hipGetLastError());
  return 0;
}
int launch_ntt32_inv(fhesi_ctx* ctx, u32* d_rows, i64 count, int nslots, int a0) {
  FHESI_TRY(aux32_init(ctx));
  if (!count) return 0;
  ProfScope prof(ctx, PROF_NTT_INV, (double)(count * nslots));
  ntt32_inv_kernel<<<(unsigned)(count * nslots), A32_T, 0, ctx->stream>>>(d_rows, count, nslots, a0, ctx->aux32->pr, ctx->aux32->d_inv);
  HIP_TRY(hipGetLastError());
  return 0;
}
// digit rows [npolys * nd][4][n] u32 straight from the scaled-down parts
int launch_ntt32_fwd_digits(fhesi_ctx* ctx, const u64* d_parts, int nl, int digit_bits, int nd, i64 npolys, u32* d_out) {
  FHESI_TRY(aux32_init(ctx));
  if (!npolys) return 0;
  ProfScope prof(ctx, PROF_NTT_FWD, (double)(npolys * nd * 4));
  ProfScope main_prof(ctx, PROF_NTT_FWD_DIGITS_MAIN, (double)(npolys * nd * 4));
  ntt32_fwd_kernel<true><<<(unsigned)(npolys * nd * 4), A32_T, 0, ctx->stream>>>(d_out, npolys * nd, 4, 0, ctx->aux32->pr, ctx->aux32->d_fwd, Dig32Src{d_parts, nl, digit_bits, nd});
  HIP_TRY(hipGetLastError());
  return 0;
}
const u32* aux32_primes(fhesi_ctx* ctx) { return aux32_init(ctx) ? nullptr : ctx->aux32->pr.p; }

// Self-test (tests/test_gpu_ntt.py): the transform pair is a ring isomorphism of Z_p[X]/(X^n + 1) -- it is linear by construction, so
// checking that monomials multiply like monomials (X^i X^j = +-X^(i+j mod n)) and that inverse(forward(x)) = x pins it.
__global__ void aux32_pointwise_kernel(u32* __restrict__ a, const u32* __restrict__ b, u32 p) {
  const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < A32_N) a[j] = (u32)(((u64)a[j] * b[j]) % p);
}
extern "C" int fhesi_selftest_aux32(fhesi_ctx* c) {
  if (!c) FHESI_FAIL("null context");
  HIP_TRY(hipSetDevice(c->device));
  FHESI_TRY(aux32_init(c));
  u32 *da, *db;
  HIP_TRY(hipMalloc(&da, A32_N * 4)); HIP_TRY(hipMalloc(&db, A32_N * 4));
  std::vector<u32> ha(A32_N), hb(A32_N), hr(A32_N);
  int rc = 0;
  const int pairs[6][2] = {{0, 0}, {1, 2}, {5, 16383}, {16383, 16383}, {8192, 8192}, {4097, 12500}};
  for (int a = 0; a < 4 && !rc; ++a) {
    const u32 p = c->aux32->pr.p[a];
    for (int t = 0; t < 6 && !rc; ++t) {
      std::fill(ha.begin(), ha.end(), 0); std::fill(hb.begin(), hb.end(), 0);
      ha[pairs[t][0]] = 3; hb[pairs[t][1]] = 5;
      hipMemcpy(da, ha.data(), A32_N * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), A32_N * 4, hipMemcpyHostToDevice);
      if (launch_ntt32_fwd(c, da, 1, 1, a) || launch_ntt32_fwd(c, db, 1, 1, a)) { rc = 1; break; }
      aux32_pointwise_kernel<<<A32_N / 256, 256, 0, c->stream>>>(da, db, p);
      if (launch_ntt32_inv(c, da, 1, 1, a)) { rc = 1; break; }
      hipStreamSynchronize(c->stream);
      hipMemcpy(hr.data(), da, A32_N * 4, hipMemcpyDeviceToHost);
      const int e = pairs[t][0] + pairs[t][1];
      const int pos = e % A32_N;
      const u32 want = e >= A32_N ? p - 15 : 15;
      for (int j = 0; j < A32_N; ++j)
        if (hr[j] != (j == pos ? want : 0u)) { fhesi_set_error("aux32 self-test: prime %d, X^%d * X^%d: coefficient %d is %u", a, pairs[t][0], pairs[t][1], j, hr[j]); rc = 1; break; }
    }
    // round trip of a dense vector
    if (!rc) {
      for (int j = 0; j < A32_N; ++j) ha[j] = (u32)((1234567u * (u32)j + 89u) % p);
      hipMemcpy(da, ha.data(), A32_N * 4, hipMemcpyHostToDevice);
      if (launch_ntt32_fwd(c, da, 1, 1, a) || launch_ntt32_inv(c, da, 1, 1, a)) rc = 1;
      hipStreamSynchronize(c->stream);
      hipMemcpy(hr.data(), da, A32_N * 4, hipMemcpyDeviceToHost);
      for (int j = 0; j < A32_N && !rc; ++j) if (hr[j] != ha[j]) { fhesi_set_error("aux32 self-test: prime %d, round trip differs at %d", a, j); rc = 1; }
    }
  }
  hipFree(da); hipFree(db);
  return rc;
}
