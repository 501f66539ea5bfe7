// kernels_ct.hip -- coefficient-domain ciphertext algebra on batches that stay in HBM: the pieces of Ciphertext.cpp that
// Matrix<Ciphertext> (Matrix.cpp) and Regression::Regress (Regression.h:102-149) call between multiplications.
//
// Unscaled ciphertext parts are [count][nparts][phi(m)][nlimbs] two's complement coefficients (centred mod 2^logQ); scaled-up
// ciphertexts (tProd) are residue rows [count][3][L][phi(m)].  All kernels are streaming (HBM-bound) integer work.
#include "fhesi_internal.h"

// Reduce (Util.cpp:3-26): sign-extend from bit logQ-1 (centred) or clear everything above it (positive)
__device__ __forceinline__ u64 reduce_limb(u64 val, int i, int logQ, u64 sbit) {
  const int bits_left = logQ - 64 * i;
  if (bits_left <= 0) return sbit ? ~0ull : 0ull;
  if (bits_left < 64) { const u64 mask = (1ull << bits_left) - 1; return sbit ? (val | ~mask) : (val & mask); }
  return val;
}

// Ciphertext::operator+= for unscaled ciphertexts (Ciphertext.cpp:123-134): parts[i] += other.parts[i]; ReduceCoefficients
template <int MAXNL>
__global__ void __launch_bounds__(256) ct_add_kernel(u64* __restrict__ dst, const u64* __restrict__ src, i64 ncoeffs, int nl, int logQ) {
  const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ncoeffs) return;
  u64 x[MAXNL];
  u64 carry = 0;
#pragma unroll
  for (int i = 0; i < MAXNL; ++i)
    if (i < nl) {
      const u64 a = dst[j * nl + i], b = src[j * nl + i];
      const u64 s = a + b, s2 = s + carry;
      carry = (s < a) | (s2 < s);
      x[i] = s2;
    }
  u64 sb = 0;
#pragma unroll
  for (int i = 0; i < MAXNL; ++i)
    if (i == (logQ - 1) >> 6) sb = (x[i] >> ((logQ - 1) & 63)) & 1;
#pragma unroll
  for (int i = 0; i < MAXNL; ++i)
    if (i < nl) dst[j * nl + i] = reduce_limb(x[i], i, logQ, sb);
}

// CiphertextPart::operator*=(long) (Ciphertext.cpp:21-27): coefficient * l, Reduce.  Two's complement product mod 2^(64 nl)
// (exact modulo 2^logQ, which divides it), then the centred residue.
template <int MAXNL>
__global__ void __launch_bounds__(256) ct_mul_long_kernel(u64* __restrict__ ct, i64 ncoeffs, int nl, int logQ, u64 mag, int negate) {
  const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ncoeffs) return;
  u64 x[MAXNL];
  u64 carry = 0;
#pragma unroll
  for (int i = 0; i < MAXNL; ++i)
    if (i < nl) {
      const u64 a = ct[j * nl + i];
      const u64 lo = a * mag, hi = d_mulhi(a, mag);
      const u64 s = lo + carry;
      carry = hi + (s < lo);
      x[i] = s;
    }
  if (negate) {
    u64 c = 1;
#pragma unroll
    for (int i = 0; i < MAXNL; ++i)
      if (i < nl) { const u64 v = ~x[i] + c; c = (c && v == 0); x[i] = v; }
  }
  u64 sb = 0;
#pragma unroll
  for (int i = 0; i < MAXNL; ++i)
    if (i == (logQ - 1) >> 6) sb = (x[i] >> ((logQ - 1) & 63)) & 1;
#pragma unroll
  for (int i = 0; i < MAXNL; ++i)
    if (i < nl) ct[j * nl + i] = reduce_limb(x[i], i, logQ, sb);
}

int launch_ct_add(fhesi_ctx* ctx, u64* d_dst, const u64* d_src, i64 ncoeffs, int nl, int logQ) {
  if (!ncoeffs) return 0;
  if (nl > 32) FHESI_FAIL("ciphertext coefficients of %d limbs exceed the supported 32", nl);
  const unsigned grid = (unsigned)((ncoeffs + 255) / 256);
  if (nl <= 2) ct_add_kernel<2><<<grid, 256, 0, ctx->stream>>>(d_dst, d_src, ncoeffs, nl, logQ);
  else if (nl <= 8) ct_add_kernel<8><<<grid, 256, 0, ctx->stream>>>(d_dst, d_src, ncoeffs, nl, logQ);
  else if (nl <= 16) ct_add_kernel<16><<<grid, 256, 0, ctx->stream>>>(d_dst, d_src, ncoeffs, nl, logQ);
  else ct_add_kernel<32><<<grid, 256, 0, ctx->stream>>>(d_dst, d_src, ncoeffs, nl, logQ);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---- Ciphertext::operator>>= on the coefficients themselves.
// The reference takes X -> X^k in evaluation form (DoubleCRT::automorph, DoubleCRT.cpp:439-465, between DoubleCRT(poly) and toPoly:
// Ciphertext.cpp:54-59); its result is the INTEGER polynomial a(X^k) mod Phi_m (the coefficients, at most two input coefficients
// added, stay far below half the chain product).  On the rings below that polynomial is a signed gather:
//   m = 2n (power of two):  X^n = -1:              out_i = +-a_j,  j k = i or i + n (mod 2n)
//   m = 2q', q' an odd prime (the reference's safe-prime rings), phi = q' - 1:  X^q' = -1 modulo X^q' + 1 = (X + 1) Phi_m,
//       R_i = +-a_j with j k = i (mod q'), sign - when j k mod 2q' >= q';  out_i = R_i - (-1)^i R_(q'-1)   (Phi_m = sum (-X)^i, monic)
//   m prime, phi = m - 1:   X^m = 1,  R_i = a_j with j k = i (mod m);  out_i = R_i - R_(m-1)   (Phi_m = sum X^i)
// so no row transform is needed at all: out = positive residue modulo 2^logQ, limb-major, as ByteDecompPart takes it (Ciphertext.cpp:94).
// mode: 0 power of two, 1 m = 2 prime, 2 m prime.  in [npolys][n][nl_in] two's complement;  out [npolys][nlq][n] (logQ > 0), or the
// integers themselves, sign-extended, coefficient-major [npolys][n][nlq] (logQ = 0: toPoly of Ciphertext >>= without a key switch).
__global__ void __launch_bounds__(256) ct_automorph_parts_kernel(const u64* __restrict__ in, i64 n, int nl_in, i64 m, i64 kk, i64 kinv, int mode, int logQ,
                                                                 u64* __restrict__ out, int nlq) {
  const i64 poly = blockIdx.y;
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // term 1: s1 a[j1], term 2: s2 a[j2]  (s = 0: absent)
  i64 j1 = 0, j2 = 0;
  int s1 = 0, s2 = 0;
  if (mode == 0) {
    const i64 j = (i64)(((u128)i * (u128)kinv) % (u128)m);          // j k = i (mod 2n)
    j1 = j < n ? j : j - n;
    s1 = j < n ? 1 : -1;
  } else if (mode == 1) {
    const i64 q = m / 2;
    auto term = [&](i64 e, i64& j, int& sg) {                        // the a_j with j k = e (mod q'), and its sign
      j = (i64)(((u128)e * (u128)(kinv % q)) % (u128)q);
      if (j > q - 2) { sg = 0; j = 0; return; }
      sg = ((i64)(((u128)j * (u128)kk) % (u128)m) >= q) ? -1 : 1;
    };
    term(i, j1, s1);
    term(q - 1, j2, s2);
    s2 = (i & 1) ? s2 : -s2;                                        // - (-1)^i R_(q'-1)
  } else {
    auto term = [&](i64 e, i64& j, int& sg) {
      j = (i64)(((u128)e * (u128)kinv) % (u128)m);
      sg = j > m - 2 ? 0 : 1;
      if (!sg) j = 0;
    };
    term(i, j1, s1);
    term(m - 1, j2, s2);
    s2 = -s2;
  }
  const u64* __restrict__ a1 = in + (poly * n + j1) * nl_in;
  const u64* __restrict__ a2 = in + (poly * n + j2) * nl_in;
  const u64 ext1 = (a1[nl_in - 1] >> 63) ? ~0ull : 0ull, ext2 = (a2[nl_in - 1] >> 63) ? ~0ull : 0ull;
  u64 carry = (s1 < 0 ? 1 : 0) + (s2 < 0 ? 1 : 0);                  // the +1 of every two's complement negation
  u64* __restrict__ o = logQ ? out + poly * nlq * n + i : out + (poly * n + i) * nlq;
  for (int w = 0; w < nlq; ++w) {
    u64 x1 = w < nl_in ? a1[w] : ext1, x2 = w < nl_in ? a2[w] : ext2;
    x1 = s1 == 0 ? 0 : (s1 < 0 ? ~x1 : x1);
    x2 = s2 == 0 ? 0 : (s2 < 0 ? ~x2 : x2);
    const u128 sum = (u128)x1 + x2 + carry;
    u64 v = (u64)sum;
    carry = (u64)(sum >> 64);
    if (logQ) {
      const int bits_left = logQ - 64 * w;
      if (bits_left < 64) v &= ((u64)1 << bits_left) - 1;
      o[(i64)w * n] = v;
    } else o[w] = v;
  }
}
// 0 = launched; 2 = this ring / exponent is not covered (the caller keeps the evaluation-form path)
int launch_ct_automorph_parts(fhesi_ctx* ctx, const u64* d_in, int nl_in, i64 npolys, i64 kk, int logQ, u64* d_parts, int nlq) {
  const i64 m = ctx->m, n = ctx->phim;
  int mode;
  if (ctx->pow2) mode = 0;
  else if (m % 2 == 0 && (m / 2) % 2 == 1 && hm::is_prime((u64)(m / 2)) && n == m / 2 - 1) mode = 1;
  else if (hm::is_prime((u64)m) && n == m - 1) mode = 2;
  else return 2;
  if (!npolys) return 0;
  // k^-1 mod m (k in Z_m^*: checked by the caller)
  i64 kinv = 0;
  {
    i64 a = kk % m, b = m, x0 = 1, x1 = 0;
    while (b) { const i64 qq = a / b; i64 t = a - qq * b; a = b; b = t; t = x0 - qq * x1; x0 = x1; x1 = t; }
    if (a != 1) FHESI_FAIL("automorph: k=%lld is not in Zm*", (long long)kk);
    kinv = ((x0 % m) + m) % m;
  }
  dim3 grid((unsigned)((n + 255) / 256), (unsigned)npolys);
  ct_automorph_parts_kernel<<<grid, 256, 0, ctx->stream>>>(d_in, n, nl_in, m, kk % m, kinv, mode, logQ, d_parts, nlq);
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_ct_mul_long(fhesi_ctx* ctx, u64* d_ct, i64 ncoeffs, int nl, int logQ, i64 l) {
  if (!ncoeffs) return 0;
  if (nl > 32) FHESI_FAIL("ciphertext coefficients of %d limbs exceed the supported 32", nl);
  const u64 mag = l < 0 ? (u64)(-(l + 1)) + 1 : (u64)l;
  const int neg = l < 0;
  const unsigned grid = (unsigned)((ncoeffs + 255) / 256);
  if (nl <= 2) ct_mul_long_kernel<2><<<grid, 256, 0, ctx->stream>>>(d_ct, ncoeffs, nl, logQ, mag, neg);
  else if (nl <= 8) ct_mul_long_kernel<8><<<grid, 256, 0, ctx->stream>>>(d_ct, ncoeffs, nl, logQ, mag, neg);
  else if (nl <= 16) ct_mul_long_kernel<16><<<grid, 256, 0, ctx->stream>>>(d_ct, ncoeffs, nl, logQ, mag, neg);
  else ct_mul_long_kernel<32><<<grid, 256, 0, ctx->stream>>>(d_ct, ncoeffs, nl, logQ, mag, neg);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---- Ciphertext::operator+=(const ZZX&) on unscaled ciphertexts (Ciphertext.cpp:147-156): scaledConstant_j = (other_j << logQ) / p
// with NTL's floor division, parts[0] += scaledConstant, ReduceCoefficients.  poly: [npoly][n] machine-word coefficients (npoly = 1:
// one constant for the whole batch).  Only the quotient modulo 2^logQ matters; it is formed limb by limb from the top by 128 / 64-bit
// long division of |other_j| 2^logQ, and a negative constant takes -(q + [remainder != 0]) (floor, not truncation).
template <int MAXNL>
__global__ void __launch_bounds__(256) ct_add_const_kernel(u64* __restrict__ ct, const i64* __restrict__ poly, int npoly, i64 n, int nparts, int nl, int logQ, u64 p) {
  const i64 c = blockIdx.y;
  const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const i64 cv = poly[(npoly == 1 ? 0 : c) * n + j];
  const u64 mag = cv < 0 ? (u64)(-(cv + 1)) + 1 : (u64)cv;
  // |cv| 2^logQ occupies limbs sh .. sh + 1 (sh = logQ / 64), shifted by logQ % 64 bits
  const int sh = logQ >> 6, bs = logQ & 63;
  const u64 lo = bs ? mag << bs : mag, hi = bs ? mag >> (64 - bs) : 0;
  u64 q[MAXNL];
#pragma unroll
  for (int i = 0; i < MAXNL; ++i) q[i] = 0;
  u64 rem = 0;
  for (int i = sh + 1; i >= 0; --i) {
    const u64 limb = i == sh + 1 ? hi : (i == sh ? lo : 0);
    const u128 cur = ((u128)rem << 64) | limb;
    const u64 ql = (u64)(cur / p);
    rem = (u64)(cur % p);
#pragma unroll
    for (int t = 0; t < MAXNL; ++t) if (t == i) q[t] = ql;       // (limbs at or above nl only matter modulo 2^logQ: dropped)
  }
  if (cv < 0) {
    // -(q + (rem != 0)) in two's complement
    u64 carry = rem != 0;
#pragma unroll
    for (int t = 0; t < MAXNL; ++t) { const u64 v = q[t] + carry; carry = v < carry; q[t] = v; }
    u64 cneg = 1;
#pragma unroll
    for (int t = 0; t < MAXNL; ++t) { const u64 v = ~q[t] + cneg; cneg = (cneg && v == 0); q[t] = v; }
  }
  u64* x0 = ct + ((c * nparts) * n + j) * nl;                     // part 0 of ciphertext c
  u64 x[MAXNL];
  u64 carry = 0;
#pragma unroll
  for (int i = 0; i < MAXNL; ++i)
    if (i < nl) { const u64 a = x0[i], s1 = a + q[i], s2 = s1 + carry; carry = (s1 < a) | (s2 < s1); x[i] = s2; }
  u64 sb = 0;
#pragma unroll
  for (int i = 0; i < MAXNL; ++i)
    if (i == (logQ - 1) >> 6) sb = (x[i] >> ((logQ - 1) & 63)) & 1;
#pragma unroll
  for (int i = 0; i < MAXNL; ++i)
    if (i < nl) x0[i] = reduce_limb(x[i], i, logQ, sb);
}
int launch_ct_add_const(fhesi_ctx* ctx, u64* d_ct, const i64* d_poly, int npoly, int nparts, int nl, int logQ, u64 p, i64 count) {
  if (!count) return 0;
  if (nl > 32) FHESI_FAIL("ciphertext coefficients of %d limbs exceed the supported 32", nl);
  if (count > 65535) FHESI_FAIL("Ciphertext += ZZX: more than 65535 ciphertexts per call");
  const dim3 grid((unsigned)((ctx->phim + 255) / 256), (unsigned)count);
  // (the quotient needs limbs 0 .. logQ / 64 + 1 <= nl + 1: the instantiation one step above nl holds them)
  if (nl < 2) ct_add_const_kernel<4><<<grid, 256, 0, ctx->stream>>>(d_ct, d_poly, npoly, ctx->phim, nparts, nl, logQ, p);
  else if (nl < 7) ct_add_const_kernel<8><<<grid, 256, 0, ctx->stream>>>(d_ct, d_poly, npoly, ctx->phim, nparts, nl, logQ, p);
  else if (nl < 15) ct_add_const_kernel<16><<<grid, 256, 0, ctx->stream>>>(d_ct, d_poly, npoly, ctx->phim, nparts, nl, logQ, p);
  else ct_add_const_kernel<34><<<grid, 256, 0, ctx->stream>>>(d_ct, d_poly, npoly, ctx->phim, nparts, nl, logQ, p);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---- gather: out[i] = pool[idx[i]], `words` u64 per element (operands of one wave of Matrix products)
__global__ void __launch_bounds__(256) gather_kernel(const u64* __restrict__ pool, const int* __restrict__ idx, u64* __restrict__ out, i64 words) {
  const u64* __restrict__ s = pool + (i64)idx[blockIdx.y] * words;
  u64* __restrict__ d = out + (i64)blockIdx.y * words;
  for (i64 w = (i64)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += (i64)gridDim.x * blockDim.x) d[w] = s[w];
}
int launch_gather(fhesi_ctx* ctx, const u64* d_pool, const int* d_idx, i64 count, i64 words, u64* d_out) {
  if (!count || !words) return 0;
  unsigned gx = (unsigned)((words + 255) / 256);
  if (gx > 256) gx = 256;
  for (i64 done = 0; done < count; done += 65535) {
    const i64 cnt = count - done < 65535 ? count - done : 65535;
    gather_kernel<<<dim3(gx, (unsigned)cnt), 256, 0, ctx->stream>>>(d_pool, d_idx + done, d_out + done * words, words);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---- segmented sum of scaled-up ciphertexts: out[g] = sum_{t in [seg[g], seg[g+1])} in[t]  (the `newMatrix(i,j) += tmp` loops of
// Matrix.cpp:62-72,157-167 and `det += tmp` of :243; Ciphertext::operator+= on tProd, Ciphertext.cpp:135-142 -> DoubleCRT +=)
__global__ void __launch_bounds__(256) segment_sum_kernel(const u64* __restrict__ in, const int* __restrict__ seg, u64* __restrict__ out, int ncomp_L,
                                                          i64 n, const PrimeConst* __restrict__ pcs, int L) {
  const int g = blockIdx.z, r = blockIdx.y;       // r = comp * L + prime
  const u64 q = pcs[r % L].q;
  const i64 ct_words = (i64)ncomp_L * n;
  const int t0 = seg[g], t1 = seg[g + 1];
  for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x) {
    u64 acc = 0;
    for (int t = t0; t < t1; ++t) {
      acc += in[(i64)t * ct_words + (i64)r * n + j];
      if (acc >= q) acc -= q;
    }
    out[(i64)g * ct_words + (i64)r * n + j] = acc;
  }
}
int launch_segment_sum(fhesi_ctx* ctx, const u64* d_in, const int* d_seg, i64 ngroups, int ncomp, u64* d_out) {
  if (!ngroups) return 0;
  unsigned gx = (unsigned)((ctx->phim + 255) / 256);
  if (gx > 64) gx = 64;
  for (i64 done = 0; done < ngroups; done += 65535) {
    const i64 cnt = ngroups - done < 65535 ? ngroups - done : 65535;
    segment_sum_kernel<<<dim3(gx, (unsigned)(ncomp * ctx->L), (unsigned)cnt), 256, 0, ctx->stream>>>(d_in, d_seg + done, d_out + done * ncomp * ctx->L * ctx->phim,
                                                                                                    ncomp * ctx->L, ctx->phim, ctx->d_pc, ctx->L);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---- FHESIPubKey::Encrypt (FHE-SI.cpp:26-27): ct[i] = pk[i] * r + e[i]  for i = 0,1, all in evaluation form.
// rows: [count][3][L][n] = (r, e0*p, e1*p); pk: [2][L][n]; out: [count][2][L][n]
__global__ void __launch_bounds__(256) encrypt_combine_kernel(const u64* __restrict__ rows, const u64* __restrict__ pk, u64* __restrict__ out, int L, i64 n,
                                                              const PrimeConst* __restrict__ pcs) {
  const i64 c = blockIdx.z;
  const int l = blockIdx.y;
  const PrimeConst pc = pcs[l];
  const u64* r = rows + ((c * 3 + 0) * L + l) * n;
  for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x) {
    const u64 rv = r[j];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const u64 e = rows[((c * 3 + 1 + i) * L + l) * n + j];
      out[((c * 2 + i) * L + l) * n + j] = d_addmod(d_mulmod(pk[((i64)i * L + l) * n + j], rv, pc), e, pc.q);
    }
  }
}
int launch_encrypt_combine(fhesi_ctx* ctx, const u64* d_rows, const u64* d_pk, i64 count, u64* d_out) {
  if (!count) return 0;
  unsigned gx = (unsigned)((ctx->phim + 255) / 256);
  if (gx > 64) gx = 64;
  for (i64 done = 0; done < count; done += 65535) {
    const i64 cnt = count - done < 65535 ? count - done : 65535;
    encrypt_combine_kernel<<<dim3(gx, (unsigned)ctx->L, (unsigned)cnt), 256, 0, ctx->stream>>>(d_rows + done * 3 * ctx->L * ctx->phim, d_pk, d_out + done * 2 * ctx->L * ctx->phim,
                                                                                              ctx->L, ctx->phim, ctx->d_pc);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---- ctxt[0] += delta * msg; ReduceCoefficients (FHE-SI.cpp:31-35).  ct: [count][2][n][nl]; msg: [count][n] small non-negative;
// delta = floor(2^logQ / p) as nl limbs.  The product is taken modulo 2^(64 nl), exact modulo 2^logQ.
template <int MAXNL>
__global__ void __launch_bounds__(256) add_scaled_msg_kernel(u64* __restrict__ ct, const i64* __restrict__ msg, const u64* __restrict__ delta, i64 n, int nl, int logQ) {
  const i64 c = blockIdx.y;
  const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const u64 mv = (u64)msg[c * n + j];
  u64* x = ct + ((c * 2) * n + j) * nl;
  u64 v[MAXNL];
  u64 mc = 0, ac = 0;
#pragma unroll
  for (int i = 0; i < MAXNL; ++i)
    if (i < nl) {
      const u64 dl = delta[i];
      const u64 lo = dl * mv, hi = d_mulhi(dl, mv);
      const u64 pr = lo + mc;
      mc = hi + (pr < lo);
      const u64 a = x[i], s = a + pr, s2 = s + ac;
      ac = (s < a) | (s2 < s);
      v[i] = s2;
    }
  u64 sb = 0;
#pragma unroll
  for (int i = 0; i < MAXNL; ++i)
    if (i == (logQ - 1) >> 6) sb = (v[i] >> ((logQ - 1) & 63)) & 1;
#pragma unroll
  for (int i = 0; i < MAXNL; ++i)
    if (i < nl) x[i] = reduce_limb(v[i], i, logQ, sb);
}
int launch_add_scaled_msg(fhesi_ctx* ctx, u64* d_ct, const i64* d_msg, const u64* d_delta, i64 count, int nl, int logQ) {
  if (!count) return 0;
  if (nl > 32) FHESI_FAIL("ciphertext coefficients of %d limbs exceed the supported 32", nl);
  const dim3 grid((unsigned)((ctx->phim + 255) / 256), (unsigned)count);
  if (count > 65535) FHESI_FAIL("Encrypt: more than 65535 plaintexts per call");
  if (nl <= 2) add_scaled_msg_kernel<2><<<grid, 256, 0, ctx->stream>>>(d_ct, d_msg, d_delta, ctx->phim, nl, logQ);
  else if (nl <= 8) add_scaled_msg_kernel<8><<<grid, 256, 0, ctx->stream>>>(d_ct, d_msg, d_delta, ctx->phim, nl, logQ);
  else if (nl <= 16) add_scaled_msg_kernel<16><<<grid, 256, 0, ctx->stream>>>(d_ct, d_msg, d_delta, ctx->phim, nl, logQ);
  else add_scaled_msg_kernel<32><<<grid, 256, 0, ctx->stream>>>(d_ct, d_msg, d_delta, ctx->phim, nl, logQ);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---- FHESISecKey::Decrypt (FHE-SI.cpp:105-107): z = c0 + c1 * t in evaluation form.  rows: [count][2][L][n]; t: [L][n]; out [count][L][n]
__global__ void __launch_bounds__(256) decrypt_dot_kernel(const u64* __restrict__ rows, const u64* __restrict__ t, u64* __restrict__ out, int L, i64 n,
                                                          const PrimeConst* __restrict__ pcs) {
  const i64 c = blockIdx.z;
  const int l = blockIdx.y;
  const PrimeConst pc = pcs[l];
  for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x)
    out[(c * L + l) * n + j] = d_addmod(rows[((c * 2) * L + l) * n + j], d_mulmod(rows[((c * 2 + 1) * L + l) * n + j], t[(i64)l * n + j], pc), pc.q);
}
int launch_decrypt_dot(fhesi_ctx* ctx, const u64* d_rows, const u64* d_t, i64 count, u64* d_out) {
  if (!count) return 0;
  unsigned gx = (unsigned)((ctx->phim + 255) / 256);
  if (gx > 64) gx = 64;
  for (i64 done = 0; done < count; done += 65535) {
    const i64 cnt = count - done < 65535 ? count - done : 65535;
    decrypt_dot_kernel<<<dim3(gx, (unsigned)ctx->L, (unsigned)cnt), 256, 0, ctx->stream>>>(d_rows + done * 2 * ctx->L * ctx->phim, d_t, d_out + done * ctx->L * ctx->phim, ctx->L,
                                                                                          ctx->phim, ctx->d_pc);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---- the rounding of Decrypt (FHE-SI.cpp:110-116): m = floor((2 p z + q) / (2 q)) mod p, q = 2^logQ.
// Writing z = zh * 2q + zl with zl = z mod 2q in [0, 2q) gives floor(...) = 2 p zh + floor((2 p zl + q) / 2q), and the first
// term vanishes modulo p: only the low logQ+1 bits of z matter, and the quotient is at most 2p.
// z: [npolys][n][nw] two's complement truncated to nw = ceil((logQ+1)/64) limbs; out: [npolys][n]
template <int MAXNL>
__global__ void __launch_bounds__(256) decrypt_round_kernel(const u64* __restrict__ z, i64 total, int nw, int logQ, u64 p, i64* __restrict__ out) {
  const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= total) return;
  const int top_bits = logQ + 1 - 64 * (nw - 1);                  // bits of zl in its top limb (1..64)
  const u64 twop = 2 * p;
  u64 P[MAXNL + 1];
  u64 carry = 0;
#pragma unroll
  for (int i = 0; i < MAXNL; ++i) {
    P[i] = 0;
    if (i < nw) {
      u64 xi = z[j * nw + i];
      if (i == nw - 1 && top_bits < 64) xi &= (1ull << top_bits) - 1;
      const u64 lo = xi * twop, hi = d_mulhi(xi, twop);
      const u64 s = lo + carry;
      carry = hi + (s < lo);
      P[i] = s;
    }
  }
  // limb nw of the product is `carry`; add q = 2^logQ with carry propagation
  const int wq = logQ >> 6;
  u64 add = 1ull << (logQ & 63);
#pragma unroll
  for (int i = 0; i < MAXNL; ++i)
    if (i < nw && i >= wq) { const u64 s = P[i] + add; add = s < add; P[i] = s; }
  carry += add;
  // bits from logQ+1 upward: at most 2p, fits one word
  const int ws = (logQ + 1) >> 6, bs = (logQ + 1) & 63;
  u64 lo = 0, hi = 0;
#pragma unroll
  for (int i = 0; i < MAXNL; ++i) {
    if (i == ws) lo = (i < nw) ? P[i] : carry;
    if (i == ws + 1) hi = (i < nw) ? P[i] : (i == nw ? carry : 0);
  }
  if (ws == MAXNL) lo = carry;
  if (ws + 1 == MAXNL) hi = (nw == MAXNL) ? carry : hi;
  const u64 t = bs ? ((lo >> bs) | (hi << (64 - bs))) : lo;
  out[j] = (i64)(t % p);
}
int launch_decrypt_round(fhesi_ctx* ctx, const u64* d_z, i64 total, int nw, int logQ, u64 p, i64* d_out) {
  if (!total) return 0;
  if (nw > 32) FHESI_FAIL("Decrypt: logQ=%d exceeds the supported 2047 bits", logQ);
  const unsigned grid = (unsigned)((total + 255) / 256);
  if (nw <= 2) decrypt_round_kernel<2><<<grid, 256, 0, ctx->stream>>>(d_z, total, nw, logQ, p, d_out);
  else if (nw <= 9) decrypt_round_kernel<9><<<grid, 256, 0, ctx->stream>>>(d_z, total, nw, logQ, p, d_out);
  else if (nw <= 17) decrypt_round_kernel<17><<<grid, 256, 0, ctx->stream>>>(d_z, total, nw, logQ, p, d_out);
  else decrypt_round_kernel<32><<<grid, 256, 0, ctx->stream>>>(d_z, total, nw, logQ, p, d_out);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---- KeySwitchSI::Init in batches (FHE-SI.cpp:153-209) -------------------------------------------------------------------------------
// b = A * t for every column: dst[col][l][j] = a[col][l][j] * t[l][j] mod q_l   (`b[ind] = A[ind]; b[ind] *= t`, :180-184)
__global__ void __launch_bounds__(256) rows_mul_bcast_kernel(u64* __restrict__ dst, const u64* __restrict__ a, const u64* __restrict__ t, int L, i64 n,
                                                              const PrimeConst* __restrict__ pcs) {
  const i64 col = blockIdx.z;
  const int l = blockIdx.y;
  const PrimeConst pc = pcs[l];
  const u64* ar = a + (col * L + l) * n;
  const u64* tr = t + (i64)l * n;
  u64* dr = dst + (col * L + l) * n;
  for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x) dr[j] = d_mulmod(ar[j], tr[j], pc);
}
int launch_rows_mul_bcast(fhesi_ctx* ctx, u64* d_dst, const u64* d_a, const u64* d_t, i64 ncols) {
  if (!ncols) return 0;
  if (ncols > 65535) FHESI_FAIL("KeySwitchSI::Init: more than 65535 columns");
  unsigned gx = (unsigned)((ctx->phim + 255) / 256);
  if (gx > 64) gx = 64;
  rows_mul_bcast_kernel<<<dim3(gx, (unsigned)ctx->L, (unsigned)ncols), 256, 0, ctx->stream>>>(d_dst, d_a, d_t, ctx->L, ctx->phim, ctx->d_pc);
  HIP_TRY(hipGetLastError());
  return 0;
}
// bCoeff += err; bCoeff += sCoeff[i] (already shifted left by digit_bits * j); ReduceCoefficients(bCoeff, logQ)   (:192-203)
//   bcoef: [ncol][n][wb] two's complement (toPoly of b), scoef: [nsrc][n][ws] (toPoly of the source key components), err: [ncol][n];
//   out: [ncol][n][nl] centred modulo 2^logQ.  Only the bits below logQ of every term matter, so all arithmetic is modulo 2^(64 nl).
template <int MAXNL>
__global__ void __launch_bounds__(256) keygen_combine_kernel(const u64* __restrict__ bcoef, int wb, const u64* __restrict__ scoef, int ws, const i64* __restrict__ err,
                                                              i64 n, int nd, int digit_bits, int nl, int logQ, u64* __restrict__ out) {
  const i64 col = blockIdx.y;
  const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const i64 comp = col / nd;
  const int sh = (int)(col % nd) * digit_bits, wsh = sh >> 6, bsh = sh & 63;
  const u64* bx = bcoef + (col * n + j) * wb;
  const u64* sx = scoef + (comp * n + j) * ws;
  const i64 e = err[col * n + j];
  const u64 bsign = (bx[wb - 1] >> 63) ? ~0ull : 0, ssign = (sx[ws - 1] >> 63) ? ~0ull : 0, esign = e < 0 ? ~0ull : 0;
  u64 v[MAXNL];
  u64 carry = 0;
#pragma unroll
  for (int i = 0; i < MAXNL; ++i)
    if (i < nl) {
      const u64 b = i < wb ? bx[i] : bsign;
      const u64 ev = i == 0 ? (u64)e : esign;
      // limb i of (s << sh): limbs i - wsh and i - wsh - 1 of the sign-extended s
      auto slimb = [&](int k) -> u64 { return k < 0 ? 0 : (k < ws ? sx[k] : ssign); };
      const u64 hi = slimb(i - wsh), lo = slimb(i - wsh - 1);
      const u64 sv = bsh ? ((hi << bsh) | (lo >> (64 - bsh))) : hi;
      const u128 t = (u128)b + ev + sv + carry;
      v[i] = (u64)t; carry = (u64)(t >> 64);
    }
  u64 sb = 0;
#pragma unroll
  for (int i = 0; i < MAXNL; ++i)
    if (i == (logQ - 1) >> 6) sb = (v[i] >> ((logQ - 1) & 63)) & 1;
  u64* o = out + (col * n + j) * nl;
#pragma unroll
  for (int i = 0; i < MAXNL; ++i)
    if (i < nl) o[i] = reduce_limb(v[i], i, logQ, sb);
}
int launch_keygen_combine(fhesi_ctx* ctx, const u64* d_bcoef, int wb, const u64* d_scoef, int ws, const i64* d_err, i64 ncols, int nd, int digit_bits, int nl, int logQ, u64* d_out) {
  if (!ncols) return 0;
  if (nl > 32) FHESI_FAIL("KeySwitchSI::Init: coefficients of %d limbs exceed the supported 32", nl);
  const dim3 grid((unsigned)((ctx->phim + 255) / 256), (unsigned)ncols);
  if (nl <= 2) keygen_combine_kernel<2><<<grid, 256, 0, ctx->stream>>>(d_bcoef, wb, d_scoef, ws, d_err, ctx->phim, nd, digit_bits, nl, logQ, d_out);
  else if (nl <= 8) keygen_combine_kernel<8><<<grid, 256, 0, ctx->stream>>>(d_bcoef, wb, d_scoef, ws, d_err, ctx->phim, nd, digit_bits, nl, logQ, d_out);
  else if (nl <= 16) keygen_combine_kernel<16><<<grid, 256, 0, ctx->stream>>>(d_bcoef, wb, d_scoef, ws, d_err, ctx->phim, nd, digit_bits, nl, logQ, d_out);
  else keygen_combine_kernel<32><<<grid, 256, 0, ctx->stream>>>(d_bcoef, wb, d_scoef, ws, d_err, ctx->phim, nd, digit_bits, nl, logQ, d_out);
  HIP_TRY(hipGetLastError());
  return 0;
}
