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
