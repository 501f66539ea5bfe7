// kernels_sample.hip -- on-device sampling for Encrypt and key generation: the randomness of FHESIPubKey::Encrypt
// (FHE-SI.cpp:14-25: binary r, two Gaussian noise polynomials), of KeySwitchSI::Init (FHE-SI.cpp:174-190: a uniform polynomial modulo
// 2^logQ and a Gaussian error per column), of sampleHWt and sampleGaussian (NumbTh.cpp:340-404), drawn from the counter-based generator
// of philox.h -- every number is a function of (seed, object index, coefficient index, purpose), so the device, the C oracle and the
// Python model produce the same polynomials and nothing crosses the host boundary.  Streaming kernels, a few integer operations per word.
#include "fhesi_internal.h"
#include "philox.h"

// rnd [count][3][n]: r (binary), e0, e1 (Gaussian, before the multiplication by p) of plaintext `first + c`
__global__ void __launch_bounds__(256) sample_encrypt_kernel(i64* __restrict__ rnd, i64 n, u64 seed, u64 first) {
  const i64 c = blockIdx.y;
  const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const u64 obj = first + (u64)c;
  i64* o = rnd + c * 3 * n + j;
  o[0] = phx_draw(seed, obj, (u32)j, PHX_BINARY, 0).w[0] & 1;
  o[n] = phx_gaussian(phx_draw(seed, obj, (u32)j, PHX_NOISE0, 0));
  o[2 * n] = phx_gaussian(phx_draw(seed, obj, (u32)j, PHX_NOISE1, 0));
}
// a [ncol][n][nl]: SampleRandom(poly, 2^logQ, n) as two's complement limbs;  err [ncol][n]: sampleGaussian -- of column `first + col`.
// The polynomial a is PUBLIC (it is the matrix's second row up to sign) and draws from `pub_seed`; the error is secret and draws from `seed`:
// a published pub_seed (keys compressed to a seed) says nothing about the errors.
__global__ void __launch_bounds__(256) sample_keygen_kernel(u64* __restrict__ a, i64* __restrict__ err, i64 n, int nl, int logQ, u64 seed, u64 pub_seed, u64 first) {
  const i64 col = blockIdx.y;
  const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const u64 obj = first + (u64)col;
  u64* x = a + (col * n + j) * nl;
  const int top = (logQ - 1) >> 6, tb = (logQ - 1) & 63;      // limb and bit of the sign position logQ - 1
  u64 sign = 0;
  for (int i = 0; i < nl; ++i) {
    u64 v = 0;
    if (i <= top) {
      const Philox4 d = phx_draw(pub_seed, obj, (u32)j, PHX_KEY_POLY, (u32)(i >> 1));
      v = (u64)d.w[2 * (i & 1)] | (u64)d.w[2 * (i & 1) + 1] << 32;
      if (i == top) {
        if (tb < 63) v &= (2ull << tb) - 1;                    // logQ random bits in all
        v ^= 1ull << tb;                                        // U - 2^(logQ-1) modulo 2^logQ: the top bit flipped ...
        sign = (v >> tb) & 1;
        if (tb < 63 && sign) v |= ~((2ull << tb) - 1);          // ... and the result sign-extended
      }
    } else v = sign ? ~0ull : 0ull;
    x[i] = v;
  }
  err[col * n + j] = phx_gaussian(phx_draw(seed, obj, (u32)j, PHX_KEY_ERR, 0));
}
// poly [n] (zeroed by the launcher): sampleHWt (NumbTh.cpp:340-360) -- a sequential rejection loop of Hwt accepted draws: one thread
__global__ void sample_hwt_kernel(i64* __restrict__ poly, i64 n, i64 hwt, u64 seed, u64 obj) {
  if (blockIdx.x || threadIdx.x) return;
  if (hwt > n) hwt = n;
  u32 t = 0;
  for (i64 i = 0; i < hwt; ++t) {
    const Philox4 d = phx_draw(seed, obj, t, PHX_HWT, 0);
    const u64 u = ((u64)d.w[0] | (u64)d.w[1] << 32) % (u64)n;
    if (poly[u] == 0) { poly[u] = (d.w[2] & 1) ? 1 : -1; ++i; }
  }
}
__global__ void __launch_bounds__(256) sample_gaussian_kernel(i64* __restrict__ poly, i64 n, u64 seed, u64 obj) {
  const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) poly[j] = phx_gaussian(phx_draw(seed, obj, (u32)j, PHX_GAUSS, 0));
}

// (the object index sits in gridDim.y: launches of at most 65535 objects, the index and the pointers advanced per launch)
int launch_sample_encrypt(fhesi_ctx* ctx, i64* d_rnd, i64 count, u64 seed, u64 first) {
  const i64 n = ctx->phim;
  for (i64 done = 0; done < count; done += 65535) {
    const i64 cnt = std::min<i64>(65535, count - done);
    sample_encrypt_kernel<<<dim3((unsigned)((n + 255) / 256), (unsigned)cnt), 256, 0, ctx->stream>>>(d_rnd + done * 3 * n, n, seed, first + (u64)done);
    HIP_TRY(hipGetLastError());
  }
  return 0;
}
int launch_sample_keygen(fhesi_ctx* ctx, u64* d_a, i64* d_err, i64 ncol, int nl, int logQ, u64 seed, u64 pub_seed, u64 first) {
  const i64 n = ctx->phim;
  for (i64 done = 0; done < ncol; done += 65535) {
    const i64 cnt = std::min<i64>(65535, ncol - done);
    sample_keygen_kernel<<<dim3((unsigned)((n + 255) / 256), (unsigned)cnt), 256, 0, ctx->stream>>>(d_a + done * n * nl, d_err + done * n, n, nl, logQ, seed, pub_seed, first + (u64)done);
    HIP_TRY(hipGetLastError());
  }
  return 0;
}
int launch_sample_poly(fhesi_ctx* ctx, i64* d_poly, int kind, i64 param, u64 seed, u64 obj) {
  if (kind == 0) {
    if (param < 0) FHESI_FAIL("sampleHWt: negative Hamming weight");
    HIP_TRY(hipMemsetAsync(d_poly, 0, (size_t)ctx->phim * 8, ctx->stream));
    sample_hwt_kernel<<<1, 1, 0, ctx->stream>>>(d_poly, ctx->phim, param, seed, obj);
  }
  else if (kind == 1) sample_gaussian_kernel<<<(unsigned)((ctx->phim + 255) / 256), 256, 0, ctx->stream>>>(d_poly, ctx->phim, seed, obj);
  else FHESI_FAIL("sample: unknown kind %d", kind);
  HIP_TRY(hipGetLastError());
  return 0;
}
