// kernels_ew.hip -- coefficient-wise DoubleCRT kernels (HBM-bound streaming kernels, 16 B per lane).
//   ew_op        DoubleCRT::Op(DoubleCRT)            DoubleCRT.cpp:79-113   (AddMod / SubMod / MulMod per element)
//   ew_scalar    DoubleCRT::Op(ZZ), /=, =(ZZ)        DoubleCRT.cpp:115-129, 407-420, 333-347
//   tensor2x2    Ciphertext::operator*= tensor step   Ciphertext.cpp:179-186 (t0=a0b0, t1=a0b1+a1b0, t2=a1b1, fused)
//   dot_accum    DotProduct<DoubleCRT>                Util.h:79-98 as used by FHE-SI.cpp:253-254 (fused multiply-accumulate)
//   automorph    DoubleCRT::automorph                 DoubleCRT.cpp:439-465
#include "fhesi_internal.h"

typedef u64 u64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u64 ew_apply(u64 a, u64 b, const PrimeConst& pc, int op) {
  switch (op) {
    case 0: return d_addmod(a, b, pc.q);
    case 1: return d_submod(a, b, pc.q);
    default: return d_mulmod(a, b, pc);
  }
}

// rows: [nrows][n]; prime of row r = prime_of_slot[r % nslots].  grid.y = row, grid.x strides the row.
template <int OP>
__global__ void __launch_bounds__(256) ew_op_kernel(u64* __restrict__ dst, const u64* __restrict__ src, i64 n, int nslots,
                                                     const int* __restrict__ prime_of_slot, const PrimeConst* __restrict__ pcs) {
  const i64 row = blockIdx.y;
  const int slot = (int)(row % nslots);
  const PrimeConst pc = pcs[prime_of_slot ? prime_of_slot[slot] : slot];
  u64* d = dst + row * n;
  const u64* s = src + row * n;
  const i64 n2 = n >> 1;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (i64)gridDim.x * blockDim.x) {
    u64x2 a = ((const u64x2*)d)[i], b = ((const u64x2*)s)[i];
    a.x = ew_apply(a.x, b.x, pc, OP);
    a.y = ew_apply(a.y, b.y, pc, OP);
    ((u64x2*)d)[i] = a;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) d[n - 1] = ew_apply(d[n - 1], s[n - 1], pc, OP);
}

// scalar ops: scalars[slot] already reduced mod the slot's prime (and inverted for DIV) on the host.
template <int OP>
__global__ void __launch_bounds__(256) ew_scalar_kernel(u64* __restrict__ dst, const u64* __restrict__ scalars, i64 n, int nslots,
                                                         const int* __restrict__ prime_of_slot, const PrimeConst* __restrict__ pcs) {
  const i64 row = blockIdx.y;
  const int slot = (int)(row % nslots);
  const PrimeConst pc = pcs[prime_of_slot ? prime_of_slot[slot] : slot];
  const u64 sc = scalars[slot];
  u64* d = dst + row * n;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
    if (OP == FHESI_OP_SET_) d[i] = sc;
    else d[i] = ew_apply(d[i], sc, pc, OP);
  }
}

// DoubleCRT::Exp (DoubleCRT.cpp:423-434): d[i] <- d[i]^e mod q, exps[slot] = the (per-prime, non-negative) exponent.
// CHECK: only count zero elements into *flag (the inverse of 0 is NTL's InvMod error for negative exponents).
template <bool CHECK>
__global__ void __launch_bounds__(256) ew_exp_kernel(u64* __restrict__ dst, const u64* __restrict__ exps, i64 n, int nslots,
                                                      const int* __restrict__ prime_of_slot, const PrimeConst* __restrict__ pcs, unsigned* __restrict__ flag) {
  const i64 row = blockIdx.y;
  const int slot = (int)(row % nslots);
  const PrimeConst pc = pcs[prime_of_slot ? prime_of_slot[slot] : slot];
  const u64 e = exps[slot];
  u64* d = dst + row * n;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
    if (CHECK) { if (d[i] == 0) atomicOr(flag, 1u); continue; }
    u64 base = d[i], r = 1;
    for (u64 k = e; k; k >>= 1) {           // e is wave-uniform: no divergence
      if (k & 1) r = d_mulmod(r, base, pc);
      base = d_mulmod(base, base, pc);
    }
    d[i] = r;
  }
}

// ca: [count][2][L][n] = NTT(a0*p), NTT(a1*p);  cb: [count][2][L][n] = NTT(b0), NTT(b1);  t: [count][3][L][n]
__global__ void __launch_bounds__(256) tensor2x2_kernel(const u64* __restrict__ ca, const u64* __restrict__ cb, u64* __restrict__ t, i64 n, int L, const PrimeConst* __restrict__ pcs) {
  const i64 ct = blockIdx.z;
  const int l = blockIdx.y;
  const PrimeConst pc = pcs[l];
  const i64 rs = (i64)L * n;
  const u64* a0 = ca + ((ct * 2 + 0) * L + l) * n;
  const u64* a1 = a0 + rs;
  const u64* b0 = cb + ((ct * 2 + 0) * L + l) * n;
  const u64* b1 = b0 + rs;
  u64* t0 = t + ((ct * 3 + 0) * L + l) * n;
  u64* t1 = t0 + rs;
  u64* t2 = t1 + rs;
  const i64 n2 = n >> 1;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (i64)gridDim.x * blockDim.x) {
    const u64x2 x0 = ((const u64x2*)a0)[i], x1 = ((const u64x2*)a1)[i], y0 = ((const u64x2*)b0)[i], y1 = ((const u64x2*)b1)[i];
    u64x2 r0, r1, r2;
    r0.x = d_mulmod(x0.x, y0.x, pc);  r0.y = d_mulmod(x0.y, y0.y, pc);
    r2.x = d_mulmod(x1.x, y1.x, pc);  r2.y = d_mulmod(x1.y, y1.y, pc);
    r1.x = d_addmod(d_mulmod(x0.x, y1.x, pc), d_mulmod(x1.x, y0.x, pc), pc.q);
    r1.y = d_addmod(d_mulmod(x0.y, y1.y, pc), d_mulmod(x1.y, y0.y, pc), pc.q);
    ((u64x2*)t0)[i] = r0;  ((u64x2*)t1)[i] = r1;  ((u64x2*)t2)[i] = r2;
  }
}

// out[ct][r][l][:] = sum_k key[r][k][l][:] * dig[ct][k][l][:]   (r = 0,1)
// One block column handles CT_TILE ciphertexts so that every key element is loaded once per CT_TILE uses; products are
// accumulated as exact 128-bit integers and reduced once at the end.  The digit values may be lazy representatives (anything
// below 4 q_tile + 2^32 < 2^63, which is what the fused digit transform stores): the launcher derives from that bound after how many
// columns the accumulators have to be folded (every 63 columns for 60-bit primes: once per sum at the metric config's 66 columns).
struct Acc128 { u64 lo, hi; };
__device__ __forceinline__ void acc_mad(Acc128& a, u64 x, u64 y) {
  const u128 s = ((u128)a.hi << 64 | a.lo) + (u128)x * y;      // one 64x64->128 multiply-add: 4 v_mad_u64_u32
  a.lo = (u64)s;
  a.hi = (u64)(s >> 64);
}
__device__ __forceinline__ u64 acc_reduce(const Acc128& a, const PrimeConst& pc) {
  const u64 q = pc.q;
  const u64 h = d_shoup(a.hi, 1, pc.one_sh, q);              // hi mod q
  const u64 t = d_shoup_lazy(h, pc.r64, pc.r64_sh, q);       // hi * 2^64 mod q, in [0,2q)
  const u64 l = d_shoup_lazy(a.lo, 1, pc.one_sh, q);         // lo mod q, in [0,2q)
  u64 r = t + l;                                              // < 4q
  if (r >= pc.two_q) r -= pc.two_q;
  if (r >= q) r -= q;
  return r;
}
// SUBORDER: the digit rows are in the sub-block order the head-fused transform of n = 2^15 leaves them in (evaluation 2j + sub at
// [sub][j]): the two evaluations a lane works on come from the two halves of the row instead of from adjacent words.
template <int CT_TILE, bool SUBORDER = false>
__global__ void __launch_bounds__(256) dot_accum_kernel(const u64* __restrict__ key, const u64* __restrict__ dig, int ncol, i64 n, int L, i64 count,
                                                         u64* __restrict__ out, const PrimeConst* __restrict__ pcs, int slot0, int fold_every) {
  // prime-major block order (blockIdx.z = prime): all ciphertext tiles of one prime run back to back, so that prime's 2*ncol key
  // rows (17 MiB at the metric config) are re-read from the Infinity Cache instead of HBM by every tile after the first
  const i64 ct0 = (i64)blockIdx.y * CT_TILE;
  const int l = blockIdx.z + slot0;
  const PrimeConst pc = pcs[l];
  const i64 rs = (i64)L * n;
  const u64* k0 = key + (i64)l * n;
  const u64* k1 = key + ((i64)ncol * L + l) * n;
  const i64 n2 = n >> 1;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (i64)gridDim.x * blockDim.x) {
    Acc128 acc[CT_TILE][4];
#pragma unroll
    for (int c = 0; c < CT_TILE; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[c][e] = Acc128{0, 0};
    int until_fold = fold_every;
    for (int k = 0; k < ncol; ++k) {
      if (fold_every && until_fold-- == 0) {        // uniform
        until_fold = fold_every - 1;
#pragma unroll
        for (int c = 0; c < CT_TILE; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[c][e] = Acc128{acc_reduce(acc[c][e], pc), 0};
      }
      const u64x2 a = ((const u64x2*)(k0 + k * rs))[i];
      const u64x2 b = ((const u64x2*)(k1 + k * rs))[i];
#pragma unroll
      for (int c = 0; c < CT_TILE; ++c) {
        if (ct0 + c < count) {
          u64x2 d;
          const u64* drow = dig + ((ct0 + c) * ncol + k) * rs + (i64)l * n;
          if (SUBORDER) { d.x = __builtin_nontemporal_load(&drow[i]); d.y = __builtin_nontemporal_load(&drow[n2 + i]); }
          else d = __builtin_nontemporal_load(&((const u64x2*)drow)[i]);   // digit rows are read once
          acc_mad(acc[c][0], a.x, d.x);
          acc_mad(acc[c][1], a.y, d.y);
          acc_mad(acc[c][2], b.x, d.x);
          acc_mad(acc[c][3], b.y, d.y);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CT_TILE; ++c) {
      if (ct0 + c < count) {
        u64x2 r0, r1;
        r0.x = acc_reduce(acc[c][0], pc);  r0.y = acc_reduce(acc[c][1], pc);
        r1.x = acc_reduce(acc[c][2], pc);  r1.y = acc_reduce(acc[c][3], pc);
        ((u64x2*)(out + (((ct0 + c) * 2 + 0) * L + l) * n))[i] = r0;
        ((u64x2*)(out + (((ct0 + c) * 2 + 1) * L + l) * n))[i] = r1;
      }
    }
  }
}

// new[idx(j)] = old[idx(j*k mod m)] for j in Z_m^*  (zms_list[i] = i-th element of Z_m^*)
__global__ void __launch_bounds__(256) automorph_kernel(u64* __restrict__ dst, const u64* __restrict__ src, i64 phim, i64 m, i64 k,
                                                         const int* __restrict__ zms_idx, const int* __restrict__ zms_list) {
  const i64 row = blockIdx.y;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < phim; i += (i64)gridDim.x * blockDim.x) {
    const i64 j = zms_list[i];
    const i64 jk = (i64)(((unsigned __int128)j * (u64)k) % (u64)m);
    dst[row * phim + i] = src[row * phim + zms_idx[jk]];
  }
}

__global__ void __launch_bounds__(256) rows_equal_kernel(const u64* __restrict__ a, const u64* __restrict__ b, i64 nwords, int* __restrict__ differ) {
  int d = 0;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (i64)gridDim.x * blockDim.x) d |= (a[i] != b[i]);
  if (d) atomicOr(differ, 1);
}

static unsigned grid_x_for(i64 work_items) {
  i64 b = (work_items + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 64 ? 64 : b));
}

int launch_ew_op(fhesi_ctx* ctx, u64* d_dst, const u64* d_src, i64 count, int nslots, const int* d_prime_of_slot, int op) {
  const i64 n = ctx->phim, nrows = count * nslots;
  if (!nrows) return 0;
  ProfScope prof(ctx, PROF_EW, (double)nrows);
  dim3 grid(grid_x_for(n / 2), (unsigned)nrows);
  switch (op) {
    case 0: ew_op_kernel<0><<<grid, 256, 0, ctx->stream>>>(d_dst, d_src, n, nslots, d_prime_of_slot, ctx->d_pc); break;
    case 1: ew_op_kernel<1><<<grid, 256, 0, ctx->stream>>>(d_dst, d_src, n, nslots, d_prime_of_slot, ctx->d_pc); break;
    case 2: ew_op_kernel<2><<<grid, 256, 0, ctx->stream>>>(d_dst, d_src, n, nslots, d_prime_of_slot, ctx->d_pc); break;
    default: FHESI_FAIL("DoubleCRT::Op: unknown operation %d", op);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_ew_scalar(fhesi_ctx* ctx, u64* d_dst, const u64* d_scalars, i64 count, int nslots, const int* d_prime_of_slot, int op) {
  const i64 n = ctx->phim, nrows = count * nslots;
  if (!nrows) return 0;
  dim3 grid(grid_x_for(n), (unsigned)nrows);
  switch (op) {
    case 0: ew_scalar_kernel<0><<<grid, 256, 0, ctx->stream>>>(d_dst, d_scalars, n, nslots, d_prime_of_slot, ctx->d_pc); break;
    case 1: ew_scalar_kernel<1><<<grid, 256, 0, ctx->stream>>>(d_dst, d_scalars, n, nslots, d_prime_of_slot, ctx->d_pc); break;
    case 2: case 3: ew_scalar_kernel<2><<<grid, 256, 0, ctx->stream>>>(d_dst, d_scalars, n, nslots, d_prime_of_slot, ctx->d_pc); break;
    case 4: ew_scalar_kernel<FHESI_OP_SET_><<<grid, 256, 0, ctx->stream>>>(d_dst, d_scalars, n, nslots, d_prime_of_slot, ctx->d_pc); break;
    default: FHESI_FAIL("DoubleCRT scalar op: unknown operation %d", op);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_ew_exp(fhesi_ctx* ctx, u64* d_dst, const u64* d_exps, i64 count, int nslots, const int* d_prime_of_slot, unsigned* d_zero_flag) {
  const i64 n = ctx->phim, nrows = count * nslots;
  if (!nrows) return 0;
  dim3 grid(grid_x_for(n), (unsigned)nrows);
  if (d_zero_flag) ew_exp_kernel<true><<<grid, 256, 0, ctx->stream>>>(d_dst, d_exps, n, nslots, d_prime_of_slot, ctx->d_pc, d_zero_flag);
  else ew_exp_kernel<false><<<grid, 256, 0, ctx->stream>>>(d_dst, d_exps, n, nslots, d_prime_of_slot, ctx->d_pc, nullptr);
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_tensor2x2(fhesi_ctx* ctx, const u64* d_a, const u64* d_b, u64* d_t, i64 count) {
  if (!count) return 0;
  if (ctx->phim & 1) FHESI_FAIL("tensor2x2: odd phi(m) not supported by the batched pipeline");
  ProfScope prof(ctx, PROF_TENSOR, (double)count);
  dim3 grid(grid_x_for(ctx->phim / 2), (unsigned)ctx->L, (unsigned)count);
  PROF_KERNEL(ctx, PROF_TENSOR, tensor2x2_kernel);
  tensor2x2_kernel<<<grid, 256, 0, ctx->stream>>>(d_a, d_b, d_t, ctx->phim, ctx->L, ctx->d_pc);
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_dot_accum(fhesi_ctx* ctx, const u64* d_key, const u64* d_dig, int ncol, i64 count, u64* d_out, int slot0, int nslot, bool dig_suborder) {
  if (!count) return 0;
  if (nslot <= 0) { slot0 = 0; nslot = ctx->L; }
  if (ctx->phim & 1) FHESI_FAIL("dot_accum: odd phi(m) not supported by the batched pipeline");
  // exact 128-bit accumulation: F columns of (digit < 4 q_tile + 2^32) * (key < q) on top of a folded value below q must stay below 2^128
  int fold_every = 0;
  for (int l = 0; l < ctx->L; ++l) {
    const u128 term = (u128)(4 * ctx->pc[l].q_tile + ((u64)1 << 32)) * ctx->pc[l].q;
    const u128 F = (~(u128)0 - ctx->pc[l].q) / term;
    if (F < 2) FHESI_FAIL("dot_accum: residues of prime %d overflow the 128-bit accumulator", l);
    if (F < (u128)ncol && (fold_every == 0 || (int)F < fold_every)) fold_every = (int)F;
  }
  ProfScope prof(ctx, PROF_DOT, (double)count);
  constexpr int CT_TILE = 2;     // with prime-major blocks the key rows come from the Infinity Cache: 2 measured best on MI355X (1: +20 %, 4: +8 %, 8: +22 % time)
  const i64 ntiles = (count + CT_TILE - 1) / CT_TILE;
  if (ntiles > 65535) FHESI_FAIL("dot_accum: more than %d ciphertexts per call", 65535 * CT_TILE);
  dim3 grid(grid_x_for(ctx->phim / 2), (unsigned)ntiles, (unsigned)nslot);
  if (dig_suborder) PROF_KERNEL(ctx, PROF_DOT, dot_accum_kernel<CT_TILE, true>); else PROF_KERNEL(ctx, PROF_DOT, dot_accum_kernel<CT_TILE>);
  if (dig_suborder) dot_accum_kernel<CT_TILE, true><<<grid, 256, 0, ctx->stream>>>(d_key, d_dig, ncol, ctx->phim, ctx->L, count, d_out, ctx->d_pc, slot0, fold_every);
  else dot_accum_kernel<CT_TILE><<<grid, 256, 0, ctx->stream>>>(d_key, d_dig, ncol, ctx->phim, ctx->L, count, d_out, ctx->d_pc, slot0, fold_every);
  HIP_TRY(hipGetLastError());
  return 0;
}

// SingleCRT += / -= ZZ (SingleCRT.cpp:137-153 with NTL::add / NTL::sub on a ZZX and a scalar): only the constant coefficient changes
__global__ void scrt_const_kernel(u64* __restrict__ rows, i64 n, int nslots, const u64* __restrict__ scalars, const int* __restrict__ prime_of_slot, const PrimeConst* __restrict__ pcs, int op) {
  const int s = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (s >= nslots) return;
  const u64 q = pcs[prime_of_slot ? prime_of_slot[s] : s].q;
  u64* x = rows + (i64)s * n;
  x[0] = op == 0 ? d_addmod(x[0], scalars[s], q) : d_submod(x[0], scalars[s], q);
}
int launch_scrt_const(fhesi_ctx* ctx, u64* d_rows, const u64* d_scalars, int nslots, const int* d_prime_of_slot, int op) {
  scrt_const_kernel<<<(unsigned)((nslots + 63) / 64), 64, 0, ctx->stream>>>(d_rows, ctx->phim, nslots, d_scalars, d_prime_of_slot, ctx->d_pc, op);
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_automorph(fhesi_ctx* ctx, u64* d_dst, const u64* d_src, i64 nrows, i64 k) {
  if (!nrows) return 0;
  dim3 grid(grid_x_for(ctx->phim), (unsigned)nrows);
  automorph_kernel<<<grid, 256, 0, ctx->stream>>>(d_dst, d_src, ctx->phim, ctx->m, k, ctx->d_zms_idx, ctx->d_zms_list);
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_rows_equal(fhesi_ctx* ctx, const u64* a, const u64* b, i64 nwords, int* equal) {
  int* d_flag;
  FHESI_TRY(ws_reserve(ctx, 4, 64, (void**)&d_flag));
  HIP_TRY(hipMemsetAsync(d_flag, 0, sizeof(int), ctx->stream));
  if (nwords) rows_equal_kernel<<<grid_x_for(nwords) * 4, 256, 0, ctx->stream>>>(a, b, nwords, d_flag);
  int h = 0;
  HIP_TRY(hipMemcpyAsync(&h, d_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  *equal = !h;
  return 0;
}

// ---- sum of tensor products over index lists: the `newMatrix(i,j) += tmp` loops of Matrix.cpp:62-72,157-167,243 with
// tmp = a *= b (Ciphertext.cpp:167-192) and += on scaled-up ciphertexts (:135-142), fused:
//   out[g][0..2] = sum_{t in [seg[g], seg[g+1])} (a0 b0, a0 b1 + a1 b0, a1 b1)   with a = ca[slot_a[t]], b = cb[slot_b[t]]
// ca: [nua][2][L][n] = evaluation form of the p-lifted left operands, cb: [nub][2][L][n] right operands; every distinct operand is
// transformed once however many products use it.  Exact 128-bit accumulation, folded every 32 terms (64 products of < 2^122).
__global__ void __launch_bounds__(256) tensor_sum_kernel(const u64* __restrict__ ca, const u64* __restrict__ cb, const int* __restrict__ slot_a,
                                                         const int* __restrict__ slot_b, const int* __restrict__ seg, int accumulate, u64* __restrict__ out,
                                                         i64 n, int L, const PrimeConst* __restrict__ pcs) {
  const int g = blockIdx.z, l = blockIdx.y;
  const PrimeConst pc = pcs[l];
  const i64 rs = (i64)L * n;
  u64* o0 = out + (((i64)g * 3 + 0) * L + l) * n;
  const int t0 = seg[g], t1 = seg[g + 1];
  for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x) {
    Acc128 r0{0, 0}, r1{0, 0}, r2{0, 0};
    if (accumulate) { r0.lo = o0[j]; r1.lo = o0[rs + j]; r2.lo = o0[2 * rs + j]; }
    for (int t = t0; t < t1; ++t) {
      const u64* a = ca + (((i64)slot_a[t] * 2) * L + l) * n + j;
      const u64* b = cb + (((i64)slot_b[t] * 2) * L + l) * n + j;
      const u64 a0 = a[0], a1 = a[rs], b0 = b[0], b1 = b[rs];
      acc_mad(r0, a0, b0);
      acc_mad(r1, a0, b1);
      acc_mad(r1, a1, b0);
      acc_mad(r2, a1, b1);
      if (((t - t0) & 31) == 31) { r0 = Acc128{acc_reduce(r0, pc), 0}; r1 = Acc128{acc_reduce(r1, pc), 0}; r2 = Acc128{acc_reduce(r2, pc), 0}; }
    }
    o0[j] = acc_reduce(r0, pc);
    o0[rs + j] = acc_reduce(r1, pc);
    o0[2 * rs + j] = acc_reduce(r2, pc);
  }
}
int launch_tensor_sum(fhesi_ctx* ctx, const u64* d_ca, const u64* d_cb, const int* d_slot_a, const int* d_slot_b, const int* d_seg, i64 ngroups, bool accumulate,
                      u64* d_out, double nproducts) {
  if (!ngroups) return 0;
  for (int l = 0; l < ctx->L; ++l) if (ctx->pc[l].bar_k > 61) FHESI_FAIL("tensor_sum: %u-bit residues overflow the 128-bit accumulator", ctx->pc[l].bar_k);
  ProfScope prof(ctx, PROF_TENSOR, nproducts);
  for (i64 done = 0; done < ngroups; done += 65535) {
    const i64 cnt = ngroups - done < 65535 ? ngroups - done : 65535;
    dim3 grid(grid_x_for(ctx->phim), (unsigned)ctx->L, (unsigned)cnt);
    tensor_sum_kernel<<<grid, 256, 0, ctx->stream>>>(d_ca, d_cb, d_slot_a, d_slot_b, d_seg + done, accumulate ? 1 : 0, d_out + done * 3 * ctx->L * ctx->phim, ctx->phim, ctx->L,
                                                     ctx->d_pc);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}
