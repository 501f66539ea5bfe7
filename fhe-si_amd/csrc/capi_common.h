// capi_common.h -- what the translation units of the C-ABI layer (capi_*.hip, the implementation of include/fhesi_hip.h) share.
// Product path only: nothing here touches oracle/ and there is no CPU fallback -- every entry point that computes fails with an
// error string if HIP is unavailable.
//   capi_ctx.hip       context, options, stopwatch, plain device memory            capi_dcrt.hip   Cmodulus, DoubleCRT, SingleCRT, row batches
//   capi_pipeline.hip  key-switch matrix, the fused multiplication + key switch     capi_ct.hip     ciphertext algebra, Encrypt / Decrypt, key generation
#pragma once
#include "../../include/fhesi_hip.h"
#include "fhesi_internal.h"
#include <map>
#include <set>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <cxxabi.h>
#include <string>

#define CHECK_CTX(c) do { if (!(c)) FHESI_FAIL("null context"); HIP_TRY(hipSetDevice((c)->device)); } while (0)

static inline int row_fwd(fhesi_ctx* ctx, u64* d_rows, i64 count, int nslots, const int* d_pos, const int* h_pos) {
  if (ctx->pow2) return launch_ntt_fwd(ctx, d_rows, count, nslots, d_pos, true);
  return launch_bluestein_fwd(ctx, d_rows, count, nslots, h_pos);
}
static inline int row_inv(fhesi_ctx* ctx, u64* d_rows, i64 count, int nslots, const int* d_pos, const int* h_pos) {
  if (ctx->pow2) return launch_ntt_inv(ctx, d_rows, count, nslots, d_pos, true);
  return launch_bluestein_inv(ctx, d_rows, count, nslots, h_pos);
}
static inline std::vector<int> full_set(const fhesi_ctx* c) { std::vector<int> v(c->L); for (int i = 0; i < c->L; ++i) v[i] = i; return v; }
// device copy of an index list (small, cached per call in workspace slot 3); identity lists need none   (capi_ctx.hip)
int upload_idx(fhesi_ctx* ctx, const std::vector<int>& idx, int** d_out);
