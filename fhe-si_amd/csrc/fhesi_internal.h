// fhesi_internal.h -- shared declarations of the gfx950 DoubleCRT backend (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <atomic>
#include <vector>

typedef uint64_t u64;
typedef int64_t i64;
typedef unsigned int u32;
typedef unsigned __int128 u128;

// --------------------------------------------------------------------------------- error plumbing
void fhesi_set_error(const char* fmt, ...);
#define FHESI_FAIL(...) do { fhesi_set_error(__VA_ARGS__); return 1; } while (0)
#define HIP_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { fhesi_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); return 1; } } while (0)
#define FHESI_OP_SET_ 4
#define FHESI_WS_SLOTS 13
#define FHESI_TRY(expr) do { int r__ = (expr); if (r__) return r__; } while (0)

// --------------------------------------------------------------------------------- per-prime device constants
// One entry per chain prime (role of the reference's Cmodulus object, CModulus.h:42-61).
struct PrimeConst {
  u64 q;            // the modulus (CModulus.h:44)
  u64 two_q;
  u64 bar_mu;       // Barrett: floor(2^(2k)/q), k = bit length of q
  u32 bar_k;        // bit length of q
  u32 norm_m;       // floor(2^(31+b) / ((q_tile >> 32) + 1)), b = bit length of q_tile >> 32: quotient estimate of the tile kernels' store (modarith63.h)
  u64 one_sh;       // floor(2^64/q): Shoup quotient of the constant 1 (reduces any 64-bit word)
  u64 ninv, ninv_sh;        // phi(m)^-1 mod q and its Shoup quotient (power-of-two m: the /m of CModulus.cpp:125 folded with X^n=-1)
  u64 ninv_w, ninv_w_sh;    // ninv * psi^-brv(1) (last inverse stage twiddle folded with the scaling)
  u64 r64, r64_sh;          // 2^64 mod q (Horner step of the big-int -> residue reduction, CModulus.cpp:96 conv)
  u64 one_q63;              // floor(2^63/q_tile)
  u64 ninv_q63, ninv_w_q63; // floor(ninv 2^63/q_tile), floor(ninv_w 2^63/q_tile): quotients in the tile kernels' 63-bit convention (modarith63.h)
  u64 q_tile;               // modulus the tile kernels compute with: q, or the largest multiple of q below 2^60 when q < 2^48 (ntt_tile.inc)
};

struct Shoup2 { u64 w, wp; };   // constant multiplier and floor(w*2^64/q)

// CRT tables for one ordered prime subset (DoubleCRT::toPoly, DoubleCRT.cpp:349-398 / NumbTh.cpp:307-335)
struct CrtTables {
  int nidx = 0, W = 0;                 // W = limbs of the product of the subset (+1 for sign)
  std::vector<int> idx;
  u64* d_blob = nullptr;               // device copy of everything below
  // device pointers into d_blob
  int* d_idx = nullptr;                // [nidx]
  Shoup2* d_pow64 = nullptr;           // [nidx][W]   2^(64 j) mod q_k (+Shoup quotient)
  Shoup2* d_pinv = nullptr;            // [nidx]      (q_0...q_{k-1})^-1 mod q_k
  u64* d_P = nullptr;                  // [nidx+1][W] partial products P_k = q_0...q_{k-1}; row nidx = full product
  u64* d_halfP = nullptr;              // [W]         (P-1)/2
  // tables of the sum form x = sum_i y_i M_i - kappa P (crt_sum_kernel): M_i = P / q_i, y_i = r_i c_i mod q_i, c_i = M_i^-1 mod q_i
  u64* d_M = nullptr;                  // [nidx][W]
  u64* d_cinv = nullptr;               // [nidx][3]   c_i, floor(c_i 2^128 / q_i) as (hi, lo)
  mutable unsigned char* d_flags = nullptr;   // per-workgroup "recompute exactly" flags of the last launch (grow-only)
  mutable size_t flags_cap = 0;
};

struct BluesteinTables;                // general-m path, defined in bluestein.hip
struct fhesi_aux32;                    // four 30-bit auxiliary primes of the key switch (kernels_aux32.hip), built on first use

// per-kernel-class HIP-event stopwatch (bench.py's live kernel timing; off by default)
enum { PROF_NTT_FWD = 0, PROF_NTT_INV = 1, PROF_RNS = 2, PROF_TENSOR = 3, PROF_CRT = 4, PROF_DIGITS = 5, PROF_DOT = 6, PROF_EW = 7, PROF_NTT_FWD_DIGITS_MAIN = 8, PROF_NCLASS = 9 };
struct ProfRec { int cls; double units; hipEvent_t e0, e1; };

// Behaviour switches of one context (fhesi_ctx_set_option).  The FHESI_* environment variables of the same meaning are read ONCE, when
// the context is created, as initial values -- never per call.
struct CtxOptions {
  long long host_chunk = 0;  // ciphertexts per stage of the host-buffer pipeline of fhesi_ct_mul_relin_batch (0 = derived from the batch)
  int host_threads = 0;     // threads that copy between the caller's pageable buffers and the pinned ring (0 = min(8, hardware threads): 8 measured best, profiles/r04_host_buffers.txt)
  int ks_long_keys = 0;     // 1: limbs cut from the key coefficient in [0, P) whatever its size (the general form; A/B and checker of the centred limbs)
  int ks_direct = 0;        // 1: per-chain-prime key-switch dot product (the reference's structure) instead of the auxiliary-prime path
  int ks_residues = 0;      // 1: auxiliary-prime key switch in residue mode (no limb mode)
  int ks_aux60 = 0;         // 1: two 60-bit auxiliary primes even where the four 30-bit primes apply
  int crt_exact = 0;        // 1: mixed-radix CRT kernel instead of the sum form
  int crt_skip_cleanup = 0; // test hook: skip the exact clean-up pass of the sum-form CRT
  int lanes = 1;            // 2: two concurrent half-batches on two streams in fhesi_ct_mul_relin_batch_dev
  int stagger = 0;          // lanes = 2: start the second lane after the first lane's digit transform
  long long batch_chunk = 0;      // ciphertexts per pipeline chunk (0 = derived from the ring)
  long long wave_operands = 0;    // distinct operands per pass of fhesi_ct_mul_sum_relin_dev (0 = about 4 GiB of rows)
  int wave_single = 1;      // 1: a wave whose groups are single products takes the batch pipeline of fhesi_ct_mul_relin_batch_dev on gathered operands (0: the sum kernels, the checker)
  int automorph_rows = 0;   // 1: Ciphertext >>= through DoubleCRT::automorph on evaluation rows (the reference's structure) even where the coefficient gather applies
  int tensor32 = 1;         // 1: the fused pipeline's tensor half runs over 30-bit primes where that path applies (fhesi_ct_mul_relin_batch_dev)
  int tensor_bits = 30;     // 30: the tensor half's primes are the largest below 2^30; 29: below 2^29 -- lazy values have room up to 8p, so the row transforms skip 8 of 14 (forward) / 6 of 13 (inverse) range steps, for one or two primes more (36 instead of 35 at the metric ring)
  int dot32_k4 = 1;         // 1: key switch with 7 or 8 limbs and at least 24 ciphertexts per call runs dot32_kernel4 (keys in LDS, digits and accumulators in registers); 0: dot32_kernel2 (A/B)
  int parts_words = 1;      // 1: inside the fused multiplication the scaled-down parts travel as 32-bit word rows (crt32_scale -> digit loader); 0 = 64-bit limb rows (A/B)
};

struct fhesi_ctx {
  CtxOptions opt;
  int live_handles = 0;                // DoubleCRT objects and key-switch matrices created on this context and not yet freed
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  i64 m = 0, phim = 0;
  int L = 0;
  bool pow2 = false;
  int n_big_primes = 0;                // chain primes >= 2^48
  bool has_small_prime = false;        // some chain prime is below 2^48: the tile kernels transform its rows modulo q_tile (ntt_tile.inc)
  int logn = 0;                        // log2(phim) when pow2
  // m = 2 q' with q' an odd prime (the reference's safe-prime rings, e.g. p = 8423) or m an odd prime, and 2 phi(m) - 1 <= 2^15: the
  // integer products of the key switch and of the tensor half are LINEAR convolutions carried by the 2^14- or 2^15-point 32-bit
  // transforms on zero-padded rows and folded afterwards, exactly, modulo X^q' + 1 and Phi_m = sum (-X)^i  (m = 2q':  out_j =
  // S_j - S_(j+q') - (-1)^j S_(q'-1))  or modulo X^m - 1 and Phi_m = sum X^i  (m prime:  out_j = S_j + S_(j+m) - S_(m-1))
  i64 lin_q = 0;                       // the fold's offset: q' (m = 2q') or m (m prime); 0 = not such a ring
  bool lin_prime = false;              // m itself is the prime
  int lin_lg = 0;                      // log2 of the padded rows: 14 .. 20
  std::vector<u64> q, root;
  std::vector<int> zms_idx;            // PAlgebra::zmsIdx (PAlgebra.cpp:50-52)
  std::vector<i64> phi;                // Phi_m(X) (PAlgebra.cpp:55)
  std::vector<PrimeConst> pc;
  PrimeConst* d_pc = nullptr;          // [L]
  Shoup2* d_tw_fwd = nullptr;          // [L][phim]  psi^brv(i)     (pow2)
  Shoup2* d_tw_inv = nullptr;          // [L][phim]  psi^-brv(i)    (pow2)
  Shoup2* d_twt_fwd = nullptr;         // [L][phim]  same values in the tile kernel's permuted order (ntt_tile.inc), logn 11..14
  Shoup2* d_twt_inv = nullptr;
  // two-pass transforms, logn 15..17 (s0 = logn-14): d_twt_fwd = [L][2^14] table of the ring with root psi^(2^s0),
  // d_twt_inv = [L][2^s0][2^14] per-sub-transform slices, d_tail_fwd = [L][phim] twiddles of ntt_fwd_tail,
  // d_sub_fold = [L][2^s0] {1/n * inverse twiddle of stage s0, 63-bit quotient}
  Shoup2* d_twt_fwd_sub = nullptr;     // [L][2^s0][2^14] forward twiddle slices of the order-free sub-transforms (convolutions)
  Shoup2* d_head_tw = nullptr;         // [L] {psi^brv(1), 63-bit quotient}: the head stage fused into the digit loader at n = 2^15
  Shoup2* d_tail_fwd = nullptr;
  Shoup2* d_sub_fold = nullptr;
  int* d_zms_idx = nullptr;            // [m]
  int* d_zms_list = nullptr;           // [phim] ascending elements of Z_m^*
  BluesteinTables* blue = nullptr;
  fhesi_aux32* aux32 = nullptr;
  struct fhesi_tensor32* tensor32 = nullptr;     // tables of the 30-bit tensor half (kernels_tensor32.hip), built on first use
  std::map<std::vector<long long>, std::pair<int, std::vector<uint32_t>>> t32_memo;      // t32_plan's answers: (lift, limbs, logQ, bits of the group size, option) -> (2 NP + generic, primes)
  std::map<std::vector<int>, CrtTables*> crt_cache;
  std::map<int, Shoup2*> pow64_cache;  // nlimbs -> device [L][nlimbs+1] table for rns_reduce
  std::map<std::vector<u64>, u64*> scalar_cache;   // rns_reduce lift scalars (per-slot residues), keyed by the scalar list
  // second "lane" (stream + workspace): the batched ciphertext pipeline runs two half-batches concurrently so that one
  // half's HBM-bound kernels (dot, CRT loads) overlap the other half's VALU-bound NTTs
  hipStream_t lane_stream = nullptr;
  void* lane_ws[FHESI_WS_SLOTS] = {};
  size_t lane_ws_bytes[FHESI_WS_SLOTS] = {};
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_mid = nullptr;
  bool ws_poison = false;              // FHESI_WS_POISON=1: ws_reserve fills what it hands out with 0xA5 (test hook)
  bool ws_oom = false;                 // the last failing ws_reserve failed in hipMalloc (mul_relin_chunks then retries with smaller chunks)
  // operands of the fused multiplication addressed through pool indices instead of two contiguous batches (single-product waves of
  // fhesi_ct_mul_sum_relin_dev): device array [2][op_idx_n] (a's, then b's) of ciphertext slots in the buffer passed as `a`; null = contiguous
  const int* op_idx = nullptr;
  long long op_idx_n = 0, op_idx_done = 0;
  bool mark_mid = false;               // record ev_mid right after the next digit-NTT launch (staggers the second lane)
  struct HostStage* host_stage = nullptr;      // pinned staging ring + copy streams + copy threads of the host-buffer entry points (capi_pipeline.hip), built on first use
  bool prof_on = false;
  std::vector<ProfRec> prof;
  const void* prof_fn[16] = {};        // host stub of the kernel the last profiled launch of each class ran (fhesi_prof_kernel_name)
  // grow-only workspace.  Slot owners (a slot may be reused by another owner only when the first one's data is dead):
  //   0 digit rows / product operands / encrypt rows    1 inverse-transform scratch (tProd copy, dot output, automorph rows)
  //   2 limb-major parts / small-coefficient staging    3 automorph source rows / encrypt public key
  //   4 wave sums, per-call constants                    5 tProd of a chunk / message staging
  //   6 row transforms above 2^14 (two-pass, bit reversal)   7 Bluestein slot map / wave operands
  //   8 Bluestein convolution buffer, index lists        9 Bluestein inverse output, scalar lists (no Bluestein call in between)
  //   10 auxiliary-prime dot product output (kernels_ksaux.hip)      11 staging of host batches (fhesi_ct_mul_relin_batch)
  void* ws[FHESI_WS_SLOTS] = {};
  size_t ws_bytes[FHESI_WS_SLOTS] = {};
};

struct fhesi_dcrt {
  fhesi_ctx* ctx = nullptr;
  std::vector<int> idx;                // ascending prime indices (IndexSet, IndexSet.h)
  u64* d_rows = nullptr;               // [idx.size()][phim]
  bool coeff_form = false;             // the rows hold COEFFICIENT residues: the object is a SingleCRT (SingleCRT.h:41-175)
};

struct fhesi_ksk {
  fhesi_ctx* ctx = nullptr;
  int ncomp = 0, ndigits = 0;
  u64* d_rows = nullptr;               // [2][ncomp*ndigits][L][phim]
  size_t bytes = 0;
  // derived table of the two-auxiliary-prime dot product (kernels_ksaux.hip), rebuilt on the device when the rows changed
  u64* d_aux = nullptr;                // [2 aux][aux_rows][n/64][2][ncomp*ndigits][64] split 60-bit words, or (aux32) [4][aux_rows][n/64][2][ncol][64] u32
  size_t aux_bytes = 0;                // allocated size of d_aux
  u64* d_aux_consts = nullptr;         // [L] q_0 q_1 mod q_i, then the int pair {0, 1} (prime_of_slot of auxiliary rows)
  bool aux_valid = false, aux_suborder = false;
  int aux_mode = 0;                    // KS_MODE_* the table was built for: rebuilt when the options select another form
  int last_form = -1;                  // KS_MODE_* of the last key switch with this matrix (fhesi_ksk_form)
  // limb mode (kernels_ksaux.hip): the table is built from the key polynomial's INTEGER coefficients (toPoly over the chain) cut into
  // aux_rows limbs of aux_limb_bits bits instead of from its aux_rows = L chain-prime residues; 0 = residue mode
  int aux_rows = 0, aux_limb_bits = 0, aux_logQ = 0;
  // centred limbs (kernels_aux32.hip): every integer coefficient of the matrix lies in [-2^nb, 2^nb] with nb far below the chain product --
  // what KeySwitchSI::Init produces (FHE-SI.cpp:176-204: the polynomial is sampled modulo 2^logQ, b is reduced modulo 2^logQ) -- so the limbs
  // are cut from the CENTRED integer (top limb signed): ceil(nb / B) of them instead of ceil(log2 P / B), 7 instead of 15 at the metric ring,
  // and the dot product as an integer is below P / 2 by itself: no reduction modulo the chain product is left, only modulo 2^logQ
  bool aux_centred = false;
  int aux_key_bits = 0;                // nb of the matrix the table was built from (measured on the device at build time)
  int aux_long_opt = 0;                // option ks_long_keys at build time (a change rebuilds the table)
  bool aux32 = false;                  // the table holds residues modulo the four 30-bit primes of kernels_aux32.hip (u32, 2^14-point rows)
  i64 aux_fold = 0;                    // q' when the rows are linear convolutions to be folded modulo X^q' + 1 and Phi_m (ctx->lin_q), -m for a prime m (modulo X^m - 1 and Phi_m), else 0
  u64* d_limb_consts = nullptr;        // [W+1] offset constant D, [2] floor(2^(64(W-2)+128) / P), then the quotient bound's bit count
};
struct KsLimbPlan { int W = 0, LQ = 0, B = 0, NLB = 0, mbits = 0; bool a32 = false; };
bool ks_limb_plan(const fhesi_ctx* ctx, const CrtTables* t, int ncol, int digit_bits, int logQ, KsLimbPlan* plan, const u32* p32 /* the 30-bit primes, or null */);
int launch_ks_recombine(fhesi_ctx* ctx, const CrtTables* t, const fhesi_ksk* k, const u64* d_o /* [npolys][aux_rows][2][n] */, i64 npolys, u64* d_out, int nl_out, bool tail_pending = false);
bool ks_recombine_takes_tail(const fhesi_ctx* ctx, const CrtTables* t, const fhesi_ksk* k);
int aux32_tail_consts(fhesi_ctx* ctx, uint32_t (*tw)[2], uint32_t (*twp)[2]);      // per auxiliary prime: (1/2, psi^-brv(1)/2) and their quotients
void aux32_free(fhesi_ctx* ctx);
// the tensor half of the fused multiplication over primes below 2^30 (kernels_tensor32.hip)
void tensor32_free(fhesi_ctx* ctx);
void host_stage_free(fhesi_ctx* ctx);            // capi_pipeline.hip
bool tensor32_applies(const fhesi_ctx* ctx, u64 p, int nlimbs, int logQ);
// parts_wm: the scaled-down parts leave as 32-bit WORD rows [count*3][2 ceil(logQ/64)][n] (same bytes per polynomial, word-major) -- the layout the
// 32-bit digit loader (launch_ntt32_fwd_digits, wm) reads in whole cache lines; false: 64-bit limb rows, what every other consumer takes
int launch_tensor32(fhesi_ctx* ctx, u64 p, const u64* d_a, const u64* d_b, int nlimbs, int logQ, i64 count, u64* d_parts /* [count*3][logQ/64][n] */, bool parts_wm = false);
// ... and for sums of products per group (fhesi_ct_mul_sum_relin_dev): begin fixes the configuration for at most gmax terms per group
bool tensor32_sum_applies(const fhesi_ctx* ctx, u64 p, int nlimbs, int logQ, i64 gmax);
int tensor32_sum_begin(fhesi_ctx* ctx, u64 p, int nlimbs, int logQ, i64 gmax);
size_t tensor32_sum_bytes(const fhesi_ctx* ctx, i64 ngroups);
int tensor32_sum_pass(fhesi_ctx* ctx, const u64* d_ops, i64 nua, i64 nub, const int* d_slot_a, const int* d_slot_b, const int* d_seg, i64 ng, i64 nterms, bool accumulate, void* d_sum);
int tensor32_sum_finish(fhesi_ctx* ctx, void* d_sum, i64 ng, u64* d_parts, bool parts_wm = false);
const u32* aux32_primes(fhesi_ctx* ctx);          // the four primes (host array), nullptr on error
int launch_ntt32_fwd(fhesi_ctx* ctx, u32* d_rows, i64 count, int nslots, int a0);
int launch_ntt32_inv(fhesi_ctx* ctx, u32* d_rows, i64 count, int nslots, int a0, bool mont /* input scaled by 2^-32: dot32_kernel2 */, bool tail = true /* false: rows of 2^15 are left as their two sub-inverses */);
constexpr i64 kDigitSubCt = 64;                   // ciphertexts per sub-chunk of the tiled 32-bit digit rows (launch_ntt32_fwd_digits <-> launch_dot32)
int launch_ntt32_fwd_digits(fhesi_ctx* ctx, const u64* d_parts, int nl, int digit_bits, int nd, i64 npolys, u32* d_out /* tiled, see ntt32_core.inc */,
                            i64 sub_units /* units (digit polynomials) per sub-chunk */, bool wm = false /* d_parts as 32-bit word rows (launch_tensor32 parts_wm) */);
int ks32_build(fhesi_ctx* ctx, fhesi_ksk* k, const u64* d_kint, int W, int B, int NLB, void* d_tmp /* one prime's rows */, bool centred = false /* d_kint holds centred two's complement values: top limb signed */);
int ks32_key_bits(fhesi_ctx* ctx, const u64* d_kint /* [count][W] centred two's complement */, i64 count, int W, int* nbits /* smallest nb with every value in [-2^nb, 2^nb] */);
bool aux32_applies(const fhesi_ctx* ctx);          // n = 2^14 or 2^15, or a ring with lin_q set
i64 aux32_row_len(const fhesi_ctx* ctx);            // 2^15 for n = 2^15 and for linear-convolution rings with 2 phi(m) - 1 > 2^14, else 2^14
static const i64 kAux32N = 1 << 14;                // row length of the 32-bit auxiliary transforms
int launch_dot32(fhesi_ctx* ctx, fhesi_ksk* k, const u32* d_dig /* [count*ncol][4][n] */, int ncol, i64 count, u32* d_out /* [count*2*rows][4][n] */, bool* mont /* out: the rows carry 2^-32 */);
enum { KS_MODE_DIRECT = 0, KS_MODE_LIMB32 = 1 /* four 30-bit primes, limbs */, KS_MODE_LIMB60 = 2 /* two largest chain primes, limbs */, KS_MODE_RESIDUE60 = 3 /* ..., residues */ };
int ksaux_mode(fhesi_ctx* ctx, const CrtTables* t, int ncol, int digit_bits, int logQ);       // the form that runs for this chain, ring and option set
int ksaux_build(fhesi_ctx* ctx, fhesi_ksk* k, int digit_bits, int logQ, int mode);
int launch_dot_aux(fhesi_ctx* ctx, const fhesi_ksk* k, const u64* d_dig /* [count*ncol][2][n] */, int ncol, i64 count, u64* d_out /* [count][2][L][2][n] */);
int launch_aux_crt(fhesi_ctx* ctx, const fhesi_ksk* k, const u64* d_o /* [nrows][2][n] */, u64* d_dst /* [nrows][n] */, i64 nrows);

int ws_reserve(fhesi_ctx* ctx, int slot, size_t bytes, void** out);

// RAII scope: records an event pair around the launches issued while it is alive (only when profiling is on)
struct ProfScope {
  fhesi_ctx* c; int idx = -1;
  ProfScope(fhesi_ctx* ctx, int cls, double units) : c(ctx) {
    if (!c->prof_on) return;
    ProfRec r; r.cls = cls; r.units = units;
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
    hipEventRecord(r.e0, c->stream);
    c->prof.push_back(r); idx = (int)c->prof.size() - 1;
  }
  ~ProfScope() { if (idx >= 0) hipEventRecord(c->prof[idx].e1, c->stream); }
};

// names the kernel of the launch that follows (only recorded while profiling is on)
static inline void prof_kernel(fhesi_ctx* ctx, int cls, const void* host_stub) { if (ctx->prof_on) ctx->prof_fn[cls] = host_stub; }
#define PROF_KERNEL(ctx, cls, ...) prof_kernel((ctx), (cls), (const void*)&__VA_ARGS__)

// --------------------------------------------------------------------------------- host number theory (hostmath.cpp)
namespace hm {
u64 mulmod(u64 a, u64 b, u64 q);
u64 powmod(u64 a, u64 e, u64 q);
u64 invmod(u64 a, u64 q);              // q prime
bool is_prime(u64 n);
u64 shoup(u64 w, u64 q);
u64 shoup63(u64 w, u64 q);             // floor(w 2^63 / q)
u64 brv(u64 x, int bits);
int ilog2_ceil(i64 n);
std::vector<int> zms_idx(i64 m, i64* phim);
std::vector<i64> cyclotomic(i64 m);
std::vector<i64> cyclotomic_cofactor(i64 m);     // (X^m - 1) / Phi_m
bool is_primitive_2m_root(u64 root, i64 m, u64 q);
u64 bn_mod(const u64* limbs, int nlimbs, u64 q);          // signed two's complement -> [0,q)  (role of NTL rem(ZZ,long))
std::vector<u64> bn_mul_small(const std::vector<u64>& a, u64 b);   // non-negative
}  // namespace hm

// --------------------------------------------------------------------------------- kernel launchers
// kernels_ntt.hip : negacyclic NTT for power-of-two m.  rows: [count][nprimes_in_layout][n]; the prime of layout
// slot s is prime_of_slot[s] (nullptr = identity).  bitrev=false leaves the forward output / takes the inverse input in
// bit-reversed order (used by the Bluestein convolution engine).
int launch_ntt_fwd(fhesi_ctx* ctx, u64* d_rows, i64 count, int nslots, const int* d_prime_of_slot, bool bitrev = true);
int launch_ntt_inv(fhesi_ctx* ctx, u64* d_rows, i64 count, int nslots, const int* d_prime_of_slot, bool bitrev = true);
// order-free transforms of rows of 2^15..2^17 points in two stages each, for callers that fuse the cheap stage themselves
// (bluestein.hip): forward = head stages (ntt_fwd_head) + sub-transforms, inverse = sub-transforms + tail stages (ntt_inv_tail)
bool ntt_orderfree_two_pass(const fhesi_ctx* ctx);
int launch_ntt_fwd_head(fhesi_ctx* ctx, u64* d_rows, i64 count, int nslots, const int* d_prime_of_slot);
int launch_ntt_sub(fhesi_ctx* ctx, bool fwd, u64* d_rows, i64 count, int nslots, const int* d_prime_of_slot);
int launch_ntt_inv_tail_inplace(fhesi_ctx* ctx, u64* d_rows, i64 count, int nslots, const int* d_prime_of_slot);
// digit-row forward NTT: source is the scaled-down part in limb-major layout, see kernels_crt.hip
int launch_ntt_fwd_digits(fhesi_ctx* ctx, const u64* d_parts_limbmajor, int nl, int logQ, int digit_bits, int nd, i64 npolys,
                          u64* d_out_rows /* [npolys*nd][L][n] */, int slot0 = 0, int nslot = 0 /* 0 = all primes */, int layout_slots = 0 /* 0 = L */);

// kernels_ew.hip
int launch_ew_op(fhesi_ctx* ctx, u64* d_dst, const u64* d_src, i64 count, int nslots, const int* d_prime_of_slot, int op);
int launch_ew_exp(fhesi_ctx* ctx, u64* d_dst, const u64* d_exps /* [nslots] */, i64 count, int nslots, const int* d_prime_of_slot, unsigned* d_zero_flag /* non-null: only flag zero elements */);
int launch_ew_scalar(fhesi_ctx* ctx, u64* d_dst, const u64* d_scalars /* [nslots] residues */, i64 count, int nslots, const int* d_prime_of_slot, int op);
int launch_tensor2x2(fhesi_ctx* ctx, const u64* d_a /* [count][2][L][n] */, const u64* d_b /* [count][2][L][n] */, u64* d_t /* [count][3][L][n] */, i64 count);
int launch_dot_accum(fhesi_ctx* ctx, const u64* d_key /* [2][ncol][L][n] */, const u64* d_dig /* [count][ncol][L][n] */, int ncol, i64 count,
                     u64* d_out /* [count][2][L][n] */, int slot0 = 0, int nslot = 0 /* 0 = all primes */, bool dig_suborder = false);
// true when launch_ntt_fwd_digits leaves its rows in sub-block order (n = 2^15): evaluation 2j + sub at [sub][j]
bool ntt_digits_suborder(const fhesi_ctx* ctx, int digit_bits);
int launch_tensor_sum(fhesi_ctx* ctx, const u64* d_ca, const u64* d_cb, const int* d_slot_a, const int* d_slot_b, const int* d_seg, i64 ngroups, bool accumulate,
                      u64* d_out /* [ngroups][3][L][n] */, double nproducts);
int launch_scrt_const(fhesi_ctx* ctx, u64* d_rows, const u64* d_scalars /* [nslots] */, int nslots, const int* d_prime_of_slot, int op /* 0 add, 1 sub */);
int launch_automorph(fhesi_ctx* ctx, u64* d_dst, const u64* d_src, i64 nrows, i64 k);
int launch_rows_equal(fhesi_ctx* ctx, const u64* a, const u64* b, i64 nwords, int* equal);

// kernels_ct.hip : coefficient-domain ciphertext algebra on device batches
int launch_ct_add(fhesi_ctx* ctx, u64* d_dst, const u64* d_src, i64 ncoeffs, int nl, int logQ);
int launch_ct_mul_long(fhesi_ctx* ctx, u64* d_ct, i64 ncoeffs, int nl, int logQ, i64 l);
// kernels_sample.hip: counter-based randomness (philox.h) for Encrypt / key generation, drawn on the device
int launch_sample_encrypt(fhesi_ctx* ctx, i64* d_rnd /* [count][3][n] */, i64 count, u64 seed, u64 first);
int launch_sample_keygen(fhesi_ctx* ctx, u64* d_a, i64* d_err, i64 ncol, int nl, int logQ, u64 seed, u64 pub_seed, u64 first);
int launch_sample_poly(fhesi_ctx* ctx, i64* d_poly /* [n] */, int kind /* 0 sampleHWt(param), 1 sampleGaussian */, i64 param, u64 seed, u64 obj);
int launch_ct_add_const(fhesi_ctx* ctx, u64* d_ct /* [count][nparts][n][nl] */, const i64* d_poly /* [npoly][n] */, int npoly, int nparts, int nl, int logQ, u64 p, i64 count);
int launch_ct_automorph_parts(fhesi_ctx* ctx, const u64* d_in /* [npolys][n][nl_in] */, int nl_in, i64 npolys, i64 kk, int logQ, u64* d_parts /* [npolys][nlq][n] */, int nlq);   // 2: ring not covered
int launch_gather(fhesi_ctx* ctx, const u64* d_pool, const int* d_idx, i64 count, i64 words, u64* d_out);
int launch_segment_sum(fhesi_ctx* ctx, const u64* d_in, const int* d_seg, i64 ngroups, int ncomp, u64* d_out);
int launch_encrypt_combine(fhesi_ctx* ctx, const u64* d_rows /* [count][3][L][n] */, const u64* d_pk /* [2][L][n] */, i64 count, u64* d_out /* [count][2][L][n] */);
int launch_add_scaled_msg(fhesi_ctx* ctx, u64* d_ct, const i64* d_msg, const u64* d_delta, i64 count, int nl, int logQ);
int launch_decrypt_dot(fhesi_ctx* ctx, const u64* d_rows /* [count][2][L][n] */, const u64* d_t /* [L][n] */, i64 count, u64* d_out /* [count][L][n] */);
int launch_decrypt_round(fhesi_ctx* ctx, const u64* d_z, i64 total, int nw, int logQ, u64 p, i64* d_out);

int launch_rows_mul_bcast(fhesi_ctx* ctx, u64* d_dst /* [ncols][L][n] */, const u64* d_a /* [ncols][L][n] */, const u64* d_t /* [L][n] */, i64 ncols);
int launch_keygen_combine(fhesi_ctx* ctx, const u64* d_bcoef, int wb, const u64* d_scoef, int ws, const i64* d_err, i64 ncols, int nd, int digit_bits, int nl, int logQ, u64* d_out);

// kernels_crt.hip
int get_crt_tables(fhesi_ctx* ctx, const std::vector<int>& idx, CrtTables** out);
// big-int coefficients [count][npoly][n][nlimbs] -> residue rows [count][npoly][L][n]; scalar_mul[poly] (0 = none) multiplies
// the residue by (scalar mod q) -- the `parts[i].poly * p` lift of Ciphertext.cpp:171.
int launch_rns_reduce(fhesi_ctx* ctx, const u64* d_limbs, int nlimbs, i64 ncoeffs, i64 count, int npoly, const u64* scalar_mul,
                      u64* d_rows, int nslots, const int* d_prime_of_slot);
// CRT reconstruct rows [npolys][nslots_layout][n] over the subset in `t` -> W-limb coefficients.
// mode 0: centered / positive value as is (toPoly), out [count][npoly][n][nl_out] coefficient-major
// mode 1: ScaleDown: round-half-up(x / 2^logQ) mod 2^logQ, positive residue, limb-major [count][npoly][nl_out][n]
// mode 2: Reduce: centered mod 2^logQ, coefficient-major two's complement [count][npoly][n][nl_out]
// d_slot_of: device [K] layout slot of the k-th prime of the subset (nullptr: slot = prime index); npolys = number of polynomials.
int launch_crt(fhesi_ctx* ctx, const CrtTables* t, const u64* d_rows, int nslots_layout, const int* d_slot_of, i64 npolys, int mode, int positive,
               int logQ, u64* d_out, int nl_out);
int launch_modswitch_delta(fhesi_ctx* ctx, const u64* d_delta, int W, const u64* consts_host /* D, D p, floor(D p / 2): 3 W words */, u64 p, u64 u, u64* d_e);
// ByteDecomp: parts limb-major [npolys][nl][n] -> digit residue rows [npolys][nd][L][n]
int launch_digits(fhesi_ctx* ctx, const u64* d_parts, int nl, int logQ, int digit_bits, int nd, i64 npolys, u64* d_rows, u64 only_below_q = 0);

// bluestein.hip (general m)
int bluestein_init(fhesi_ctx* ctx);
void bluestein_destroy(fhesi_ctx* ctx);
int launch_bluestein_fwd(fhesi_ctx* ctx, u64* d_rows, i64 count, int nslots, const int* prime_of_slot_host);
int launch_bluestein_inv(fhesi_ctx* ctx, u64* d_rows, i64 count, int nslots, const int* prime_of_slot_host);

// --------------------------------------------------------------------------------- device arithmetic
#if defined(__HIPCC__)
// All residues are < q < 2^62 (NTL_SP_NBITS <= 60 in the reference, FHEContext.cpp:92), so 4q fits a word.
__device__ __forceinline__ u64 d_mulhi(u64 a, u64 b) { return __umul64hi(a, b); }

// Shoup / Harvey multiplication by a constant w with wp = floor(w 2^64 / q): returns y*w mod q in [0, 2q)
// for ANY 64-bit y (role of NTL MulModPrecon, DoubleCRT.cpp:195-197).
__device__ __forceinline__ u64 d_shoup_lazy(u64 y, u64 w, u64 wp, u64 q) {
  u64 Q = d_mulhi(y, wp);
  return y * w - Q * q;
}
__device__ __forceinline__ u64 d_shoup(u64 y, u64 w, u64 wp, u64 q) {
  u64 r = d_shoup_lazy(y, w, wp, q);
  return r >= q ? r - q : r;
}
__device__ __forceinline__ u64 d_addmod(u64 a, u64 b, u64 q) { u64 s = a + b; return s >= q ? s - q : s; }
__device__ __forceinline__ u64 d_submod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }

// Barrett reduction of x = (hi:lo) < 2^(2k) (k = bit length of q): result in [0,q)
__device__ __forceinline__ u64 d_barrett128(u64 hi, u64 lo, u64 q, u64 mu, u32 k) {
  // xs = floor(x / 2^(k-1))  (k+1 bits)
  u64 xs = (k == 1) ? lo : ((lo >> (k - 1)) | (hi << (65 - k)));
  if (k == 1) xs = lo;
  // qhat = floor(xs * mu / 2^(k+1))
  u64 ph = d_mulhi(xs, mu), pl = xs * mu;
  u64 qhat = (k + 1 >= 64) ? ph : ((pl >> (k + 1)) | (ph << (63 - k)));
  u64 r = lo - qhat * q;                   // true remainder in [0, 3q) -> fits since 3q < 2^64
  if (r >= q) r -= q;
  if (r >= q) r -= q;
  return r;
}
// generic a*b mod q for a,b in [0,q)  (role of NTL MulMod, DoubleCRT.cpp:110)
__device__ __forceinline__ u64 d_mulmod(u64 a, u64 b, const PrimeConst& pc) {
  u64 hi = d_mulhi(a, b), lo = a * b;
  return d_barrett128(hi, lo, pc.q, pc.bar_mu, pc.bar_k);
}
#endif
