/* fhesi_hip.h -- C ABI of the MI355X (gfx950) DoubleCRT backend for fhe-si.
 *
 * The reference (dwu4/fhe-si) has no FFI: its hot path is the C++ class graph
 *   Ciphertext / KeySwitchSI  ->  DoubleCRT methods  ->  Cmodulus::FFT / iFFT  ->  BluesteinFFT  ->  NTL fftRep.
 * This header is the seam a maintainer binds the *bodies* of those methods to (INTEGRATION.md shows the
 * binding); every entry point names the reference interface it replaces (file:line under /root/reference).
 *
 * Conventions
 *   - plain C, no C++/torch/NTL types; all functions return 0 on success, non-zero on error
 *     (fhesi_last_error() gives the message).  The reference aborts through NTL Error()/assert
 *     (e.g. DoubleCRT.cpp:83,316,443; FHEContext.cpp:34): the C++ mirror turns a non-zero status
 *     into the same abort-style Error(msg).  No exception crosses this boundary.
 *   - residue rows: uint64_t[phi(m)], canonical values in [0,q_i), Z_m^* ascending order
 *     (the reference's vec_long rows, CModulus.cpp:103-106).
 *   - big integers: little-endian 64-bit limbs, two's complement, fixed nlimbs per call,
 *     coefficient-major [ncoeffs][nlimbs] (the reference's ZZX coefficients).
 *   - "host" pointers are ordinary memory; "_dev" entry points take device (HBM) pointers that
 *     must belong to the context's device.  Work is enqueued on the context's HIP stream;
 *     functions that return data to the host synchronise that stream.
 *   - one host thread per context; contexts are independent (one per GPU in the multi-GPU model).
 */
#ifndef FHESI_HIP_H_
#define FHESI_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fhesi_ctx fhesi_ctx;     /* FHEcontext + vector<Cmodulus> + PAlgebra (FHEContext.h, CModulus.h, PAlgebra.h) */
typedef struct fhesi_dcrt fhesi_dcrt;   /* one DoubleCRT object (DoubleCRT.h:83-365), rows resident in HBM */
typedef struct fhesi_ksk fhesi_ksk;     /* one KeySwitchSI matrix (FHE-SI.cpp:206-208), resident in HBM */
typedef struct fhesi_comm fhesi_comm;   /* one rank of a multi-GPU group: an RCCL communicator (ncclComm_t) over xGMI */

enum { FHESI_OP_ADD = 0, FHESI_OP_SUB = 1, FHESI_OP_MUL = 2, FHESI_OP_DIV = 3, FHESI_OP_SET = 4 };

const char* fhesi_last_error(void);
int fhesi_device_count(int32_t* count);
/* ABI revision of this header.  It changes whenever an existing entry point changes its parameters (revision 5:
 * fhesi_keyswitch_init_batch_seeded took its public_seed argument in round 4; revision 6 adds this query and
 * fhesi_host_stage_release).  A binding compiled or written against another revision must refuse to run: the Python binding
 * (fhe-si_amd/binding.py) and the C++ mirror (fhe-si_amd/host/fhesi_context.h) compare FHESI_ABI_VERSION with the library's
 * answer when they load it -- a stale ctypes table or mirror would otherwise link and silently shift arguments. */
#define FHESI_ABI_VERSION 7
int32_t fhesi_abi_version(void);

/* ---- context: FHEcontext::AddPrime (FHEContext.cpp:30-43) + Cmod::privateInit (CModulus.cpp:60-86) +
 * PAlgebra::init (PAlgebra.cpp:40-56).  Tables (twiddles / Bluestein powers and Rb) are built eagerly and
 * uploaded once (the reference fills them lazily: bluestein.cpp:103,121).  Rejects q not prime, q != 1 mod 2m,
 * duplicate q (FHEContext.cpp:34), m outside [2, 2^20] (FHEContext.cpp:89), or root not a primitive 2m-th root.
 * root[i] must be supplied (the reference draws a random one when 0, NumbTh.cpp:101-115; the C++ mirror does
 * that search on the host and passes the result here so row values are reproducible). */
int fhesi_ctx_create(fhesi_ctx** out, int64_t m, int32_t nprimes, const uint64_t* q, const uint64_t* root, int32_t device);
int fhesi_ctx_destroy(fhesi_ctx* ctx);
int64_t fhesi_ctx_m(const fhesi_ctx* ctx);
int64_t fhesi_ctx_phim(const fhesi_ctx* ctx);               /* PAlgebra::phiM (PAlgebra.h:78) */
int32_t fhesi_ctx_nprimes(const fhesi_ctx* ctx);            /* FHEcontext::numPrimes (FHEContext.h:149) */
int fhesi_ctx_prime(const fhesi_ctx* ctx, int32_t i, uint64_t* q, uint64_t* root);   /* ithPrime / getRoot */
int fhesi_ctx_zms_idx(const fhesi_ctx* ctx, int32_t* out_m);                        /* PAlgebra::indexInZmstar table */
int fhesi_ctx_phi_m(const fhesi_ctx* ctx, int64_t* out_phim_plus_1);                /* PAlgebra::PhimX coefficients */
int fhesi_ctx_sync(fhesi_ctx* ctx);
void* fhesi_ctx_stream(fhesi_ctx* ctx);                     /* the hipStream_t all work of this context runs on */
/* Behaviour switches of one context (A/B measurements, test hooks).  Names: "ks_direct" (1 = per-chain-prime key-switch dot product, the
 * reference's own structure FHE-SI.cpp:251-254), "ks_residues", "ks_aux60", "crt_exact", "crt_skip_cleanup" (test hook), "lanes" (2 = two
 * concurrent half-batches), "stagger", "batch_chunk", "wave_operands", "tensor32" (0 = the fused pipeline keeps the tensor product on the
 * chain primes), "tensor_bits" (30 / 29: the size of the tensor half's primes -- below 2^29 the row transforms skip half of their range steps
 * for one or two primes more; the integers formed are the same); layout switches that never change a result (round 5): "dot32_k4" (0 = the digit-tile dot product dot32_kernel2 where the
 * key-in-LDS form dot32_kernel4 would run), "parts_words" (0 = 64-bit limb rows between the tensor half and the digit loader),
 * "ks_long_keys", "host_chunk", "host_threads".
 * The FHESI_<NAME> environment variables give the initial values, read once in fhesi_ctx_create -- never per call.  FHESI_LIN_LG (also read
 * there; a test hook) asks for LONGER zero-padded rows than a linear-convolution ring needs (15 .. 20: the fused loaders of rows of 2^15 / 2^16 and
 * the paths of rings with safe primes beyond 65 537, on rings small enough for an oracle).  FHESI_WS_POISON=1 (also a test hook) fills every
 * workspace reservation with 0xA5 bytes before use: results must not depend on what a workspace held.  FHESI_PHI_CONV=1 (test hook) makes
 * the reduction modulo Phi_m of a generic m (not a power of two, a prime or twice a prime) run as two convolutions at any size; above
 * m = 16384 that is the only form.
 * fhesi_ctx_destroy fails while DoubleCRT / key-switch handles of the context are alive (they hold the reference's `const FHEcontext&`). */
int fhesi_ctx_set_option(fhesi_ctx* ctx, const char* name, int64_t value);
int fhesi_ctx_get_option(const fhesi_ctx* ctx, const char* name, int64_t* value);
int fhesi_ctx_copy_options(fhesi_ctx* dst, const fhesi_ctx* src);      /* every switch of src onto dst: the per-GPU replicas of one FHEcontext run the same forms */
/* HIP-event stopwatch on the context's stream (bench.py's per-kernel timing) */
int fhesi_timer_start(fhesi_ctx* ctx);
int fhesi_timer_stop(fhesi_ctx* ctx, float* elapsed_ms);

/* per-kernel-class stopwatch: while enabled every launch of a class is bracketed by a HIP-event pair on the context's
 * stream; read returns (#launches, units processed -- rows for the NTT classes, polynomials/ciphertexts otherwise --, total ms) */
enum { FHESI_PROF_NTT_FWD = 0, FHESI_PROF_NTT_INV = 1, FHESI_PROF_RNS = 2, FHESI_PROF_TENSOR = 3, FHESI_PROF_CRT = 4,
       FHESI_PROF_DIGITS = 5, FHESI_PROF_DOT = 6, FHESI_PROF_EW = 7,
       FHESI_PROF_NTT_FWD_DIGITS_MAIN = 8 /* the fused ByteDecomp + forward-NTT tile kernel alone; units = rows it transformed */ };
int fhesi_prof_enable(fhesi_ctx* ctx, int32_t on);      /* also clears the records */
int fhesi_prof_read(fhesi_ctx* ctx, int32_t kernel_class, int64_t* launches, double* units, double* total_ms);
/* demangled name (as rocprofv3 prints it, without the argument list) of the kernel the most recent profiled launch of that class ran:
 * the roofline line of bench.py names kernels by what the library launched, not by string literals */
int fhesi_prof_kernel_name(fhesi_ctx* ctx, int32_t kernel_class, char* name_out, size_t name_cap);

/* ---- Cmodulus::FFT / iFFT, one row (CModulus.h:165-166; CModulus.cpp:90-107, 110-132) */
int fhesi_cmod_fft(fhesi_ctx* ctx, int32_t prime, const uint64_t* coeff_limbs, int32_t nlimbs, int64_t ncoeffs, uint64_t* y_out);
int fhesi_cmod_ifft(fhesi_ctx* ctx, int32_t prime, const uint64_t* y, uint64_t* x_out /* phi(m) values in [0,q) */);

/* ---- DoubleCRT objects (DoubleCRT.h:83-365).  prime_idx = ascending index set; nidx==0 means ctxtPrimes = all
 * (DoubleCRT.cpp:244-250).  A new object holds the zero polynomial (DoubleCRT.cpp:261-311). */
int fhesi_dcrt_alloc(fhesi_ctx* ctx, const int32_t* prime_idx, int32_t nidx, fhesi_dcrt** out);
int fhesi_dcrt_free(fhesi_dcrt* d);
int fhesi_dcrt_copy(fhesi_dcrt* dst, const fhesi_dcrt* src);        /* operator=(DoubleCRT) DoubleCRT.cpp:313-320: error if contexts differ */
int fhesi_dcrt_index_set(const fhesi_dcrt* d, int32_t* idx_out, int32_t* nidx);   /* getIndexSet (DoubleCRT.h:303) */
int fhesi_dcrt_equal(const fhesi_dcrt* a, const fhesi_dcrt* b, int32_t* equal);  /* operator== (DoubleCRT.h:167-169) */
int fhesi_dcrt_upload_row(fhesi_dcrt* d, int32_t prime, const uint64_t* row);      /* setMap / Import (Serialization.cpp:56-81) */
int fhesi_dcrt_download_row(const fhesi_dcrt* d, int32_t prime, uint64_t* row);    /* getMap / Export */
void* fhesi_dcrt_device_ptr(fhesi_dcrt* d);                                          /* [nidx][phi(m)] uint64 in HBM */
int fhesi_dcrt_from_poly(fhesi_dcrt* d, const uint64_t* coeff_limbs, int32_t nlimbs, int64_t ncoeffs);   /* DoubleCRT(const ZZX&) / operator=(ZZX): DoubleCRT.cpp:212-257,323-331 */
int fhesi_dcrt_to_poly(const fhesi_dcrt* d, const int32_t* prime_idx, int32_t nidx, int32_t positive,
                       uint64_t* coeff_limbs_out, int32_t nlimbs);                   /* toPoly: DoubleCRT.cpp:349-404 (+ intVecCRT NumbTh.cpp:307-335) */
int fhesi_dcrt_op(fhesi_dcrt* dst, const fhesi_dcrt* src, int32_t op);              /* Op(DoubleCRT): DoubleCRT.cpp:79-113 (equal index sets; ADD/SUB/MUL) */
int fhesi_dcrt_op_scalar(fhesi_dcrt* d, const uint64_t* num_limbs, int32_t nlimbs, int32_t op);
                                                                                     /* Op(ZZ) :115-129, operator/= :407-420, operator=(ZZ) :333-347 */
int fhesi_dcrt_exp(fhesi_dcrt* d, int64_t e);                                       /* Exp: DoubleCRT.cpp:423-434 (PowerMod per element; e < 0 needs every element invertible) */
int fhesi_dcrt_automorph(fhesi_dcrt* d, int64_t k);                                 /* automorph: DoubleCRT.cpp:439-465; error if k not in Zm* */
int fhesi_dcrt_add_primes(fhesi_dcrt* d, const int32_t* prime_idx, int32_t nidx);   /* addPrimes: DoubleCRT.cpp:142-156 */
int fhesi_dcrt_remove_primes(fhesi_dcrt* d, const int32_t* prime_idx, int32_t nidx);/* removePrimes: DoubleCRT.h:197-199 */
/* BGV-style modulus switching (no caller inside fhe-si, part of the DoubleCRT surface); p = FHEcontext::ModulusP().
 * addPrimesAndScale (DoubleCRT.cpp:162-208): existing rows *= F (F^-1 mod p), F = product of the added primes; the added rows are zero;
 *   *log_factor_out (may be null) = log F + log(F^-1 mod p), the method's return value.  Errors: sets not disjoint, F not invertible mod p.
 * scaleDownToSet (DoubleCRT.cpp:518-558): drop the primes outside `prime_idx`, dividing by their product D with the correction term
 *   delta (D (D^-1 mod p) - 1) reduced modulo D p.  Errors mirror the asserts of :525-526 (empty intersection, nothing to drop). */
int fhesi_dcrt_add_primes_and_scale(fhesi_dcrt* d, const int32_t* prime_idx, int32_t nidx, uint64_t p, double* log_factor_out);
int fhesi_dcrt_scale_down_to_set(fhesi_dcrt* d, const int32_t* prime_idx, int32_t nidx, uint64_t p);
/* SingleCRT <-> DoubleCRT (DoubleCRT.cpp:484-515): coefficient-domain residues per prime, [nidx][phi(m)] */
int fhesi_dcrt_from_scrt(fhesi_dcrt* d, const uint64_t* coeff_rows);
int fhesi_dcrt_to_scrt(const fhesi_dcrt* d, uint64_t* coeff_rows_out);

/* ---- SingleCRT (SingleCRT.h:41-175, SingleCRT.cpp): the coefficient-domain RNS form -- for every prime of the index set the
 * polynomial's phi(m) coefficients modulo that prime.  It shares the DoubleCRT handle type and row storage; a handle made by
 * fhesi_scrt_alloc holds coefficient residues, and calls of the other family on it fail.  Shared entry points: fhesi_dcrt_free,
 * fhesi_dcrt_copy (operator=, SingleCRT.cpp:219-228), fhesi_dcrt_index_set, fhesi_dcrt_equal (SingleCRT.h:98-100),
 * fhesi_dcrt_upload_row / download_row (getMap), fhesi_dcrt_op with ADD / SUB (Op(SingleCRT, AddMod / SubMod), SingleCRT.cpp:61-103),
 * fhesi_dcrt_remove_primes (SingleCRT.h:117-119). */
int fhesi_scrt_alloc(fhesi_ctx* ctx, const int32_t* prime_idx, int32_t nidx, fhesi_dcrt** out);     /* SingleCRT(context, s): zero polynomial (SingleCRT.cpp:188-203) */
int fhesi_scrt_from_poly(fhesi_dcrt* s, const uint64_t* coeff_limbs, int32_t nlimbs, int64_t ncoeffs);   /* operator=(ZZX): PolyRed(poly, p_i, abs=true) per prime
                                                                                        (SingleCRT.cpp:239-251, NumbTh.cpp:210-233); ncoeffs <= phi(m) */
int fhesi_scrt_to_poly(const fhesi_dcrt* s, const int32_t* prime_idx, int32_t nidx, uint64_t* coeff_limbs_out, int32_t nlimbs);
                                                                                     /* toPoly (SingleCRT.cpp:299-334): centred CRT over (index set & s) */
int fhesi_scrt_op_scalar(fhesi_dcrt* s, const uint64_t* num_limbs, int32_t nlimbs, int32_t op);
                                                                                     /* Op(ZZ, add / sub / mul) (SingleCRT.cpp:137-153) and operator/= (:279-296):
                                                                                        ADD / SUB change the constant coefficient only (NTL add(ZZX, ZZX, ZZ)) */
int fhesi_dcrt_assign_scrt(fhesi_dcrt* d, const fhesi_dcrt* s);                      /* DoubleCRT::operator=(SingleCRT): DoubleCRT.cpp:484-496, in HBM */
int fhesi_scrt_assign_dcrt(fhesi_dcrt* s, const fhesi_dcrt* d, const int32_t* prime_idx, int32_t nidx);
                                                                                     /* DoubleCRT::toSingleCRT(scrt, s) / SingleCRT::operator=(DoubleCRT):
                                                                                        DoubleCRT.cpp:498-515, SingleCRT.cpp:231-235; nidx = 0 and null = all */

/* ---- batched device-resident row kernels.  rows_dev: [count][L][phi(m)] uint64 in HBM, all L primes */
int fhesi_rows_ntt_fwd_dev(fhesi_ctx* ctx, uint64_t* rows_dev, int64_t count);      /* coefficient residues -> evaluations, in place (Cmod::FFT after conv) */
int fhesi_rows_ntt_inv_dev(fhesi_ctx* ctx, uint64_t* rows_dev, int64_t count);      /* evaluations -> coefficient residues, in place (Cmod::iFFT) */
int fhesi_rows_op_dev(fhesi_ctx* ctx, uint64_t* dst_dev, const uint64_t* src_dev, int64_t count, int32_t op);   /* DoubleCRT::Op over a batch */

/* ---- key-switch matrix (KeySwitchSI::keySwitchMatrix, FHE-SI.cpp:206-208): [2][ncomp*ndigits][L][phi(m)],
 * matrix[0]=b, matrix[1]=A, column i*ndigits+j (FHE-SI.cpp:176-177) */
int fhesi_ksk_create(fhesi_ctx* ctx, int32_t ncomp, int32_t ndigits, fhesi_ksk** out);
int fhesi_ksk_free(fhesi_ksk* k);
int fhesi_ksk_upload(fhesi_ksk* k, const uint64_t* rows_host);                       /* whole matrix, host layout as above */
void* fhesi_ksk_device_ptr(fhesi_ksk* k);                                            /* HBM buffer of the rows (pure getter) */
int fhesi_ksk_mark_dirty(fhesi_ksk* k);                                              /* call after writing the rows through the pointer (a collective that
                                                                                        received into them): the library keeps tables derived from the rows
                                                                                        and rebuilds them at the next key switch */
int fhesi_ksk_upload_dev(fhesi_ksk* k, const uint64_t* rows_dev);                    /* whole matrix from another HBM buffer (copies and invalidates) */
size_t fhesi_ksk_bytes(const fhesi_ksk* k);
/* Which form of KeySwitchSI::ApplyKeySwitch's dot product (Util.h:79-98 at FHE-SI.cpp:251-254) the last key switch with this matrix ran:
 * form 0 = one dot product per chain prime (the reference's own structure), 1 = exact integer dot product over four 30-bit auxiliary
 * primes in limb mode, 2 = over the two largest chain primes in limb mode, 3 = over them in residue mode; rows = limbs (or residues) per
 * key coefficient, limb_bits = their width (0 in residue mode).  All forms give the reference's bits; a caller (bench, tests) reads this to
 * state which one it measured.  Before the first key switch: form -1. */
int fhesi_ksk_form(const fhesi_ksk* k, int32_t* form, int32_t* rows, int32_t* limb_bits);
/* Form 1 looks at the matrix it is given: KeySwitchSI::Init (FHE-SI.cpp:176-204) samples its polynomial modulo 2^logQ and reduces b modulo
 * 2^logQ, so the integer coefficients of a generated matrix lie in [-2^(logQ-1), 2^(logQ-1)] -- less than half the bits of the chain
 * product.  The library measures the coefficients when it builds its table; when they are that small it cuts the limbs from the CENTRED
 * integers (7 instead of 15 at the metric ring; `rows` above says how many) and the dot product needs no reduction modulo the chain product.
 * Any other matrix (uniform residues, say) takes the general limbs.  centred: 1 if the last table was built that way; key_bits: the measured
 * nb with every coefficient in [-2^nb, 2^nb] (0 if the table was not measured).  Option "ks_long_keys" = 1 forces the general limbs. */
int fhesi_ksk_key_bits(const fhesi_ksk* k, int32_t* centred, int32_t* key_bits);

int fhesi_selftest_aux32(fhesi_ctx* c);                                              /* diagnostic: checks the 32-bit auxiliary transforms of the key switch
                                                                                        (n = 2^14 only) as a ring isomorphism; 0 = ok */

/* ---- the metric's unit of work, batched: Ciphertext::operator*= (Ciphertext.cpp:167-192) followed by
 * KeySwitchSI::ApplyKeySwitch (FHE-SI.cpp:241-260) = ScaleDown (Ciphertext.cpp:194-218) + ByteDecomp (:82-121)
 * + DotProduct (Util.h:79-98) + toPoly + ReduceCoefficients (Util.cpp:3-33).
 * a, b: [count][2][phi(m)][nlimbs] two's complement coefficients mod 2^logQ (centered); out: same shape.
 * nlimbs must be >= ceil(logQ/64).  decomp_bytes = FHEcontext::decompSize (3 at every reference call site). */
int fhesi_ct_mul_relin_batch(fhesi_ctx* ctx, const fhesi_ksk* k, int32_t logQ, uint64_t p, int32_t decomp_bytes,
                             const uint64_t* a_host, const uint64_t* b_host, uint64_t* out_host, int32_t nlimbs, int64_t count);
/* The host-buffer form runs as a pipeline of stages over a pinned staging ring: upload of stage i + 1, compute of stage i and download of
 * stage i - 1 overlap on three streams, and pageable buffers are copied into / out of the ring by several threads (options "host_chunk":
 * ciphertexts per stage, "host_threads").  Buffers allocated with fhesi_host_alloc (pinned) are read and written by the DMA engines
 * directly, without the copy (only when the WHOLE batch lies in pinned / registered memory; a partly registered range is treated as
 * pageable).  Device or managed pointers are rejected with an error: this entry takes host memory, the _dev form device memory.
 * Same bits as the _dev form. */
int fhesi_host_alloc(fhesi_ctx* ctx, size_t bytes, void** out);                      /* pinned host memory for ciphertext batches */
int fhesi_host_free(fhesi_ctx* ctx, void* p);                                        /* ctx is not dereferenced (may be null or already destroyed) */
int fhesi_host_stage_release(fhesi_ctx* ctx);                                        /* frees the pinned + device staging ring kept between host-buffer calls */
int fhesi_ct_mul_relin_batch_dev(fhesi_ctx* ctx, const fhesi_ksk* k, int32_t logQ, uint64_t p, int32_t decomp_bytes,
                                 const uint64_t* a_dev, const uint64_t* b_dev, uint64_t* out_dev, int32_t nlimbs, int64_t count);
/* Separately callable stages of the same pipeline (parity tests check each against the oracle) */
int fhesi_ct_mul_dev(fhesi_ctx* ctx, uint64_t p, const uint64_t* a_dev, const uint64_t* b_dev, int32_t nlimbs, int64_t count,
                     uint64_t* tprod_dev /* [count][3][L][phi(m)] */);              /* Ciphertext::operator*= */
int fhesi_apply_key_switch_dev(fhesi_ctx* ctx, const fhesi_ksk* k, int32_t logQ, int32_t decomp_bytes,
                               const uint64_t* tprod_dev, int64_t count, uint64_t* out_dev, int32_t nlimbs);   /* ApplyKeySwitch */

/* ---- ciphertext algebra between multiplications, on batches resident in HBM: what Matrix<Ciphertext> (Matrix.cpp:57-98,
 * 150-174,182-263) and Regression::Regress / SumBatchedData (Regression.h:102-149,166-178) call on Ciphertext objects.
 * Unscaled ciphertexts: [count][nparts][phi(m)][nlimbs] two's complement; scaled-up ones (tProd): [count][3][L][phi(m)] rows. */
int fhesi_ct_add_dev(fhesi_ctx* ctx, int32_t logQ, uint64_t* dst_dev, const uint64_t* src_dev, int32_t nparts, int32_t nlimbs, int64_t count);
                                                                                     /* Ciphertext::operator+= unscaled: Ciphertext.cpp:123-134 (scaled-up: fhesi_rows_op_dev) */
int fhesi_ct_add_const_dev(fhesi_ctx* ctx, int32_t logQ, uint64_t p, uint64_t* ct_dev, int32_t nparts, int32_t nlimbs, int64_t count,
                           const int64_t* poly_host /* [npoly][phi(m)] */, int32_t npoly /* 1 = the same constant for every ciphertext, or count */);
                                                                                     /* Ciphertext::operator+=(const ZZX&) unscaled: Ciphertext.cpp:147-156 -- part 0 += (other << logQ) / p
                                                                                        (floor), ReduceCoefficients; the scaled-up branch (:157-159) is DoubleCRT += ZZX = fhesi_dcrt_from_poly + fhesi_dcrt_op */
int fhesi_ct_mul_poly_dev(fhesi_ctx* ctx, int32_t logQ, uint64_t* ct_dev, int32_t nparts, int32_t nlimbs, int64_t count,
                          const int64_t* poly_host /* [npoly][phi(m)] */, int32_t npoly);
                                                                                     /* Ciphertext::operator*=(const ZZX&) unscaled: Ciphertext.cpp:245-249 -> CiphertextPart::operator*=(ZZX) :29-36
                                                                                        (integer product, rem Phi_m, Reduce); scaled-up (:250-254): tProd[i] *= DoubleCRT(other) = fhesi_dcrt_op */
int fhesi_ct_mul_long_dev(fhesi_ctx* ctx, int32_t logQ, uint64_t* ct_dev, int64_t l, int32_t nparts, int32_t nlimbs, int64_t count);
                                                                                     /* Ciphertext::operator*=(long) unscaled: Ciphertext.cpp:232-237 -> :21-27 */
int fhesi_rows_mul_long_dev(fhesi_ctx* ctx, uint64_t* rows_dev, int64_t l, int64_t count);   /* ... scaled-up: Ciphertext.cpp:238-241 (DoubleCRT *= long); count DoubleCRTs */
int fhesi_ct_automorph_dev(fhesi_ctx* ctx, int64_t k, const uint64_t* in_dev, int32_t nparts, int32_t nlimbs_in, int64_t count,
                           uint64_t* out_dev, int32_t nlimbs_out);                   /* Ciphertext::operator>>= unscaled: Ciphertext.cpp:264-269 -> :54-59; the result is
                                                                                        centred modulo the prime chain, not modulo 2^logQ; error if k not in Zm* */
/* (ctxt >>= k) followed by KeySwitchSI::ApplyKeySwitch with the matrix of KeySwitchSI(secretKey, k) (FHE-SI.cpp:229-260): one
 * step of Regression::SumBatchedData (Regression.h:170-172).  k = 1 applies the key switch to the unscaled ciphertext as it is.
 * in: [count][ncomp][phi(m)][nlimbs_in], ncomp = the matrix's source components; out: [count][2][phi(m)][nlimbs] */
int fhesi_ct_automorph_key_switch_dev(fhesi_ctx* ctx, const fhesi_ksk* k_matrix, int32_t logQ, int32_t decomp_bytes, int64_t k,
                                      const uint64_t* in_dev, int32_t nlimbs_in, int64_t count, uint64_t* out_dev, int32_t nlimbs);
/* out[i] = pool[idx[i]] for elements of `words` uint64 each (operands of one wave of products) */
int fhesi_ct_gather_dev(fhesi_ctx* ctx, const uint64_t* pool_dev, const int32_t* idx_host, int64_t count, int64_t words, uint64_t* out_dev);
/* One wave of Matrix<Ciphertext> arithmetic: for every group g
 *     out[g] = ApplyKeySwitch( sum_{t in [seg[g], seg[g+1])}  pool[a_idx[t]] *= pool[b_idx[t]] )
 * i.e. the inner loops of Matrix::operator*= (Matrix.cpp:57-79), MultByTranspose (:150-174) and Determinant (:227-263) -- products of
 * unscaled ciphertexts (Ciphertext.cpp:167-192) summed while scaled up (:135-142) -- followed by the `reduce` / MapAll key switch
 * (Regression.h:112-115,127-134).  pool: unscaled 2-part ciphertexts [npool][2][phi(m)][nlimbs]; a_idx, b_idx, seg: host arrays. */
int fhesi_ct_mul_sum_relin_dev(fhesi_ctx* ctx, const fhesi_ksk* k, int32_t logQ, uint64_t p, int32_t decomp_bytes, const uint64_t* pool_dev,
                               int32_t nlimbs, const int32_t* a_idx, const int32_t* b_idx, const int32_t* seg, int64_t ngroups, uint64_t* out_dev);

/* ---- Encrypt / Decrypt in batches.  The polynomial arithmetic runs on the device; the randomness stays the
 * caller's (the reference draws it from NTL's PRNG, FHE-SI.cpp:14-25), which keeps results reproducible bit for bit.
 * FHESIPubKey::Encrypt (FHE-SI.cpp:10-36): pk0, pk1 = publicKey[0..1] over all primes;
 *   rand_host [count][3][phi(m)] int64 = (r: binary polynomial, e0, e1: Gaussian samples BEFORE the multiplication by p);
 *   msg_host [count][phi(m)] int64 in [0,p) (coefficient form);  out_dev [count][2][phi(m)][nlimbs] centred mod 2^logQ. */
int fhesi_encrypt_batch(fhesi_ctx* ctx, const fhesi_dcrt* pk0, const fhesi_dcrt* pk1, int32_t logQ, uint64_t p, const int64_t* rand_host,
                        const int64_t* msg_host, int64_t count, uint64_t* out_dev, int32_t nlimbs);
/* FHESISecKey::Decrypt (FHE-SI.cpp:93-119) of unscaled 2-part ciphertexts: sk1 = sKeys[1] (sKeys[0] = 1);
 *   msg_host [count][phi(m)] = round(p * (c0 + c1 t) / 2^logQ) mod p, floor((2 p z + q) / (2 q)) as in the reference. */
int fhesi_decrypt_batch(fhesi_ctx* ctx, const fhesi_dcrt* sk1, int32_t logQ, uint64_t p, const uint64_t* ct_dev, int32_t nlimbs, int64_t count,
                        int64_t* msg_host);

/* KeySwitchSI::Init (FHE-SI.cpp:153-209) for every column of a matrix in one call: k = the matrix (ncomp = components of the source
 * key: 3 for InitS2's (1, t, t^2), 2 for InitAutomorph), src[i] = the source key's DoubleCRT components, dst_t = dst[1].
 *   a_host   [ncomp*ndigits][phi(m)][nlimbs]  the SampleRandom polynomials (:174-175), centred modulo 2^logQ, column i*ndigits+j
 *   err_host [ncomp*ndigits][phi(m)] int64    the Gaussian errors (:189-190)
 * drawn by the caller in the reference's order (poly, err per column).  Result: k[1][col] = -DoubleCRT(a), k[0][col] =
 * DoubleCRT(Reduce(toPoly(a t) + err + (toPoly(s_i) << 8 decompSize j))), all transforms, products and conversions on the device. */
int fhesi_keyswitch_init_batch(fhesi_ksk* k, const fhesi_dcrt* const* src, int32_t nsrc, const fhesi_dcrt* dst_t, int32_t logQ, int32_t decomp_bytes,
                               const uint64_t* a_host, int32_t nlimbs, const int64_t* err_host);
int fhesi_ksk_download(const fhesi_ksk* k, uint64_t* rows_host);                         /* whole matrix to the host (Export, FHE-SI.cpp:270-272) */

/* ---- the same three with the randomness drawn ON THE DEVICE (sampleHWt / sampleGaussian NumbTh.cpp:340-404, the binary
 * and Gaussian polynomials of Encrypt FHE-SI.cpp:14-25, SampleRandom + sampleGaussian per key-switch column FHE-SI.cpp:174-190).  NTL's
 * sequential PRNG cannot be reproduced outside NTL, so these entry points use a counter-based generator instead -- Philox-4x32-10 keyed by
 * `seed`, counter = (coefficient, object index, purpose); definition in fhe-si_amd/csrc/philox.h, restated by the oracle and the Python
 * model -- which makes a ciphertext or a key a function of (seed, index) alone: the same on every GPU, in any batch split, and on the CPU.
 * The explicit-randomness forms above remain for callers that bring their own randomness (and for reproducing fixtures).
 *
 * WHAT THE CALLER OWES THESE ENTRY POINTS.  (1) Philox is a reproducible counter-based generator with a 64-bit key, NOT a CSPRNG: use it for
 * fixtures, tests and deployments whose threat model accepts that; key material that must withstand more takes the explicit-randomness
 * entry points with randomness from the caller's CSPRNG.  `seed` must in any case be secret, uniformly random 64 bits.  (2) An
 * (seed, object index) pair must never be used twice: two encryptions under the same pair share r, e0, e1, so their difference is
 * delta * (m1 - m2) in the clear; two key-switch columns under the same pair share a and the error.  There is no default index -- every call
 * names the first index of a range nobody else uses (the C++ mirror's SeedSequence hands out disjoint ranges from one counter shared by
 * Encrypt and every KeySwitchSI).  (3) The polynomials `a` of a key-switch matrix are public; they draw from `public_seed`, the errors from
 * `seed`: publishing public_seed (a matrix shipped as its seed) discloses nothing secret.  public_seed != seed. */
int fhesi_encrypt_batch_seeded(fhesi_ctx* ctx, const fhesi_dcrt* pk0, const fhesi_dcrt* pk1, int32_t logQ, uint64_t p, uint64_t seed, uint64_t first_index,
                               const int64_t* msg_host, int64_t count, uint64_t* out_dev, int32_t nlimbs);      /* plaintext i <-> object index first_index + i */
int fhesi_keyswitch_init_batch_seeded(fhesi_ksk* k, const fhesi_dcrt* const* src, int32_t nsrc, const fhesi_dcrt* dst_t, int32_t logQ, int32_t decomp_bytes,
                                      uint64_t seed, uint64_t public_seed, uint64_t first_index);                /* column c <-> object index first_index + c; a from public_seed, errors from seed */
int fhesi_dcrt_sample(fhesi_dcrt* d, int32_t kind, int64_t param, uint64_t seed, uint64_t index);               /* DoubleCRT::sampleHWt(param) (kind 0), ::sampleGaussian() with
                                                                                                                    stdev 3.2 (kind 1): DoubleCRT.h:340-345 */

/* ---- multi-GPU: independent ciphertexts are data-parallel, every GPU holds the context tables and a replica of
 * the key-switch matrices; RCCL collectives run on the context's stream.  librccl is loaded on first use (no RCCL needed on one GPU).
 * fhesi_comm_init_all: one process, one host thread per GPU -- ncclCommInitAll over `devices`, comms_out[r] is rank r's handle.
 *   (If `devices` repeats a GPU -- RCCL refuses that -- the group is a "loopback" group that moves the same bytes with device copies
 *   and host barriers between the calling threads: for exercising N > 1 host logic on a single-GPU box, never for measurements.)
 * fhesi_comm_from_rccl: wraps a caller-owned ncclComm_t (process-per-GPU launchers: ncclCommInitRank); not destroyed by us.
 * Collective calls must be made by every rank of the group (from its own thread or process), like the RCCL calls they are. */
int fhesi_comm_init_all(int32_t ndev, const int32_t* devices, fhesi_comm** comms_out);
int fhesi_comm_from_rccl(void* nccl_comm, fhesi_comm** out);
int fhesi_comm_destroy(fhesi_comm* comm);
int32_t fhesi_comm_rank(const fhesi_comm* comm);
int32_t fhesi_comm_size(const fhesi_comm* comm);
/* the set-up collective of the data-parallel model: rank `root`'s key-switch matrix (KeySwitchSI::keySwitchMatrix, FHE-SI.cpp:206-208;
 * 297 MiB at the metric ring) into every rank's replica `k` (same shape on every rank); derived tables are rebuilt on the receivers */
int fhesi_ksk_broadcast(fhesi_ksk* k, fhesi_comm* comm, int32_t root);
int fhesi_comm_broadcast_dev(fhesi_ctx* ctx, fhesi_comm* comm, void* buf_dev, size_t bytes, int32_t root);   /* any HBM buffer (bytes % 8 == 0) */
/* exchange of sharded wave outputs (Matrix<Ciphertext> waves, Matrix.cpp:150-174): rank r produced the words
 * [offsets_words[r], offsets_words[r+1]) of base_dev; afterwards every rank holds all of them (one grouped broadcast per producer) */
int fhesi_comm_exchange(fhesi_ctx* ctx, fhesi_comm* comm, uint64_t* base_dev, const int64_t* offsets_words);
/* the same exchange split for overlap with compute (the waves of Regression::Regress, Regression.h:102-149 over Matrix.cpp:182-263: a
 * wave's outputs are read by the NEXT wave only).  _begin enqueues the exchange on the communicator's own stream behind everything already
 * on the context's stream and returns without waiting for the GPU -- the context's stream goes on with the next chunk of the wave;
 * _end returns when every exchange begun since the last _end has landed.  All ranks call _begin for the same exchanges in the same order. */
int fhesi_comm_exchange_begin(fhesi_ctx* ctx, fhesi_comm* comm, uint64_t* base_dev, const int64_t* offsets_words);
int fhesi_comm_exchange_end(fhesi_ctx* ctx, fhesi_comm* comm);
/* exact all-reduce of partial scaled-up sums held by the ranks (Ciphertext::operator+= on scaled-up ciphertexts is linear,
 * Ciphertext.cpp:135-142): rows_dev [count][L][phi(m)] <- (sum over ranks) mod q_i; at most 16 ranks (64-bit partial sums of 60-bit residues) */
int fhesi_comm_allreduce_rows(fhesi_ctx* ctx, fhesi_comm* comm, uint64_t* rows_dev, int64_t count);

/* plain device-memory helpers so C callers need no HIP headers */
int fhesi_dev_alloc(fhesi_ctx* ctx, size_t bytes, void** out_dev);
int fhesi_dev_free(fhesi_ctx* ctx, void* dev);
int fhesi_dev_upload(fhesi_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int fhesi_dev_download(fhesi_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int fhesi_dev_copy(fhesi_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes);    /* e.g. RCCL receive buffer -> key matrix */

#ifdef __cplusplus
}
#endif
#endif /* FHESI_HIP_H_ */
