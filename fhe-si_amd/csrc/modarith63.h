// modarith63.h -- the butterfly arithmetic of the tuned NTT kernels (gfx950 VALU, no MFMA).
//
// Multiplication by a constant w modulo q < 2^60 uses a Shoup quotient scaled by 2^63 instead of 2^64:
//   wq = floor(w * 2^63 / q) < 2^63,   Q = floor(y * wq / 2^63),   T = y*w - Q*q  in [0, q + y*q/2^63)
// so for y < 2^63 - 2^33 the result is below 2q.  With y1 < 2^31 and wq1 < 2^31 the middle sum
//   y1*wq0 + y0*wq1 + hi32(y0*wq0)
// cannot overflow 64 bits, which makes the whole quotient one carry-free chain of v_mad_u64_u32, and the exact floor is
//   Q = 2*y1*wq1 + (middle >> 31).
// The remainder is accumulated as y*w + Q*(-q) mod 2^64 (cross terms through a second carry-free chain whose low word is
// added to the high word), so there is no 64-bit subtraction and no carry pair anywhere.
// Lazy ranges: forward values stay below 4q + 2^32, inverse values below 2q + 2^47 (q < 2^60): the range correction only
// compares HIGH words against (2q)>>32 -- v >= 2q + 2^32 is always corrected, v < 2q never -- one v_cmp + two v_cndmask.
//
// hipcc rewrites such chains into v_mul_hi + zero-extension moves and pads every one-instruction asm with a wait
// state, so the hot butterflies are emitted as ONE asm block per PAIR of butterflies (two interleaved dependency chains)
// using fixed scratch VGPRs (clobbered), which gives access to register halves: 24 instructions / ~40 issue slots per
// forward butterfly instead of hipcc's 39 / ~60.  Hazards inside the block: the only VCC producer (v_cmp) is always
// followed by >= 2 unrelated VALU instructions before the v_cndmask that reads it; the carry-outs of v_mad_u64_u32 go to
// a dummy SGPR pair that nothing reads.
#pragma once
#include "fhesi_internal.h"

struct Tw63 { u32 w0, w1, p0, p1; };     // w = w1:w0,  floor(w 2^63 / q) = p1:p0   (same 16 bytes as Shoup2)
struct Mod63 {
  u32 nq0, nq1;        // -q mod 2^64
  u32 twoq_hi;         // (2q) >> 32
  u32 m2q_lo, m2q_hi;  // -2q mod 2^64
  u64 twoq1;           // 2q + 1   (x - T = x + ~T + 1)
  u64 threeq1;         // 3q + 1
  u64 q, twoq;
};

__device__ __forceinline__ Mod63 make_mod63(u64 q) {
  Mod63 m;
  const u64 nq = 0 - q, m2q = 0 - 2 * q;
  m.nq0 = (u32)nq; m.nq1 = (u32)(nq >> 32);
  m.twoq_hi = (u32)((2 * q) >> 32);
  m.m2q_lo = (u32)m2q; m.m2q_hi = (u32)(m2q >> 32);
  m.twoq1 = 2 * q + 1; m.threeq1 = 3 * q + 1;
  m.q = q; m.twoq = 2 * q;
  return m;
}

// ---- plain C++ version (compiler-scheduled): used outside the hot loops (last inverse stage, tests of the asm)
__device__ __forceinline__ u64 mad32(u32 a, u32 b, u64 c) { return (u64)a * b + c; }
__device__ __forceinline__ u64 mulmod63(u64 y, const Tw63& t, const Mod63& m) {
  const u32 y0 = (u32)y, y1 = (u32)(y >> 32);
  u64 M = (u64)__umulhi(y0, t.p0);
  M = mad32(y1, t.p0, M);
  M = mad32(y0, t.p1, M);
  const u64 Q = mad32(y1 << 1, t.p1, M >> 31);
  const u32 q0 = (u32)Q, q1 = (u32)(Q >> 32);
  u64 R = mad32(y0, t.w0, 0);
  R = mad32(q0, m.nq0, R);
  const u32 hi = (u32)(R >> 32) + y0 * t.w1 + y1 * t.w0 + q0 * m.nq1 + q1 * m.nq0;
  return (R & 0xffffffffull) | ((u64)hi << 32);
}

// ---- asm butterflies.  Scratch register sets (clobbered): A = v[104:115], B = v[116:127].
//  +0,+1 : M / Q      +2,+3 : C (cross terms)     +4,+5 : R = T      +6,+7 : D (inverse) / 2*y1      +8,+9 : sel, xc
#define FHESI_STR2(x) #x
#define FHESI_STR(x) FHESI_STR2(x)
#define VR_(n) "v" FHESI_STR(n)
#define VP_(a, b) "v[" FHESI_STR(a) ":" FHESI_STR(b) "]"

// quotient + remainder of (y1:y0) * w, result T in v[b+4 : b+5]; y0,y1,w0,w1,p0,p1 are operand strings
#define MULMOD63_ASM(b0, b1, b2, b3, b4, b5, b6, Y0, Y1, W0, W1, P0, P1)                      \
  "v_mul_hi_u32 " VR_(b0) ", " Y0 ", " P0 "\n\t"                                               \
  "v_mov_b32 " VR_(b1) ", 0\n\t"                                                               \
  "v_mad_u64_u32 " VP_(b0, b1) ", %[cy], " Y1 ", " P0 ", " VP_(b0, b1) "\n\t"                  \
  "v_mad_u64_u32 " VP_(b2, b3) ", %[cy], " Y0 ", " W1 ", 0\n\t"                                \
  "v_mad_u64_u32 " VP_(b0, b1) ", %[cy], " Y0 ", " P1 ", " VP_(b0, b1) "\n\t"                  \
  "v_lshlrev_b32 " VR_(b6) ", 1, " Y1 "\n\t"                                                   \
  "v_mad_u64_u32 " VP_(b2, b3) ", %[cy], " Y1 ", " W0 ", " VP_(b2, b3) "\n\t"                  \
  "v_lshrrev_b64 " VP_(b0, b1) ", 31, " VP_(b0, b1) "\n\t"                                     \
  "v_mad_u64_u32 " VP_(b4, b5) ", %[cy], " Y0 ", " W0 ", 0\n\t"                                \
  "v_mad_u64_u32 " VP_(b0, b1) ", %[cy], " VR_(b6) ", " P1 ", " VP_(b0, b1) "\n\t"             \
  "v_mad_u64_u32 " VP_(b2, b3) ", %[cy], " VR_(b0) ", %[nq1], " VP_(b2, b3) "\n\t"             \
  "v_mad_u64_u32 " VP_(b4, b5) ", %[cy], " VR_(b0) ", %[nq0], " VP_(b4, b5) "\n\t"             \
  "v_mad_u64_u32 " VP_(b2, b3) ", %[cy], " VR_(b1) ", %[nq0], " VP_(b2, b3) "\n\t"             \
  "v_add_u32 " VR_(b5) ", " VR_(b5) ", " VR_(b2) "\n\t"

#define BFLY_FWD_NAME bfly_fwd63_x2
#define BFLY_INV_NAME bfly_inv63_x2
#define BFLY_TWC "v"
#include "bfly63_body.inc"
#undef BFLY_FWD_NAME
#undef BFLY_INV_NAME
#undef BFLY_TWC
// wave-uniform twiddles held in SGPRs (phase A / A' of the tile kernels: the twiddle depends only on the register index)
#define BFLY_FWD_NAME bfly_fwd63_x2_s
#define BFLY_INV_NAME bfly_inv63_x2_s
#define BFLY_TWC "s"
#include "bfly63_body.inc"
#undef BFLY_FWD_NAME
#undef BFLY_INV_NAME
#undef BFLY_TWC

// ---- the first two forward stages of a DIGIT row (phase A, SGPR twiddles).  The loader hands over values below 2^32, so
//  stage 0: y1 = 0 -- the quotient chain loses three multiplies and the shift of y1 -- and X needs no range step
//           (X' = X + T < 2q + 2^32,  Y' = X + 2q - T < 2q + 2^32);
//  stage 1: inputs below 2q + 2^32 need no range step either (outputs below 4q + 2^32, the invariant of the later stages).
#define MULMOD63_SMALL_ASM(b0, b1, b2, b3, b4, b5, Y0, W0, W1, P0, P1)                         \
  "v_mul_hi_u32 " VR_(b0) ", " Y0 ", " P0 "\n\t"                                               \
  "v_mov_b32 " VR_(b1) ", 0\n\t"                                                               \
  "v_mad_u64_u32 " VP_(b2, b3) ", %[cy], " Y0 ", " W1 ", 0\n\t"                                \
  "v_mad_u64_u32 " VP_(b0, b1) ", %[cy], " Y0 ", " P1 ", " VP_(b0, b1) "\n\t"                  \
  "v_mad_u64_u32 " VP_(b4, b5) ", %[cy], " Y0 ", " W0 ", 0\n\t"                                \
  "v_lshrrev_b64 " VP_(b0, b1) ", 31, " VP_(b0, b1) "\n\t"                                     \
  "v_mad_u64_u32 " VP_(b2, b3) ", %[cy], " VR_(b0) ", %[nq1], " VP_(b2, b3) "\n\t"             \
  "v_mad_u64_u32 " VP_(b4, b5) ", %[cy], " VR_(b0) ", %[nq0], " VP_(b4, b5) "\n\t"             \
  "v_mad_u64_u32 " VP_(b2, b3) ", %[cy], " VR_(b1) ", %[nq0], " VP_(b2, b3) "\n\t"             \
  "v_add_u32 " VR_(b5) ", " VR_(b5) ", " VR_(b2) "\n\t"
// X' = X + T,  Y' = X + (2q + 1) + ~T   (T in v[b4:b5]; b6,b7: ~T; b8,b9: X + 2q + 1)
#define BFLY_NC_TAIL(X, YN, b4, b5, b6, b7, b8, b9)                                             \
  "v_not_b32 " VR_(b6) ", " VR_(b4) "\n\t"                                                     \
  "v_not_b32 " VR_(b7) ", " VR_(b5) "\n\t"                                                     \
  "v_lshl_add_u64 " VP_(b8, b9) ", " X ", 0, %[twoq1]\n\t"                                     \
  "v_lshl_add_u64 " X ", " X ", 0, " VP_(b4, b5) "\n\t"                                        \
  "v_lshl_add_u64 " YN ", " VP_(b8, b9) ", 0, " VP_(b6, b7) "\n\t"
__device__ __forceinline__ void bfly_fwd63_x2_s_small(u64& Xa, u64& Ya, const Tw63& ta, u64& Xb, u64& Yb, const Tw63& tb, const Mod63& m) {
  u64 cy, ya_new, yb_new;
  asm(MULMOD63_SMALL_ASM(104, 105, 106, 107, 108, 109, "%[ya0]", "%[wa0]", "%[wa1]", "%[pa0]", "%[pa1]")
      MULMOD63_SMALL_ASM(116, 117, 118, 119, 120, 121, "%[yb0]", "%[wb0]", "%[wb1]", "%[pb0]", "%[pb1]")
      BFLY_NC_TAIL("%[xa]", "%[yan]", 108, 109, 110, 111, 112, 113)
      BFLY_NC_TAIL("%[xb]", "%[ybn]", 120, 121, 122, 123, 124, 125)
      : [xa] "+v"(Xa), [xb] "+v"(Xb), [yan] "=&v"(ya_new), [ybn] "=&v"(yb_new), [cy] "=&s"(cy)
      : [ya0] "v"((u32)Ya), [yb0] "v"((u32)Yb), [wa0] "s"(ta.w0), [wa1] "s"(ta.w1), [pa0] "s"(ta.p0), [pa1] "s"(ta.p1), [wb0] "s"(tb.w0), [wb1] "s"(tb.w1),
        [pb0] "s"(tb.p0), [pb1] "s"(tb.p1), [nq0] "s"(m.nq0), [nq1] "s"(m.nq1), [twoq1] "s"(m.twoq1)
      : "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123",
        "v124", "v125");
  Ya = ya_new;
  Yb = yb_new;
}
__device__ __forceinline__ void bfly_fwd63_x2_s_nc(u64& Xa, u64& Ya, const Tw63& ta, u64& Xb, u64& Yb, const Tw63& tb, const Mod63& m) {
  u64 cy, ya_new, yb_new;
  asm(MULMOD63_ASM(104, 105, 106, 107, 108, 109, 110, "%[ya0]", "%[ya1]", "%[wa0]", "%[wa1]", "%[pa0]", "%[pa1]")
      MULMOD63_ASM(116, 117, 118, 119, 120, 121, 122, "%[yb0]", "%[yb1]", "%[wb0]", "%[wb1]", "%[pb0]", "%[pb1]")
      BFLY_NC_TAIL("%[xa]", "%[yan]", 108, 109, 110, 111, 112, 113)
      BFLY_NC_TAIL("%[xb]", "%[ybn]", 120, 121, 122, 123, 124, 125)
      : [xa] "+v"(Xa), [xb] "+v"(Xb), [yan] "=&v"(ya_new), [ybn] "=&v"(yb_new), [cy] "=&s"(cy)
      : [ya0] "v"((u32)Ya), [ya1] "v"((u32)(Ya >> 32)), [yb0] "v"((u32)Yb), [yb1] "v"((u32)(Yb >> 32)), [wa0] "s"(ta.w0), [wa1] "s"(ta.w1), [pa0] "s"(ta.p0),
        [pa1] "s"(ta.p1), [wb0] "s"(tb.w0), [wb1] "s"(tb.w1), [pb0] "s"(tb.p0), [pb1] "s"(tb.p1), [nq0] "s"(m.nq0), [nq1] "s"(m.nq1), [twoq1] "s"(m.twoq1)
      : "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123",
        "v124", "v125");
  Ya = ya_new;
  Yb = yb_new;
}

// two independent multiplications by wave-uniform constants (the last inverse stage: sum * 1/n and difference * w/n); results below 2q
__device__ __forceinline__ void mulmod63_x2_s(u64& Ya, const Tw63& ta, u64& Yb, const Tw63& tb, const Mod63& m) {
  u64 cy, ra, rb;
  asm(MULMOD63_ASM(104, 105, 106, 107, 108, 109, 110, "%[ya0]", "%[ya1]", "%[wa0]", "%[wa1]", "%[pa0]", "%[pa1]")
      MULMOD63_ASM(116, 117, 118, 119, 120, 121, 122, "%[yb0]", "%[yb1]", "%[wb0]", "%[wb1]", "%[pb0]", "%[pb1]")
      "v_lshl_add_u64 %[ra], v[108:109], 0, 0\n\t"
      "v_lshl_add_u64 %[rb], v[120:121], 0, 0\n\t"
      : [ra] "=&v"(ra), [rb] "=&v"(rb), [cy] "=&s"(cy)
      : [ya0] "v"((u32)Ya), [ya1] "v"((u32)(Ya >> 32)), [yb0] "v"((u32)Yb), [yb1] "v"((u32)(Yb >> 32)), [wa0] "s"(ta.w0), [wa1] "s"(ta.w1), [pa0] "s"(ta.p0),
        [pa1] "s"(ta.p1), [wb0] "s"(tb.w0), [wb1] "s"(tb.w1), [pb0] "s"(tb.p0), [pb1] "s"(tb.p1), [nq0] "s"(m.nq0), [nq1] "s"(m.nq1)
      : "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v116", "v117", "v118", "v119", "v120", "v121", "v122");
  Ya = ra;
  Yb = rb;
}

// exact normalisations on store.  Conditional subtraction without compares: values stay below 2^63, so the sign of v - c
// tells whether to add c back (5 VALU instructions per step instead of a compare / select / borrow chain with VCC wait states)
__device__ __forceinline__ u64 csub63(u64 v, u64 c) {                 // v < 2^63, c < 2^62:  v >= c ? v - c : v
  const u64 d = v - c;
  u32 mask;                                                   // (asm: keeps the compiler from turning this back into cmp + select)
  asm("v_ashrrev_i32 %0, 31, %1" : "=v"(mask) : "v"((u32)(d >> 32)));
  return d + (((u64)(mask & (u32)(c >> 32)) << 32) | (mask & (u32)c));
}
// Forward store, two residues per block.  v < 4q + 2^32 < 5q: the quotient k = floor(v / q) <= 4 is estimated from the high word,
//   e = hi32(v.hi * M + 2^(31+b)) >> (b - 1) = floor(v.hi * M / 2^(31+b)) + 1,   M = floor(2^(31+b) / (q.hi + 1))   (PrimeConst::norm_m),
// an under-estimate of v / q by less than 5 / q.hi + 2^-29 (q.hi >= 2^16), plus one: e is k or k + 1, so r = v - e q lies in [-q, q) and the
// sign of its high word says whether to add q back.  9 instructions per residue where three csub63 steps took 24 (they were all of
// the 3.4 instructions per butterfly that the forward kernel issued above its 24-instruction butterflies).
struct Norm63 { u32 m, sh; u64 c; };      // M, b - 1, 2^(31+b)
__device__ __forceinline__ Norm63 make_norm63(u64 q, u32 norm_m) {
  const u32 b = 32u - (u32)__builtin_clz((u32)(q >> 32));
  return Norm63{norm_m, b - 1, 1ull << (31 + b)};
}
__device__ __forceinline__ void norm_fwd63_x2(u64& Va, u64& Vb, const Mod63& m, const Norm63& nm) {
  u64 cy, oa, ob;
  asm("v_mad_u64_u32 v[104:105], %[cy], %[va1], %[M], %[C]\n\t"
      "v_mad_u64_u32 v[116:117], %[cy], %[vb1], %[M], %[C]\n\t"
      "v_lshrrev_b32 v106, %[sh], v105\n\t"
      "v_lshrrev_b32 v118, %[sh], v117\n\t"
      "v_mad_u64_u32 v[108:109], %[cy], v106, %[nq0], %[va]\n\t"
      "v_mul_lo_u32 v107, v106, %[nq1]\n\t"
      "v_mad_u64_u32 v[120:121], %[cy], v118, %[nq0], %[vb]\n\t"
      "v_mul_lo_u32 v119, v118, %[nq1]\n\t"
      "v_add_u32 v109, v109, v107\n\t"
      "v_add_u32 v121, v121, v119\n\t"
      "v_ashrrev_i32 v106, 31, v109\n\t"
      "v_ashrrev_i32 v118, 31, v121\n\t"
      "v_and_b32 v104, %[q0], v106\n\t"
      "v_and_b32 v105, %[q1], v106\n\t"
      "v_and_b32 v116, %[q0], v118\n\t"
      "v_and_b32 v117, %[q1], v118\n\t"
      "v_lshl_add_u64 %[oa], v[104:105], 0, v[108:109]\n\t"
      "v_lshl_add_u64 %[ob], v[116:117], 0, v[120:121]\n\t"
      : [oa] "=&v"(oa), [ob] "=&v"(ob), [cy] "=&s"(cy)
      : [va] "v"(Va), [vb] "v"(Vb), [va1] "v"((u32)(Va >> 32)), [vb1] "v"((u32)(Vb >> 32)), [M] "v"(nm.m), [C] "s"(nm.c), [sh] "s"(nm.sh),
        [nq0] "s"(m.nq0), [nq1] "s"(m.nq1), [q0] "s"((u32)m.q), [q1] "s"((u32)(m.q >> 32))
      : "v104", "v105", "v106", "v107", "v108", "v109", "v116", "v117", "v118", "v119", "v120", "v121");
  Va = oa;
  Vb = ob;
}
// v < 2q -> [0,q), two residues per block (outputs of mulmod63): 5 instructions each
__device__ __forceinline__ void norm_inv63_x2(u64& Va, u64& Vb, const Mod63& m) {
  u64 oa, ob;
  const u64 nq = ((u64)m.nq1 << 32) | m.nq0;
  asm("v_lshl_add_u64 v[104:105], %[va], 0, %[nq]\n\t"
      "v_lshl_add_u64 v[116:117], %[vb], 0, %[nq]\n\t"
      "v_ashrrev_i32 v106, 31, v105\n\t"
      "v_ashrrev_i32 v118, 31, v117\n\t"
      "v_and_b32 v108, %[q0], v106\n\t"
      "v_and_b32 v109, %[q1], v106\n\t"
      "v_and_b32 v120, %[q0], v118\n\t"
      "v_and_b32 v121, %[q1], v118\n\t"
      "v_lshl_add_u64 %[oa], v[108:109], 0, v[104:105]\n\t"
      "v_lshl_add_u64 %[ob], v[120:121], 0, v[116:117]\n\t"
      : [oa] "=&v"(oa), [ob] "=&v"(ob)
      : [va] "v"(Va), [vb] "v"(Vb), [nq] "s"(nq), [q0] "s"((u32)m.q), [q1] "s"((u32)(m.q >> 32))
      : "v104", "v105", "v106", "v108", "v109", "v116", "v117", "v118", "v120", "v121");
  Va = oa;
  Vb = ob;
}
