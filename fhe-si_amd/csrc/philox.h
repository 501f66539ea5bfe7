// philox.h -- the counter-based random numbers behind on-device sampling (NumbTh.cpp:340-404, FHE-SI.cpp:14-25,174-190).
//
// The reference draws from NTL's sequential PRNG (SetSeed / RandomBnd) and lrand48, which nothing outside an NTL process can reproduce;
// a device needs a generator whose k-th number does not depend on who asks.  The definition shared by this library, the C oracle and
// the Python model (each states it on its own):
//   generator   Philox-4x32-10 (Salmon et al., SC'11): key = (seed low word, seed high word),
//               counter = (coefficient index j, object index low word, object index high word, purpose << 16 | block)
//   purposes    0 binary polynomial r of Encrypt (FHE-SI.cpp:14-18)    1, 2 noise of ciphertext part 0, 1 (:24-25)
//               3 random polynomial of a key-switch column (:176-179)   4 its error (:190)   5 sampleHWt draws   6 DoubleCRT::sampleGaussian
//   binary      word 0, bit 0
//   Gaussian    the distribution of round(N(0, 3.2^2)) -- what sampleGaussian's Box-Muller + floor(x + 0.5) produces (NumbTh.cpp:377-404 with
//               FHEContext.h:106's stdev) -- by inversion in INTEGER arithmetic: u = word0 | word1 << 32, magnitude = number of table
//               entries below or equal to u (kGaussCdf: floor(2^64 P(|X| <= k)), k = 0 .. 29), sign = word 2 bit 0.  No floating point
//               anywhere, so host and device agree bit for bit.
//   uniform     SampleRandom(poly, 2^logQ, n) (Util.cpp:49-55: RandomBnd(q) - q / 2): logQ random bits taken limb by limb from blocks
//               0, 1, ... (limb i = words 2 (i mod 2), 2 (i mod 2) + 1 of block i / 2), minus 2^(logQ-1)
//   sampleHWt   draw t = 0, 1, ...: position (word0 | word1 << 32) mod n, value +1 if word 2 bit 0 else -1, kept when the position is
//               still zero, until Hwt positions are set (NumbTh.cpp:340-360)
//
// NOT a cryptographic generator: Philox-4x32-10 has a 64-bit key and no security claim.  It is here because its k-th number is a pure function
// of (key, counter) that three independent implementations can agree on bit for bit -- fixtures, parity tests, multi-GPU runs that must
// produce one key on every rank.  Rules for callers (include/fhesi_hip.h): the seed is secret uniform 64 bits, an (seed, object index) pair is
// never used twice, and the one PUBLIC stream (purpose 3, the polynomial a of a key-switch column) is keyed by a separate public seed so that
// publishing it says nothing about the secret streams.  Key material with a stronger requirement uses the explicit-randomness entry points.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define PHILOX_HD __host__ __device__ __forceinline__
#else
#define PHILOX_HD inline
#endif

struct Philox4 { uint32_t w[4]; };
PHILOX_HD Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    if (r) { k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
  }
  return Philox4{{c0, c1, c2, c3}};
}
enum { PHX_BINARY = 0, PHX_NOISE0 = 1, PHX_NOISE1 = 2, PHX_KEY_POLY = 3, PHX_KEY_ERR = 4, PHX_HWT = 5, PHX_GAUSS = 6 };
PHILOX_HD Philox4 phx_draw(uint64_t seed, uint64_t object, uint32_t j, uint32_t purpose, uint32_t block) {
  return philox4x32_10(j, (uint32_t)object, (uint32_t)(object >> 32), purpose << 16 | block, (uint32_t)seed, (uint32_t)(seed >> 32));
}
// floor(2^64 P(|X| <= k)) for X = round(N(0, 3.2^2)), k = 0 .. 29 (the last entry saturated); tools/gauss_table.py regenerates it
#define PHX_GAUSS_ENTRIES 30
#define PHX_GAUSS_TABLE { \
  0x1fc936cfb902b000ull, 0x5c5a3878a5513000ull, 0x90ba6b457f6e7800ull, 0xb9d6e65c2e45a000ull, 0xd7212f1da26f6200ull, 0xea12314c4c373a00ull, \
  0xf53070328c8acd80ull, 0xfb1cdac9f40b6980ull, 0xfdfa2ace3c107960ull, 0xff3c09d0d6606540ull, 0xffbc45110abb0cdcull, 0xffeaa36c3c86f3c9ull, \
  0xfff9db4fc8bf1e60ull, 0xfffe63d9a0b88f51ull, 0xffff9da028c31301ull, 0xffffea9fbab7b7e9ull, 0xfffffbc5e81f8a58ull, 0xffffff3d57e1db7bull, \
  0xffffffe027626c9eull, 0xfffffffb4348bc92ull, 0xffffffff5c0429c2ull, 0xffffffffebd8d7e9ull, 0xfffffffffdbfd855ull, 0xffffffffffc58a4cull, \
  0xfffffffffffa9c86ull, 0xffffffffffff8c81ull, 0xfffffffffffff738ull, 0xffffffffffffff65ull, 0xfffffffffffffff7ull, 0xffffffffffffffffull }
PHILOX_HD int64_t phx_gaussian(const Philox4& d) {
  const uint64_t tab[PHX_GAUSS_ENTRIES] = PHX_GAUSS_TABLE;
  const uint64_t u = (uint64_t)d.w[0] | (uint64_t)d.w[1] << 32;
  int k = 0;
  for (int i = 0; i < PHX_GAUSS_ENTRIES; ++i) k += u > tab[i];          // magnitude = number of entries below u
  return (d.w[2] & 1) ? -(int64_t)k : (int64_t)k;
}
