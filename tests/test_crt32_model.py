"""CPU: an integer model of crt32_scale_kernel<512, EXACT, 28, 38, 0> (fhe-si_amd/csrc/kernels_tensor32.hip) -- the carry-free CRT of the tensor half
with its truncated word window and the "undecided" flag.

The kernel forms x = sum_i y_i M_i - kappa M (M_i = M / p_i in words of 28 bits, kappa from a 25-bit fixed-point sum) and returns
round(x / 2^logQ) mod 2^logQ (Ciphertext.cpp:194-218: ScaleDown's round-half-up, then Reduce).  In its fast form only the words from bit 392 upwards
are formed; a workgroup whose bits logQ-64 .. logQ-1 read 0x7fff...f is flagged and redone with every word.  The model follows the kernel (tables as
t32_config builds them, wrap-around included) and checks, for coefficients placed ON the rounding boundaries as well as random ones:
  * no 64-bit accumulator overflows;
  * kappa is the right multiple of M;
  * whenever the fast form does NOT flag a coefficient its limbs equal the exact rounding -- and the exact form always does."""
import random

import pytest

import test_arith32_models as A

M32, M64 = (1 << 32) - 1, (1 << 64) - 1
LQ, R, WT = 512, 28, 38
WU = (2 * LQ + R - 1) // R
J0_FAST = (LQ - 64 - 30 - 8) // R


def tables(primes):
    M = 1
    for p in primes:
        M *= p
    Mi = [M // p for p in primes]
    cinv = [pow(mi % p, -1, p) for mi, p in zip(Mi, primes)]
    inv57 = [(1 << 57) // p for p in primes]
    word = lambda v, l: (v >> (R * l)) & ((1 << R) - 1)
    Mw = [[word(mi, l) for l in range(WT)] for mi in Mi]
    N = ((1 << (R * WT)) - M) & ((1 << (R * WT)) - 1)
    Nw = [word(N, l) for l in range(WT)]
    return M, cinv, inv57, Mw, Nw


def crt32_scale(res, primes, tb, exact):
    M, cinv, inv57, Mw, Nw = tb
    J0 = 0 if exact else J0_FAST
    NW = WU - J0
    acc = [0] * NW
    fsum = 0
    for i, p in enumerate(primes):
        w, wp = cinv[i], (cinv[i] << 32) // p
        y = A.mul_lazy32(res[i], w, wp, p)
        y = y - p if y >= p else y
        assert y < p and y == res[i] * cinv[i] % p
        fsum = (fsum + ((y * inv57[i]) >> 32)) & M32
        for l in range(NW):
            acc[l] += y * Mw[i][J0 + l]
            assert acc[l] <= M64
    kappa = ((fsum + (1 << 24)) & M32) >> 25
    for l in range(NW):
        acc[l] += kappa * Nw[J0 + l]
        assert acc[l] <= M64
    carry = 0
    for l in range(NW):
        v = acc[l] + carry
        assert v <= M64
        acc[l] = v & ((1 << R) - 1)
        carry = v >> R

    def limb(B):
        l0, o = B // R - J0, B % R
        v = acc[l0] >> o
        if l0 + 1 < NW: v |= acc[l0 + 1] << (R - o)
        if l0 + 2 < NW: v |= acc[l0 + 2] << (2 * R - o)
        if l0 + 3 < NW and 3 * R - o < 64: v |= acc[l0 + 3] << (3 * R - o)
        return v & M64
    G = limb(LQ - 64)
    undecided = (not exact) and G == 0x7FFFFFFFFFFFFFFF
    c = G >> 63
    out = 0
    for i in range(LQ // 64):
        v = (limb(LQ + 64 * i) + c) & M64
        c = 1 if (c and v == 0) else 0
        out |= v << (64 * i)
    return out, undecided, kappa


def expected(x):
    return ((x + (1 << (LQ - 1))) >> LQ) & ((1 << LQ) - 1)          # floor((2x + q) / 2q) with q = 2^logQ, then mod 2^logQ (two's complement)


def test_crt32_scale_window_and_flag():
    primes = A.primes_below_2_30(35, 1 << 15)
    tb = tables(primes)
    M = tb[0]
    rng = random.Random(512)
    bound = M // 8                                   # what t32_plan leaves: |x| < M / 8, so the fixed-point kappa cannot be off
    xs = [0, 1, -1, bound - 1, -(bound - 1), (1 << 511), (1 << 511) - 1, -(1 << 511), -(1 << 511) - 1, (1 << 1030), -(1 << 1030)]
    for _ in range(60):                              # ON the rounding boundary: x + 2^511 = t 2^512 - 1 - delta, delta from 0 to beyond the dropped part
        t = rng.randrange(-(bound >> 513), bound >> 513)
        for delta in (0, 1, rng.randrange(1 << 60), rng.randrange(1 << 392), rng.randrange(1 << 430), (1 << 448) + rng.randrange(1 << 440)):
            xs.append(t * (1 << 512) - (1 << 511) - 1 - delta)
            xs.append(t * (1 << 512) - (1 << 511) + delta)
    xs += [rng.randrange(-bound + 1, bound) for _ in range(150)]
    nflag = 0
    for x in xs:
        assert abs(x) < bound
        res = [x % p for p in primes]
        o_fast, und, kappa = crt32_scale(res, primes, tb, exact=False)
        o_exact, _, kappa_e = crt32_scale(res, primes, tb, exact=True)
        y = [r * c % p for r, c, p in zip(res, tb[1], primes)]
        s = sum(yi * (M // p) for yi, p in zip(y, primes))
        assert kappa == kappa_e and s - kappa * M == x, "kappa is the multiple of M that brings the sum back to x"
        assert o_exact == expected(x)
        if und:
            nflag += 1
        else:
            assert o_fast == expected(x), hex(x)
    assert nflag > 0, "the boundary cases must exercise the flag"
