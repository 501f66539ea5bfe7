"""CPU: an integer model of the 64-bit row kernels' store normalisation (norm_fwd63_x2, fhe-si_amd/csrc/modarith63.h) on the host.

The kernel reduces a lazy butterfly output v < 4q + 2^32 to [0, q) with ONE quotient estimate taken from the high word of v and one sign
fix-up.  The model below follows the instruction sequence word for word (32 / 64-bit wrap-around included) with the constant as
capi_ctx.hip computes it (PrimeConst::norm_m), and is checked against v mod q on the boundaries of every quotient value and on random values, for
every modulus size the tile kernels compute with (2^48 <= q_tile < 2^60; smaller chain primes are scaled up to a multiple in [2^59, 2^60), ntt_tile.inc).
The GPU side of the same statement is tests/test_gpu_ntt.py::test_tile_rows_across_prime_sizes.

Second part: the 60-bit butterflies themselves -- the carry-free quotient chain of mulmod63 is the exact floor and never overflows, forward values
stay below 4q + 2^32, inverse values below 2q + 2^47 with the high-word-only range step (the invariants of modarith63.h's header)."""
import random

import pytest

import params as P

M64 = (1 << 64) - 1
M32 = (1 << 32) - 1


def norm_m(q_tile: int) -> int:
    """capi_ctx.hip: floor(2^(31+b) / ((q_tile >> 32) + 1)), b = bit length of q_tile >> 32"""
    qh = q_tile >> 32
    b = qh.bit_length()
    m = (1 << (31 + b)) // (qh + 1)
    assert (1 << 31) <= m < (1 << 32)
    return m


def norm_fwd63(v: int, q: int, m: int) -> int:
    """the nine instructions of one residue in norm_fwd63_x2"""
    b = (q >> 32).bit_length()
    c = 1 << (31 + b)
    vh = v >> 32
    e64 = vh * m + c                                   # v_mad_u64_u32: no overflow (asserted)
    assert e64 <= M64
    k = ((e64 >> 32) & M32) >> (b - 1)                 # v_lshrrev_b32 of the high word
    nq = (-q) & M64
    r = (k * (nq & M32) + v) & M64                     # v_mad_u64_u32 k, nq0, v
    t = (k * (nq >> 32)) & M32                         # v_mul_lo_u32 k, nq1
    r = (r & M32) | ((((r >> 32) + t) & M32) << 32)    # v_add_u32 on the high word
    neg = (r >> 63) & 1                                # v_ashrrev_i32 31 of the high word
    return (r + (q if neg else 0)) & M64               # two v_and, v_lshl_add_u64


def tile_modulus(q: int) -> int:
    """capi_ctx.hip: q itself, or the largest multiple of a small prime below 2^60"""
    return q if q >= (1 << 48) else q * (((1 << 60) - 1) // q)


@pytest.mark.parametrize("bits", [20, 37, 47, 48, 49, 50, 53, 56, 59, 60])
def test_store_normalisation_model(bits):
    rng = random.Random(bits)
    primes, _ = P.first_primes(1 << 14, 3, sp_nbits=bits)
    for p in primes:
        q = tile_modulus(p)
        assert (1 << 48) <= q < (1 << 60)
        m = norm_m(q)
        top = 4 * q + (1 << 32)                        # exclusive bound of the butterflies' lazy range
        edge = []
        for k in range(5):
            for d in (-2, -1, 0, 1, 2):
                edge.append(k * q + d)
            edge += [k * q + (1 << 32) - 1, k * q + (1 << 32), (k * q) | M32, ((k * q) >> 32) << 32]
        edge += [top - 1, top - 2, 0, 1, M32, 1 << 32]
        vals = [v for v in edge if 0 <= v < top] + [rng.randrange(top) for _ in range(20000)]
        for v in vals:
            assert norm_fwd63(v, q, m) == v % q, (bits, p, v)


def test_constant_is_the_one_the_library_computes():
    """the expression in capi_ctx.hip, as text (the library itself cannot be asked without a GPU)"""
    import os
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fhe-si_amd", "csrc", "capi_ctx.hip")).read()
    assert "pc.norm_m = (u32)((((u128)1) << (31 + b)) / (qh + 1));" in src
    assert "const int b = 64 - __builtin_clzll(qh);" in src


# ---------------------------------------------------------------------------------------------- the 60-bit butterflies (modarith63.h)
def mulmod63(y: int, w: int, q: int) -> int:
    """mulmod63 / MULMOD63_ASM: the quotient floor(y wq / 2^63), wq = floor(w 2^63 / q), as one carry-free chain; remainder mod 2^64"""
    wq = (w << 63) // q
    y0, y1, p0, p1, w0, w1 = y & M32, y >> 32, wq & M32, wq >> 32, w & M32, w >> 32
    assert y1 < (1 << 31) and p1 < (1 << 31)
    m = (y0 * p0) >> 32
    m = y1 * p0 + m
    assert m <= M64
    m = y0 * p1 + m
    assert m <= M64, "the middle sum of the quotient chain overflowed"
    qq = ((y1 << 1) & M32) * p1 + (m >> 31)
    assert qq <= M64 and qq == (y * wq) >> 63, "the chain is the exact floor"
    nq = (-q) & M64
    q0, q1 = qq & M32, (qq >> 32) & M32
    r = y0 * w0 + q0 * (nq & M32)                      # two v_mad_u64_u32 (mod 2^64: the carry out is dropped)
    hi = ((r >> 32) + y0 * w1 + y1 * w0 + q0 * (nq >> 32) + q1 * (nq & M32)) & M32
    t = (r & M32) | (hi << 32)
    assert t == (y * w - qq * q) & M64 and t % q == y * w % q
    return t


def csel(v: int, q: int) -> int:
    """the butterflies' range step: compares HIGH words against (2q) >> 32 only"""
    return (v - 2 * q) & M64 if (v >> 32) > ((2 * q) >> 32) else v


@pytest.mark.parametrize("bits", [48, 50, 55, 59, 60])
def test_butterflies_63_keep_their_ranges(bits):
    rng = random.Random(bits * 13)
    primes, _ = P.first_primes(1 << 14, 2, sp_nbits=bits)
    for q in primes + (tile_modulus(P.first_primes(1 << 14, 1, sp_nbits=37)[0][0]),):
        q = int(q)
        fwd_top = 4 * q + (1 << 32)
        ws = [1, q - 1, 2, q // 2, rng.randrange(1, q), rng.randrange(1, q)]
        edge = [0, 1, q - 1, q, 2 * q - 1, 2 * q, 2 * q + (1 << 32) - 1, 2 * q + (1 << 32), ((2 * q) >> 32 << 32) + M32, 4 * q, fwd_top - 1]
        for w in ws:
            for x in edge + [rng.randrange(fwd_top) for _ in range(60)]:
                for y in edge + [rng.randrange(fwd_top) for _ in range(12)]:
                    t = mulmod63(y, w, q)
                    assert t < 2 * q                                           # y < 2^63 - 2^33
                    xc = csel(x, q)
                    assert xc < 2 * q + (1 << 32)
                    xo, yo = (xc + t) & M64, (xc + 2 * q - t) & M64
                    assert xo < fwd_top and yo < fwd_top
                    assert xo % q == (x + w * y) % q and yo % q == (x - w * y) % q
            # inverse: inputs below 2q + e, e doubling from 2^32 to 2^47 over the stages
            e = 1 << 32
            while e <= (1 << 47):
                top = 2 * q + e
                for x in [0, q, 2 * q, top - 1] + [rng.randrange(top) for _ in range(10)]:
                    for y in [0, q, top - 1] + [rng.randrange(top) for _ in range(6)]:
                        s = x + y
                        xo = csel(s, q)
                        d = x + 3 * q - y
                        assert d < (1 << 63) - (1 << 33)
                        yo = mulmod63(d, w, q)
                        assert xo < 2 * q + max(1 << 32, 2 * e) and yo < 2 * q
                        assert xo % q == (x + y) % q and yo % q == (x - y) * w % q
                e <<= 1
