"""CPU: integer models of the 32-bit kernels' lazy arithmetic, step for step with wrap-around, on worst-case and random operands.

The kernels keep values in ranges that are argued in their comments (below 4p, below 2^61, "at most 2 below the quotient", no 64-bit
overflow of a multiply-add chain).  These models follow the instruction sequences of
  * rns32_one            (kernels_tensor32.hip: big integer -> residue below 3p, one v_mad_u64_u32 chain with folds through 2^32 mod p),
  * the tensor loader's red()  (ntt32_core.inc, ntt32_inv_kernel3<.., TENSOR>: a product or a sum of two products -> [0, 2p)),
  * a32_ct<NEGW> / a32_gs (ntt32_core.inc: the forward / inverse butterflies on lazy values),
  * mul_lazy32            (ntt32_core.inc),
  * dot32_kernel4 / dot32_kernel2(p)  (kernels_aux32.hip: 64-bit totals of products between folds, the Montgomery step of the epilogue),
  * tensor_sum32_kernel   (kernels_tensor32.hip: eight products per reduction),
with the table entries as the host builders compute them, and check every intermediate bound the comments state plus the congruence of
the result.  What the GPU computes is checked against the oracle in the `-m gpu` tests; this file checks that the RANGES hold for operands the
random GPU inputs may never hit (all-ones words, p - 1 everywhere, the most negative coefficient)."""
import random

import pytest

import params as P
import fhesi_pyref as R

M32 = (1 << 32) - 1
M64 = (1 << 64) - 1


def primes_below_2_30(count, step):
    """the tensor half's rule (t32_plan): the largest primes below 2^30 that are 1 mod `step`"""
    out = []
    p = (1 << 30) - ((1 << 30) % step) + 1
    while len(out) < count:
        p -= step
        if R.is_prime(p):
            out.append(p)
    return out


def mid_primes(count, step, start):
    out = []
    p = start - (start % step) + 1
    while len(out) < count:
        p -= step
        if R.is_prime(p):
            out.append(p)
    return out


def primes_below_2_29(count, step):
    """option tensor_bits = 29 (t32_plan): the largest primes below 2^29 that are 1 mod `step`"""
    out = []
    p = (1 << 29) - ((1 << 29) % step) + 1
    while len(out) < count:
        p -= step
        if R.is_prime(p):
            out.append(p)
    return out


def t32_shift(p):
    """ntt32_core.inc: 29 for primes of 30 bits, 28 for primes of 29 bits; mu = floor(2^(32 + sh) / p) fits 32 bits"""
    return 29 if p >> 29 else 28


TENSOR_PRIMES = primes_below_2_30(70, 1 << 16)[::9] + primes_below_2_30(35, 1 << 15)[-2:]
TENSOR_PRIMES_29 = primes_below_2_29(72, 1 << 16)[::9] + primes_below_2_29(37, 1 << 15)[-2:] + primes_below_2_29(76, 1 << 17)[-2:]
GENERIC_PRIMES = mid_primes(2, 1 << 15, (1 << 30) - (1 << 28)) + mid_primes(2, 1 << 15, (1 << 29) + (1 << 27)) + mid_primes(1, 1 << 15, 1 << 29)


# ---------------------------------------------------------------------------------------------- rns32_one
def rns_table(p, lift, nl):
    b32 = (1 << 32) % p
    t, cur = [], lift % p
    for _ in range(2 * nl):
        t.append(cur)
        cur = cur * b32 % p
    mu = (1 << (32 + t32_shift(p))) // p
    assert mu <= M32 or p < (1 << 28), "the quotient constant is a 32-bit word"
    return t, (p - cur) % p, b32, mu


def rns32_one(x, neg, t, tneg, r32, mu, p):
    acc, room = 0, 4

    def fold(a):
        v = (a >> 32) * r32 + (a & M32)
        assert v <= M64
        return v
    for k in range(len(x)):
        if room == 0:
            acc = fold(acc)
            room = 3
        acc = x[k] * t[k] + acc
        assert acc <= M64, "the multiply-add chain overflowed 64 bits"
        room -= 1
    acc = neg * tneg + acc
    assert acc <= M64
    acc = fold(acc)
    sh = t32_shift(p)
    if r32 >= (1 << (sh - 1)):
        acc = fold(acc)
    assert acc < (1 << (32 + sh)), "the Barrett step takes a value below 2^(32 + sh)"
    q = (((acc >> sh) & M32) * mu) >> 32
    assert (acc >> sh) <= M32
    r = ((acc & M32) - q * p) & M32
    assert acc // p - 2 <= q <= acc // p
    return r


@pytest.mark.parametrize("nl", [8, 16])
@pytest.mark.parametrize("p", TENSOR_PRIMES + TENSOR_PRIMES_29 + GENERIC_PRIMES)
def test_rns32_chain_never_overflows_and_stays_below_3p(nl, p):
    rng = random.Random(p * 31 + nl)
    for lift in (1, 23, 65537, 32603, (1 << 20) - 3):
        t, tneg, r32, mu = rns_table(p, lift, nl)
        cases = [[M32] * (2 * nl), [0] * (2 * nl), [M32] * (2 * nl - 1) + [0x7FFFFFFF], [0] * (2 * nl - 1) + [0x80000000], [1] + [0] * (2 * nl - 1),
                 [M32] * (2 * nl - 1) + [0x80000000]]
        cases += [[rng.getrandbits(32) for _ in range(2 * nl)] for _ in range(300)]
        for x in cases:
            neg = x[-1] >> 31
            r = rns32_one(x, neg, t, tneg, r32, mu, p)
            val = sum(w << (32 * k) for k, w in enumerate(x)) - (neg << (64 * nl))
            assert r < 3 * p and r % p == (val * lift) % p, (p, lift, x[-1])


# ---------------------------------------------------------------------------------------------- tensor loader
def red61(x, p, mu):
    sh = t32_shift(p)
    assert x < (1 << (32 + sh)) and mu <= M32
    q = (((x >> sh) & M32) * mu) >> 32
    r = ((x & M32) - q * p) & M32
    assert r < 3 * p
    return r - 2 * p if r >= 2 * p else r


@pytest.mark.parametrize("p", TENSOR_PRIMES + TENSOR_PRIMES_29 + GENERIC_PRIMES)
def test_tensor_loader_reduction(p):
    rng = random.Random(p)
    mu = (1 << (32 + t32_shift(p))) // p
    ext = [0, 1, p - 1, p - 2, p // 2]
    pairs = [(a, b) for a in ext for b in ext] + [(rng.randrange(p), rng.randrange(p)) for _ in range(2000)]
    for a, b in pairs:
        r = red61(a * b, p, mu)
        assert r < 2 * p and r % p == a * b % p
    quads = [(p - 1,) * 4, (p - 1, p - 1, 0, 0), (p - 1, p - 2, p - 2, p - 1)] + [tuple(rng.randrange(p) for _ in range(4)) for _ in range(2000)]
    for a0, b1, a1, b0 in quads:
        x = a0 * b1 + a1 * b0
        r = red61(x, p, mu)
        assert r < 2 * p and r % p == x % p


# ---------------------------------------------------------------------------------------------- butterflies
def tw(w, p):
    return w, (w << 32) // p


def mul_lazy32(y, w, wp, p):
    return (y * w - ((y * wp) >> 32) * p) & M32


def a32_ct_negw(x, y, w, p):
    """forward butterfly, table entry (-w mod 2^32, floor(w 2^32 / p)): inputs below 4p -> outputs below 4p"""
    wneg, wp = (-w) & M32, (w << 32) // p
    twop = 2 * p
    X = min(x, (x - twop) & M32)
    Q = (y * wp) >> 32
    nT = (Q * p + y * wneg) & M32                     # -T mod 2^32, T in [0, 2p)
    T = (-nT) & M32
    assert T < twop and T % p == y * w % p
    return (X - nT) & M32, (X + nT + twop) & M32


def a32_gs(x, y, w, p):
    """inverse butterfly (A32_GS7): inputs below 2p -> sum below 2p, product below 2p"""
    wv, wp = tw(w, p)
    twop = 2 * p
    s, d = (x + y) & M32, (x - y + twop) & M32
    xo = min(s, (s - twop) & M32)
    Q = (d * wp) >> 32
    yo = (Q * ((-p) & M32) + d * wv) & M32
    return xo, yo


@pytest.mark.parametrize("p", TENSOR_PRIMES[:4] + GENERIC_PRIMES[:2])
def test_butterflies_keep_their_ranges(p):
    rng = random.Random(p + 7)
    ws = [1, p - 1, 2, p // 2, rng.randrange(1, p), rng.randrange(1, p)]
    lazy = [0, 1, p - 1, p, 2 * p - 1, 2 * p, 3 * p, 4 * p - 1]
    for w in ws:
        for x in lazy + [rng.randrange(4 * p) for _ in range(200)]:
            for y in lazy + [rng.randrange(4 * p) for _ in range(20)]:
                xo, yo = a32_ct_negw(x, y, w, p)
                assert xo < 4 * p and yo < 4 * p
                assert xo % p == (x + w * y) % p and yo % p == (x - w * y) % p
        small = [v for v in lazy if v < 2 * p]
        for x in small + [rng.randrange(2 * p) for _ in range(200)]:
            for y in small + [rng.randrange(2 * p) for _ in range(20)]:
                xo, yo = a32_gs(x, y, w, p)
                assert xo < 2 * p and yo < 2 * p
                assert xo % p == (x + y) % p and yo % p == (x - y) * w % p
        for y in [0, 1, p, M32, M32 - 1, 1 << 31] + [rng.getrandbits(32) for _ in range(500)]:      # any 32-bit value -> [0, 2p)
            wv, wp = tw(w, p)
            r = mul_lazy32(y, wv, wp, p)
            assert r < 2 * p and r % p == y * w % p


# ---------------------------------------------------------------------------------------------- primes below 2^29: fewer range steps
def a32_ct29(x, y, w, p, step):
    """a32_ct<NEGW>(.., step): 0 = x as it is, 2 = min(x, x - 4p); every sum must stay a 32-bit value WITHOUT wrapping"""
    wneg, wp = (-w) & M32, (w << 32) // p
    X = x if step == 0 else min(x, (x - 4 * p) & M32)
    Q = (y * wp) >> 32
    nT = (Q * p + y * wneg) & M32
    T = (-nT) & M32
    assert T < 2 * p and T % p == y * w % p
    assert X + T <= M32 and X + 2 * p - T >= 0 and X - T + 2 * p <= M32
    return (X - nT) & M32, (X + nT + 2 * p) & M32


def fwd_step29(g):
    return 2 if g >= 2 and not g & 1 else 0


@pytest.mark.parametrize("p", TENSOR_PRIMES_29[:3] + TENSOR_PRIMES_29[-2:])
def test_forward_rows_of_29_bit_primes_skip_range_steps(p):
    """ntt32_fwd_kernel3<.., PB = 29>: rows enter below 4p; stage g takes its x input as it is except the even stages from 2 on; the bound
    4p -> 6p -> 8p -> (step) 6p -> 8p ... never wraps 32 bits and the store's three range steps end below p"""
    assert 8 * p <= M32 + 1
    rng = random.Random(p)
    bound = 4 * p
    vals = [4 * p - 1, 0, p, 3 * p + 5] + [rng.randrange(4 * p) for _ in range(60)]
    want = [v % p for v in vals]
    for g in range(14):
        st = fwd_step29(g)
        if st == 2:
            assert bound <= 8 * p
            bound = 4 * p
        bound += 2 * p
        assert bound <= 8 * p
        nxt, nwant = [], []
        for i in range(0, len(vals), 2):
            w = rng.choice([1, p - 1, rng.randrange(1, p)])
            for (x, y, wx, wy) in ((vals[i], vals[i + 1], want[i], want[i + 1]), (bound - 2 * p - 1 if st == 0 else 8 * p - 1, M32, None, None)):
                if wx is None:
                    if st == 2:
                        x = min(x, 8 * p - 1)
                    xo, yo = a32_ct29(x, y, w, p, st)
                    assert xo < bound and yo < bound
                    continue
                xo, yo = a32_ct29(x, y, w, p, st)
                assert xo < bound and yo < bound
                nxt += [xo, yo]
                nwant += [(wx + w * wy) % p, (wx - w * wy) % p]
        vals, want = nxt, nwant
    assert bound == 8 * p
    for v, wv in zip(vals + [8 * p - 1, 4 * p, 2 * p, p], want + [(8 * p - 1) % p, 0, 0, 0]):
        v = min(v, (v - 4 * p) & M32)
        v = min(v, (v - 2 * p) & M32)
        v = min(v, (v - p) & M32)
        assert v < p and v == wv


def a32_gs29(x, y, w, p, fresh):
    """a32_gs29 (ntt32_core.inc): fresh = both inputs below 2p (product outputs of the stage before); otherwise both below 4p"""
    wv, wp = tw(w, p)
    lim = 2 * p if fresh else 4 * p
    assert x < lim and y < lim
    s = x + y
    d = x - y + lim
    assert 0 <= d <= M32 and s <= M32
    xo = s if fresh else min(s, (s - 4 * p) & M32)
    Q = (d * wp) >> 32
    yo = (Q * ((-p) & M32) + d * wv) & M32
    return xo, yo


@pytest.mark.parametrize("p", TENSOR_PRIMES_29[:3] + TENSOR_PRIMES_29[-2:])
def test_inverse_butterfly_of_29_bit_primes(p):
    """sum outputs below 4p, product outputs below 2p, whichever form runs; the last stage's sums and differences (below 8p) enter mulc"""
    assert 8 * p <= M32 + 1
    rng = random.Random(p + 1)
    ws = [1, p - 1, 2, rng.randrange(1, p), rng.randrange(1, p)]
    for w in ws:
        for fresh in (True, False):
            lim = 2 * p if fresh else 4 * p
            ext = [0, 1, p, lim - 1, lim // 2]
            for x in ext + [rng.randrange(lim) for _ in range(100)]:
                for y in ext + [rng.randrange(lim) for _ in range(20)]:
                    xo, yo = a32_gs29(x, y, w, p, fresh)
                    assert xo < 4 * p and yo < 2 * p
                    assert xo % p == (x + y) % p and yo % p == (x - y) * w % p
                    # the last stage (ntt32_inv_kernel3): sm = x + y, df = x - y + lim, each through mulc (any 32-bit value -> [0, 2p))
                    sm, df = x + y, x - y + lim
                    assert sm <= M32 and 0 <= df <= M32
                    wv, wp = tw(w, p)
                    r = mul_lazy32(df, wv, wp, p)
                    assert r < 2 * p and r % p == (x - y) * w % p


# ---------------------------------------------------------------------------------------------- dot32_kernel4
def dot4_model(digs, keys, p, KC):
    """one output of dot32_kernel4 (kernels_aux32.hip): digits reduced below p at use, KC products per chunk on top of a folded total, the fold
    hi (2^32 mod p) + lo with 2^32 mod p = 2^32 - 4p, the Montgomery step of the epilogue"""
    r32 = (0 - 4 * p) & M32
    assert r32 == (1 << 32) % p and r32 < (1 << 28), "launch_dot32_k4 admits primes with 2^32 mod p below 2^28 only"
    mont = (-pow(p, -1, 1 << 32)) & M32
    acc = 0
    for c0 in range(0, len(digs), KC):
        for d, k in zip(digs[c0:c0 + KC], keys[c0:c0 + KC]):
            t = min(d, (d - 2 * p) & M32)
            t = min(t, (t - p) & M32)
            assert t < p and k < p
            acc += k * t
            assert acc <= M64, "a chunk's products overflowed the 64-bit total"
        acc = (acc >> 32) * r32 + (acc & M32)
        assert acc < (1 << 60) + (1 << 32)
    mq = ((acc & M32) * mont) & M32
    ov = (acc + mq * p) >> 32
    assert (acc + mq * p) & M32 == 0 and ov < 2 * p
    return min(ov, (ov - p) & M32)


@pytest.mark.parametrize("step", [1 << 15, 1 << 16, 1 << 17, 1 << 18, 1 << 20, 1 << 21])
@pytest.mark.parametrize("ncol", [24, 66, 129])
def test_dot32_kernel4_totals(step, ncol):
    rng = random.Random(step + ncol)
    for p in primes_below_2_30(4, step):
        if (1 << 32) % p >= (1 << 28):          # (rows of 2^20: above the launcher's guard -- those rings take dot32_kernel2)
            assert step == 1 << 21
            continue
        rinv = pow(1 << 32, -1, p)
        lazy = lambda v: v + rng.randrange(4) * p if v + 3 * p <= M32 else v       # the digit rows arrive below 4p
        cases = [([p - 1 + 3 * p] * ncol, [p - 1] * ncol), ([0] * ncol, [p - 1] * ncol), ([4 * p - 1] * ncol, [p - 1] * ncol)]
        cases += [([lazy(rng.randrange(p)) for _ in range(ncol)], [rng.randrange(p) for _ in range(ncol)]) for _ in range(200)]
        for digs, keys in cases:
            for KC in (8, 12):
                o = dot4_model(digs, keys, p, KC)
                assert o < p and o == sum(d * k for d, k in zip(digs, keys)) * rinv % p


# ---------------------------------------------------------------------------------------------- dot32_kernel2 / dot32_kernel2p
def dot2_model(digs, keys, p, ncp):
    """one output of dot32_kernel2p (column parts of ncp columns; dot32_kernel2 = one part): 64-bit totals of products below p^2, folded into
    48-bit units (the high part counted in a 32-bit word) every 16 columns, r48 = 2^48 mod p and one Montgomery step at the end"""
    r48 = (1 << 48) % p
    mont = (-pow(p, -1, 1 << 32)) & M32
    tot, th = 0, 0

    def add(k0, n):
        nonlocal tot
        for k in range(k0, k0 + n):
            assert digs[k] < p and keys[k] < p               # (the tile loader reduces the lazy digit words before they enter LDS)
            tot += digs[k] * keys[k]
            assert tot <= M64, "sixteen products on top of a folded total overflowed 64 bits"

    def fold():
        nonlocal tot, th
        th += tot >> 48
        tot &= (1 << 48) - 1
        assert th <= M32
    CH = 4
    for kbeg in range(0, len(digs), ncp):
        nc = min(ncp, len(digs) - kbeg)
        nfull, n2 = nc & ~(CH - 1), nc & ~(2 * CH - 1)
        for i in range(n2 // (2 * CH)):
            add(kbeg + i * 2 * CH, 2 * CH)
            if i & 1:
                fold()
        if nfull & CH:
            add(kbeg + n2, CH)
        if nfull & (3 * CH):
            fold()
        add(kbeg + nfull, nc - nfull)
        fold()
    v = tot + th * r48
    assert v < (1 << 62)
    mq = ((v & M32) * mont) & M32
    o = (v + mq * p) >> 32
    assert o < 2 * p
    return min(o, (o - p) & M32)


@pytest.mark.parametrize("ncol,ncp", [(66, 66), (129, 72), (24, 24), (17, 17), (43, 24), (258, 136)])
def test_dot32_kernel2_totals(ncol, ncp):
    rng = random.Random(ncol * 7 + ncp)
    for step in (1 << 15, 1 << 17):
        for p in primes_below_2_30(4, step):
            rinv = pow(1 << 32, -1, p)
            cases = [([p - 1] * ncol, [p - 1] * ncol), ([0] * ncol, [p - 1] * ncol)]
            cases += [([rng.randrange(p) for _ in range(ncol)], [rng.randrange(p) for _ in range(ncol)]) for _ in range(100)]
            for digs, keys in cases:
                o = dot2_model(digs, keys, p, ncp)
                assert o < p and o == sum(d * k for d, k in zip(digs, keys)) * rinv % p


# ---------------------------------------------------------------------------------------------- tensor_sum32_kernel (sums of tensor products)
def tsum_red(v, p, r32, mu):
    sh = t32_shift(p)
    assert v <= M64 and mu <= M32
    v = (v >> 32) * r32 + (v & M32)
    assert v < (1 << 62) + (1 << 32)
    v = (v >> 32) * r32 + (v & M32)
    assert v < (1 << (32 + sh))
    q = (((v >> sh) & M32) * mu) >> 32
    r = ((v & M32) - q * p) & M32
    assert r < 3 * p
    return r - 2 * p if r >= 2 * p else r


@pytest.mark.parametrize("p", TENSOR_PRIMES[:3] + TENSOR_PRIMES_29[:3] + TENSOR_PRIMES_29[-2:] + GENERIC_PRIMES)
def test_tensor_sum32_totals(p):
    """four terms per round, the middle component two products per term: eight products below p^2 on top of a reduced total"""
    rng = random.Random(p + 99)
    r32, mu = (1 << 32) % p, (1 << (32 + t32_shift(p))) // p
    for nterms in (1, 3, 4, 5, 8, 13):
        for worst in (True, False):
            r1, ref = 0, 0
            t = 0
            while t < nterms:
                n = min(4, nterms - t) if t + 4 <= nterms else 1
                for _ in range(n):
                    a0, a1, b0, b1 = ((p - 1,) * 4) if worst else tuple(rng.randrange(p) for _ in range(4))
                    r1 += a0 * b1 + a1 * b0
                    ref += a0 * b1 + a1 * b0
                    assert r1 <= M64
                t += n
                if n == 4:
                    r1 = tsum_red(r1, p, r32, mu)
            r1 = tsum_red(r1, p, r32, mu)
            assert r1 < 2 * p and r1 % p == ref % p
