"""Pure-Python big-integer restatement of fhe-si's DoubleCRT hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product path: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker.  This file is the *independent* restatement (Python ``int`` is exact)
used to generate ``tests/golden/*.json`` and to cross-check the C restatement
(``oracle/fhesi_oracle.c``).  It is meant for small cases (pure-Python loops).

PARITY UNPINNED: the reference (dwu4/fhe-si) needs NTL, which is not installed and not vendored,
so it cannot be built or run here; its own tests hold no golden vectors or known-answer tests
for this path.  What pins this restatement is (i) algebra -- for fixed
(m, q, root, input) every function below has one canonical output -- (ii) agreement of the
literal Bluestein restatement with the reference's own slow definition ``tDFT``
(bluestein.cpp:149-172), and (iii) the reference's end-to-end predicate
decrypt(f(enc x)) == f(x) (Test_AddMul.cpp:84-86).

Every function cites the reference file:line it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

MASK64 = (1 << 64) - 1


# --------------------------------------------------------------------------------------------
# PRNG shared by the Python restatement, the C oracle and the C++ host harness (replaces
# srand48/SetSeed/RandomBnd/lrand48, which are NTL-version specific: SURVEY.md H7).
# --------------------------------------------------------------------------------------------
class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & MASK64

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)

    def bits(self, nbits: int) -> int:
        """nbits uniform bits, little-endian 64-bit words."""
        out, sh = 0, 0
        while sh < nbits:
            out |= self.next() << sh
            sh += 64
        return out & ((1 << nbits) - 1)

    def bnd(self, n: int) -> int:
        """Uniform in [0,n) by rejection on ceil(log2 n) bits (role of NTL RandomBnd)."""
        if n <= 1:
            return 0
        k = (n - 1).bit_length()
        while True:
            v = self.bits(k)
            if v < n:
                return v


# --------------------------------------------------------------------------------------------
# number theory helpers (NumbTh.cpp)
# --------------------------------------------------------------------------------------------
def is_prime(n: int) -> bool:
    """Deterministic Miller-Rabin for n < 2^64 (role of NTL ProbPrime, FHEContext.cpp:34,108)."""
    if n < 2:
        return False
    small = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37)
    for p in small:
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in small:
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def factorize(n: int) -> List[int]:
    """distinct prime factors (NumbTh.cpp factorize, used by FindPrimRootT :97-98)."""
    f, d = [], 2
    while d * d <= n:
        if n % d == 0:
            f.append(d)
            while n % d == 0:
                n //= d
        d += 1
    if n > 1:
        f.append(n)
    return f


def mobius(n: int) -> int:
    """NumbTh.cpp:124-137."""
    mu = 1
    for p in factorize(n):
        if (n // p) % p == 0:
            return 0
        mu = -mu
    return mu


def poly_mul(a: Sequence[int], b: Sequence[int]) -> List[int]:
    if not a or not b:
        return []
    r = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                r[i + j] += x * y
    return r


def poly_divexact(num: List[int], den: List[int]) -> List[int]:
    num = list(num)
    q = [0] * (len(num) - len(den) + 1)
    for i in range(len(q) - 1, -1, -1):
        c = num[i + len(den) - 1] // den[-1]
        q[i] = c
        for j, d in enumerate(den):
            num[i + j] -= c * d
    assert not any(num)
    return q


def cyclotomic(m: int) -> List[int]:
    """Phi_m(X), low-to-high integer coefficients (NumbTh.cpp:142-158 Cyclotomic)."""
    num, den = [1], [1]
    for d in range(1, m + 1):
        if m % d == 0:
            g = [0] * (m // d + 1)
            g[0], g[-1] = -1, 1
            mu = mobius(d)
            if mu == 1:
                num = poly_mul(num, g)
            elif mu == -1:
                den = poly_mul(den, g)
    return poly_divexact(num, den)


def zms_idx(m: int) -> Tuple[List[int], int]:
    """PAlgebra::init (PAlgebra.cpp:40-56): zmsIdx[t] = rank of t in Z_m^* or -1; returns (idx, phi(m))."""
    idx, k = [-1] * m, 0
    for i in range(m):
        if math.gcd(i, m) == 1:
            idx[i] = k
            k += 1
    return idx, k


def find_root_2m(q: int, m: int) -> int:
    """A primitive 2m-th root of unity mod q.

    The reference draws it at random (NumbTh.cpp:101-115 FindPrimRootT); any primitive 2m-th
    root is valid and the row values depend on which one is used, so every fixture states its
    root explicitly.  This restatement is deterministic: smallest base s>=2 whose
    s^((q-1)/2m) has exact order 2m (same acceptance test as NumbTh.cpp:104-113).
    """
    e = 2 * m
    assert (q - 1) % e == 0
    facts = factorize(e)
    ex = (q - 1) // e
    for s in range(2, 1000):
        r = pow(s, ex, q)
        if pow(r, e, q) != 1:
            continue
        if all(pow(r, e // f, q) != 1 for f in facts):
            return r
    raise RuntimeError("no 2m-th root found")


# --------------------------------------------------------------------------------------------
# FHEcontext chain (FHEContext.cpp / FHEContext.h)
# --------------------------------------------------------------------------------------------
def ndigits(logQ: int, decomp_size: int = 3) -> int:
    """FHEContext.h:115."""
    return (logQ + 8 * decomp_size - 1) // (8 * decomp_size)


def si_context_size(logQ: int, p: int, phim: int, xi: int = 1) -> float:
    """FHEContext.cpp:83-85 SetUpSIContext: natural-log size handed to AddPrimesBySize."""
    return logQ * math.log(2.0) * 2 + math.log(p) + math.log(phim) * 2 + math.log(2) + math.log(xi)


def add_primes_by_size(m: int, total_size: float, sp_nbits: int = 60) -> List[int]:
    """FHEContext.cpp:88-115 AddPrimesBySize (special=false)."""
    chain: List[int] = []
    p = (1 << sp_nbits) - 1
    two_m = 2 * m
    p -= p % two_m
    p += two_m + 1
    last = False
    left = total_size
    while left > 0.0:
        if left < math.log(float(p)) and not last:
            last = True
            p = int(math.ceil(math.exp(left)))
            p -= (p % two_m) - 1
            two_m = -two_m
        while True:
            p -= two_m
            if is_prime(p):
                break
        if p not in chain:
            chain.append(p)
            left -= math.log(float(p))
    return chain


# --------------------------------------------------------------------------------------------
# Bluestein / Cmodulus (bluestein.cpp, CModulus.cpp)
# --------------------------------------------------------------------------------------------
def tdft(a: Sequence[int], n: int, root: int, q: int) -> List[int]:
    """bluestein.cpp:149-172 tDFT: x[k] = sum_i a[i] root^{ki} (slow definition)."""
    if not any(a) or n <= 0:
        return [0] * max(n, 0)
    a = list(a[:n]) + [0] * (n - len(a))
    return [sum(a[i] * pow(root, k * i, q) for i in range(n)) % q for k in range(n)]


def bluestein_fft(a: Sequence[int], n: int, root: int, q: int) -> List[int]:
    """bluestein.cpp:93-144 tBluesteinFFT, restated literally.

    x[k] = sum_i a[i] root^{2ik}, unscaled; a truncated/zero-padded to n (":111-113").
    powers[i] = root^{i^2 mod 2n} (:103-109); b[n-1+-i] = root^{-i^2} (:121-133); cyclic product
    of size N = 2^ceil(log2(2n-1)) (:116-119,138); window n-1..2n-2 (:139); post-multiply (:140-142).
    """
    if not any(x % q for x in a[:n]) or n <= 0:
        return [0] * max(n, 0)
    powers = [pow(root, (i * i) % (2 * n), q) for i in range(n)]
    x = [(a[i] if i < len(a) else 0) * powers[i] % q for i in range(n)]
    k = (2 * n - 2).bit_length() if n > 1 else 0   # NextPowerOfTwo(2n-1)
    N = 1 << k
    rinv = pow(root, -1, q)
    b = [0] * N
    b[n - 1] = 1
    for i in range(1, n):
        bi = pow(rinv, (i * i) % (2 * n), q)
        b[n - 1 + i] = bi
        b[n - 1 - i] = bi
    # exact cyclic convolution mod q of length N (role of NTL fftRep mul, bluestein.cpp:138)
    c = [0] * N
    for i, xi in enumerate(x):
        if xi:
            for j, bj in enumerate(b):
                if bj:
                    c[(i + j) % N] += xi * bj
    out = [c[n - 1 + i] % q * powers[i] % q for i in range(n)]
    return out


def cmod_fft(x: Sequence[int], m: int, q: int, root: int, use_bluestein: bool = True) -> List[int]:
    """Cmod::FFT (CModulus.cpp:90-107): y[j] = x(omega^{i_j}), omega=root^2, i_j ascending in Z_m^*.

    x: signed big-integer coefficients of any magnitude (reduced mod q by conv, :96).
    """
    idx, _ = zms_idx(m)
    xin = [c % q for c in x[:m]]
    full = bluestein_fft(xin, m, root, q) if use_bluestein else tdft(xin, m, root * root % q, q)
    return [full[i] for i in range(m) if idx[i] >= 0]


def poly_rem_monic(a: List[int], f: Sequence[int], q: int) -> List[int]:
    a = [c % q for c in a]
    df = len(f) - 1
    for i in range(len(a) - 1, df - 1, -1):
        c = a[i]
        if c:
            for j in range(df + 1):
                a[i - df + j] = (a[i - df + j] - c * f[j]) % q
    return a[:df]


def cmod_ifft(y: Sequence[int], m: int, q: int, root: int, use_bluestein: bool = True) -> List[int]:
    """Cmod::iFFT (CModulus.cpp:110-132): scatter (:117-121), Bluestein with rInv (:124), /m (:125),
    reduce mod (Phi_m, q) (:128-129).  Returns phi(m) coefficients in [0,q)."""
    idx, phim = zms_idx(m)
    vec = [0] * m
    for i in range(m):
        if idx[i] >= 0:
            vec[i] = y[idx[i]] % q
    rinv = pow(root, -1, q)
    full = bluestein_fft(vec, m, rinv, q) if use_bluestein else tdft(vec, m, rinv * rinv % q, q)
    minv = pow(m, -1, q)
    full = [v * minv % q for v in full]
    return poly_rem_monic(full, cyclotomic(m), q)


def negacyclic_ntt_direct(x: Sequence[int], n: int, q: int, psi: int) -> List[int]:
    """for m=2n a power of two Cmod::FFT is y[j] = sum_k x_k psi^{(2j+1)k}, psi=root^2."""
    return [sum(x[k] * pow(psi, (2 * j + 1) * k, q) for k in range(len(x))) % q for j in range(n)]


# --------------------------------------------------------------------------------------------
# DoubleCRT (DoubleCRT.cpp) -- a DoubleCRT is a dict {prime_index: row}
# --------------------------------------------------------------------------------------------
class Ctx:
    """Minimal FHEcontext restatement (FHEContext.h:105-118): m, logQ, p, chain primes and roots."""

    def __init__(self, m: int, logQ: int, p: int, primes: Sequence[int], roots: Sequence[int] = None,
                 decomp_size: int = 3):
        self.m, self.logQ, self.p = m, logQ, p
        self.idx, self.phim = zms_idx(m)
        self.primes = list(primes)
        for q in self.primes:
            # FHEContext.cpp:31-34
            assert is_prime(q) and q % (2 * m) == 1
        assert len(set(self.primes)) == len(self.primes)
        self.roots = list(roots) if roots is not None else [find_root_2m(q, m) for q in self.primes]
        self.decomp_size = decomp_size
        self.ndigits = ndigits(logQ, decomp_size)
        self.phi = cyclotomic(m)
        self.pow2 = (m & (m - 1)) == 0

    @property
    def L(self):
        return len(self.primes)

    def fft(self, i: int, x: Sequence[int]) -> List[int]:
        if self.pow2 and self.m >= 4:
            n = self.m // 2
            xin = [c % self.primes[i] for c in x[:self.m]]
            # fold degree >= n terms?  Cmod::FFT evaluates x (deg < m) at primitive m-th roots; X^n = -1 there.
            xx = [0] * n
            for k, c in enumerate(xin):
                if k < n:
                    xx[k] = (xx[k] + c) % self.primes[i]
                else:
                    xx[k - n] = (xx[k - n] - c) % self.primes[i]
            return _ntt_pow2(xx, n, self.primes[i], self.roots[i] ** 2 % self.primes[i])
        return cmod_fft(x, self.m, self.primes[i], self.roots[i])

    def ifft(self, i: int, y: Sequence[int]) -> List[int]:
        if self.pow2 and self.m >= 4:
            n = self.m // 2
            return _intt_pow2(list(y), n, self.primes[i], self.roots[i] ** 2 % self.primes[i])
        return cmod_ifft(y, self.m, self.primes[i], self.roots[i])


def _brv(x: int, bits: int) -> int:
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


def _ntt_pow2(a: List[int], n: int, q: int, psi: int) -> List[int]:
    """Fast path used only to keep the pure-Python model usable at n up to ~2^12; checked against
    negacyclic_ntt_direct / cmod_fft in tests."""
    lg = n.bit_length() - 1
    tab = [pow(psi, _brv(i, lg), q) for i in range(n)]
    a = [c % q for c in a] + [0] * (n - len(a))
    t, mm = n, 1
    while mm < n:
        t //= 2
        for i in range(mm):
            s = tab[mm + i]
            j1 = 2 * i * t
            for j in range(j1, j1 + t):
                u, v = a[j], a[j + t] * s % q
                a[j], a[j + t] = (u + v) % q, (u - v) % q
        mm *= 2
    return [a[_brv(j, lg)] for j in range(n)]


def _intt_pow2(y: List[int], n: int, q: int, psi: int) -> List[int]:
    lg = n.bit_length() - 1
    ipsi = pow(psi, -1, q)
    tab = [pow(ipsi, _brv(i, lg), q) for i in range(n)]
    a = [y[_brv(p, lg)] % q for p in range(n)]
    t, mm = 1, n
    while mm > 1:
        h = mm // 2
        j1 = 0
        for i in range(h):
            s = tab[h + i]
            for j in range(j1, j1 + t):
                u, v = a[j], a[j + t]
                a[j], a[j + t] = (u + v) % q, (u - v) * s % q
            j1 += 2 * t
        t *= 2
        mm = h
    ninv = pow(n, -1, q)
    return [c * ninv % q for c in a]


def dcrt_from_poly(ctx: Ctx, poly: Sequence[int], idxset: Sequence[int] = None) -> dict:
    """DoubleCRT(const ZZX&, ...) (DoubleCRT.cpp:212-257): one Cmod::FFT per prime in the set."""
    s = range(ctx.L) if idxset is None else idxset
    return {i: ctx.fft(i, poly) for i in s}


def dcrt_op(ctx: Ctx, a: dict, b: dict, op: str) -> dict:
    """DoubleCRT::Op (DoubleCRT.cpp:79-113) for matching index sets; op in add/sub/mul."""
    assert set(a) == set(b)
    f = {"add": lambda x, y, q: (x + y) % q, "sub": lambda x, y, q: (x - y) % q,
         "mul": lambda x, y, q: x * y % q}[op]
    return {i: [f(x, y, ctx.primes[i]) for x, y in zip(a[i], b[i])] for i in a}


def dcrt_op_scalar(ctx: Ctx, a: dict, num: int, op: str) -> dict:
    """DoubleCRT::Op(const ZZ&) (DoubleCRT.cpp:115-129): n = num % q_i (non-negative), then Fnc."""
    f = {"add": lambda x, y, q: (x + y) % q, "sub": lambda x, y, q: (x - y) % q,
         "mul": lambda x, y, q: x * y % q}[op]
    return {i: [f(x, num % ctx.primes[i], ctx.primes[i]) for x in a[i]] for i in a}


def dcrt_exp(ctx: Ctx, a: dict, e: int) -> dict:
    """DoubleCRT::Exp (DoubleCRT.cpp:423-434): row[j] = PowerMod(row[j], e, q_i); a negative exponent inverts first
    (NTL PowerMod), which is an error for a zero element."""
    return {i: [pow(x, e, ctx.primes[i]) for x in a[i]] for i in a}


def dcrt_div_scalar(ctx: Ctx, a: dict, num: int) -> dict:
    """DoubleCRT::operator/=(ZZ) (DoubleCRT.cpp:407-420)."""
    return {i: [x * pow(num % ctx.primes[i], -1, ctx.primes[i]) % ctx.primes[i] for x in a[i]] for i in a}


def dcrt_automorph(ctx: Ctx, a: dict, k: int) -> dict:
    """DoubleCRT::automorph (DoubleCRT.cpp:439-465): new[idx(j)] = old[idx(j*k mod m)]."""
    m, idx = ctx.m, ctx.idx
    assert 0 < k < m and idx[k] >= 0, "k not in Zm*"
    out = {}
    for i, row in a.items():
        new = list(row)
        for j in range(1, m):
            if idx[j] >= 0:
                new[idx[j]] = row[idx[(j * k) % m]]
        out[i] = new
    return out


def int_vec_crt(vp: List[int], p: int, vq: Sequence[int], q: int) -> None:
    """intVecCRT (NumbTh.cpp:307-335), in place on vp; short vq treated as zero tail (:325-333)."""
    pinv = pow(p % q, -1, q)
    q2 = q // 2
    for i in range(len(vp)):
        vqi = vq[i] if i < len(vq) else 0
        d = ((vqi - vp[i] % q) % q) * pinv % q
        if d > q2:
            d -= q
        vp[i] += d * p


def dcrt_to_poly(ctx: Ctx, a: dict, idxset: Sequence[int] = None, positive: bool = False) -> List[int]:
    """DoubleCRT::toPoly (DoubleCRT.cpp:349-398).  Returns phi(m) coefficients (not normalized)."""
    s1 = sorted(set(a) if idxset is None else set(a) & set(idxset))
    if not s1:
        return [0] * ctx.phim
    i0 = s1[0]
    p = ctx.primes[i0]
    vp = ctx.ifft(i0, a[i0])
    vp = vp + [0] * (ctx.phim - len(vp))
    vp = [v - p if v > p // 2 else v for v in vp]
    for i in s1[1:]:
        q = ctx.primes[i]
        int_vec_crt(vp, p, ctx.ifft(i, a[i]), q)
        p *= q
    if positive:
        vp = [v + p if v < 0 else v for v in vp]
    return vp


# --------------------------------------------------------------------------------------------
# Util.cpp / Ciphertext.cpp / FHE-SI.cpp
# --------------------------------------------------------------------------------------------
def reduce_logq(val: int, logQ: int, positive: bool = False) -> int:
    """Reduce (Util.cpp:3-26): canonical residue mod 2^logQ, centered to [-q/2, q/2) unless positive."""
    v = val & ((1 << logQ) - 1)
    if not positive and v >= (1 << (logQ - 1)):
        v -= 1 << logQ
    return v


def scale_down_coeff(x: int, logQ: int) -> int:
    """Ciphertext::ScaleDown per coefficient (Ciphertext.cpp:205-213): floor((2x+q)/(2q)) then Reduce."""
    q = 1 << logQ
    return reduce_logq((2 * x + q) // (2 * q), logQ)   # Python // floors, like NTL ZZ division


def byte_decomp_part(poly: Sequence[int], logQ: int, nd: int, decomp_size: int = 3) -> List[List[int]]:
    """Ciphertext::ByteDecompPart (Ciphertext.cpp:82-105): nd digit polys of decomp_size bytes each."""
    bits = 8 * decomp_size
    out = [[0] * len(poly) for _ in range(nd)]
    for i, c in enumerate(poly):
        v = reduce_logq(c, logQ, True)
        for d in range(nd):
            out[d][i] = (v >> (bits * d)) & ((1 << bits) - 1)
    return out


def byte_decomp(parts: Sequence[Sequence[int]], logQ: int, nd: int, decomp_size: int = 3) -> List[List[int]]:
    """Ciphertext::ByteDecomp (Ciphertext.cpp:107-121): part-major, digit-minor order."""
    out = []
    for part in parts:
        out.extend(byte_decomp_part(part, logQ, nd, decomp_size))
    return out


def ct_mul(ctx: Ctx, a_parts: Sequence[Sequence[int]], b_parts: Sequence[Sequence[int]]) -> List[dict]:
    """Ciphertext::operator*= (Ciphertext.cpp:167-192): tProd[i+j] += DoubleCRT(a_i*p)*DoubleCRT(b_j)."""
    c1 = [dcrt_from_poly(ctx, [c * ctx.p for c in part]) for part in a_parts]
    c2 = [dcrt_from_poly(ctx, part) for part in b_parts]
    zero = {i: [0] * ctx.phim for i in range(ctx.L)}
    t = [dict(zero) for _ in range(len(c1) + len(c2) - 1)]
    for i, x in enumerate(c1):
        for j, y in enumerate(c2):
            t[i + j] = dcrt_op(ctx, t[i + j], dcrt_op(ctx, x, y, "mul"), "add")
    return t


def ct_scale_down(ctx: Ctx, tprod: Sequence[dict]) -> List[List[int]]:
    """Ciphertext::ScaleDown (Ciphertext.cpp:194-218)."""
    return [[scale_down_coeff(x, ctx.logQ) for x in dcrt_to_poly(ctx, t)] for t in tprod]


def dot_product(ctx: Ctx, v1: Sequence[dict], v2: Sequence[dict]) -> dict:
    """DotProduct<DoubleCRT> (Util.h:79-98)."""
    res = dcrt_op(ctx, v1[0], v2[0], "mul")
    for x, y in zip(v1[1:], v2[1:]):
        res = dcrt_op(ctx, res, dcrt_op(ctx, x, y, "mul"), "add")
    return res


def apply_key_switch(ctx: Ctx, ksm: Sequence[Sequence[dict]], tprod: Sequence[dict]) -> List[List[int]]:
    """KeySwitchSI::ApplyKeySwitch (FHE-SI.cpp:241-260) on a scaled-up ciphertext."""
    parts = ct_scale_down(ctx, tprod)
    digits = byte_decomp(parts, ctx.logQ, ctx.ndigits, ctx.decomp_size)
    bd = [dcrt_from_poly(ctx, d) for d in digits]
    out = []
    for r in range(len(ksm)):
        dp = dot_product(ctx, ksm[r], bd)
        out.append([reduce_logq(c, ctx.logQ) for c in dcrt_to_poly(ctx, dp)])
    return out


def ct_mul_relin(ctx: Ctx, ksm, a_parts, b_parts) -> List[List[int]]:
    """The metric's unit of work (Test_AddMul.cpp:59-67): operator*= then ApplyKeySwitch."""
    return apply_key_switch(ctx, ksm, ct_mul(ctx, a_parts, b_parts))


# ---- key generation / encrypt / decrypt with the documented PRNG (FHE-SI.cpp) -----------------
def poly_mul_mod_phi(ctx: Ctx, a: Sequence[int], b: Sequence[int]) -> List[int]:
    prod = poly_mul(list(a), list(b))
    f, df = ctx.phi, len(ctx.phi) - 1
    prod = prod + [0] * max(0, df - len(prod))
    for i in range(len(prod) - 1, df - 1, -1):
        c = prod[i]
        if c:
            for j in range(df + 1):
                prod[i - df + j] -= c * f[j]
    return prod[:df]


def sample_hwt(rng: SplitMix64, hwt: int, n: int) -> List[int]:
    """sampleHWt (NumbTh.cpp:340-360) with the documented PRNG."""
    poly = [0] * n
    hwt = min(hwt, n)
    i = 0
    while i < hwt:
        u = rng.bnd(n)
        if poly[u] == 0:
            poly[u] = (rng.next() & 2) - 1
            i += 1
    return poly


def sample_gaussian(rng: SplitMix64, n: int, stdev: float = 3.2) -> List[int]:
    """sampleGaussian (NumbTh.cpp:377-404), Box-Muller, with the documented PRNG."""
    bignum = 0xFFFFFFF
    poly = [0] * n
    for i in range(0, n, 2):
        r1 = (1 + rng.bnd(bignum)) / (bignum + 1.0)
        r2 = (1 + rng.bnd(bignum)) / (bignum + 1.0)
        theta = 2 * math.pi * r1
        rr = math.sqrt(-2.0 * math.log(r2)) * stdev
        poly[i] = int(math.floor(rr * math.cos(theta) + 0.5))
        if i + 1 < n:
            poly[i + 1] = int(math.floor(rr * math.sin(theta) + 0.5))
    return poly


def sample_random(rng: SplitMix64, modulus: int, n: int) -> List[int]:
    """SampleRandom (Util.cpp:49-55)."""
    off = modulus // 2
    return [rng.bnd(modulus) - off for _ in range(n)]


def keygen(ctx: Ctx, rng: SplitMix64):
    """FHESISecKey::Init (FHE-SI.cpp:86-91) + FHESIPubKey::Init (:42-63).  Returns (t_coeffs, pk_parts)."""
    n, Q = ctx.phim, 1 << ctx.logQ
    t = sample_hwt(rng, 64, n)
    c0 = sample_gaussian(rng, n)
    c1 = sample_random(rng, Q, n)
    tc1 = poly_mul_mod_phi(ctx, t, c1)
    c0 = [reduce_logq(a + b, ctx.logQ) for a, b in zip(c0, tc1)]
    c1 = [reduce_logq(-c, ctx.logQ) for c in c1]
    return t, [c0, c1]


def encrypt_with(ctx: Ctx, pk, msg: Sequence[int], small: Sequence[int], noise: Sequence[Sequence[int]]) -> List[List[int]]:
    """FHESIPubKey::Encrypt (FHE-SI.cpp:10-36) with the randomness made explicit: `small` = the binary polynomial r (:14-18),
    noise[i] = the Gaussian sample of part i before the multiplication by p (:24-25)."""
    Q = 1 << ctx.logQ
    r = dcrt_from_poly(ctx, small)
    parts = []
    for i in range(2):
        e = dcrt_op_scalar(ctx, dcrt_from_poly(ctx, noise[i]), ctx.p, "mul")
        c = dcrt_op(ctx, dcrt_op(ctx, dcrt_from_poly(ctx, pk[i]), r, "mul"), e, "add")
        parts.append(dcrt_to_poly(ctx, c))
    delta = Q // ctx.p
    parts[0] = [c + delta * (msg[k] if k < len(msg) else 0) for k, c in enumerate(parts[0])]
    return [[reduce_logq(c, ctx.logQ) for c in part] for part in parts]


def encrypt(ctx: Ctx, pk, msg: Sequence[int], rng: SplitMix64) -> List[List[int]]:
    """FHESIPubKey::Encrypt (FHE-SI.cpp:10-36); draw order: r, then the noise of part 0, then of part 1."""
    n = ctx.phim
    small = [rng.bnd(2) for _ in range(n)]
    noise = [sample_gaussian(rng, n), sample_gaussian(rng, n)]
    return encrypt_with(ctx, pk, msg, small, noise)


def decrypt(ctx: Ctx, t: Sequence[int], parts: Sequence[Sequence[int]]) -> List[int]:
    """FHESISecKey::Decrypt (FHE-SI.cpp:93-119): round(p * <c, s> / q) mod p."""
    Q = 1 << ctx.logQ
    skeys = [[1] + [0] * (ctx.phim - 1), list(t)]
    cp = [dcrt_from_poly(ctx, part) for part in parts[:2]]
    sp = [dcrt_from_poly(ctx, s) for s in skeys]
    z = dcrt_to_poly(ctx, dot_product(ctx, cp, sp))
    return [((2 * ctx.p * c + Q) // (2 * Q)) % ctx.p for c in z]


def key_switch_init(ctx: Ctx, src_keys: Sequence[Sequence[int]], t: Sequence[int], rng: SplitMix64):
    """KeySwitchSI::Init (FHE-SI.cpp:153-209): matrix[0]=b, [1]=A, index i*ndigits+j."""
    n, Q = ctx.phim, 1 << ctx.logQ
    tD = dcrt_from_poly(ctx, t)
    A, b = [], []
    for s in src_keys:
        s = list(s)
        for _ in range(ctx.ndigits):
            a_poly = sample_random(rng, Q, n)
            aD = dcrt_from_poly(ctx, a_poly)
            bcoef = dcrt_to_poly(ctx, dcrt_op(ctx, aD, tD, "mul"))
            err = sample_gaussian(rng, n)
            bcoef = [reduce_logq(x + e + sk, ctx.logQ) for x, e, sk in zip(bcoef, err, s)]
            s = [c << (8 * ctx.decomp_size) for c in s]
            A.append(dcrt_op_scalar(ctx, aD, -1, "mul"))
            b.append(dcrt_from_poly(ctx, bcoef))
    return [b, A]


def key_switch_init_s2(ctx: Ctx, t: Sequence[int], rng: SplitMix64):
    """KeySwitchSI::InitS2 (FHE-SI.cpp:211-227): source key (1, t, t^2) in coefficient form mod P."""
    one = [1] + [0] * (ctx.phim - 1)
    # the reference builds `FHESISecKey tensoredKey(context)` (FHE-SI.cpp:221), whose constructor samples (and then
    # discards) a fresh Hamming-weight-64 key: the draw is reproduced so PRNG streams stay aligned with the C++ mirror
    sample_hwt(rng, 64, ctx.phim)
    tD = dcrt_from_poly(ctx, t)
    t2 = dcrt_to_poly(ctx, dcrt_op(ctx, tD, tD, "mul"))
    return key_switch_init(ctx, [one, list(t), t2], t, rng)


# --------------------------------------------------------------------------------------------
# Counter-based randomness for sampling on the device: Philox-4x32-10, key = seed, counter = (coefficient j, object
# index low / high word, purpose << 16 | block).  Independent statement of fhe-si_amd/csrc/philox.h; purposes: 0 binary r, 1 / 2 noise of
# part 0 / 1, 3 key-switch column polynomial, 4 its error, 5 sampleHWt draws, 6 sampleGaussian.
# --------------------------------------------------------------------------------------------
def philox4x32_10(ctr: Sequence[int], key: Sequence[int]) -> List[int]:
    c, k = list(ctr), list(key)
    for r in range(10):
        if r:
            k = [(k[0] + 0x9E3779B9) & 0xFFFFFFFF, (k[1] + 0xBB67AE85) & 0xFFFFFFFF]
        p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & 0xFFFFFFFF, (p0 >> 32) ^ c[3] ^ k[1], p0 & 0xFFFFFFFF]
    return c


def phx_draw(seed: int, obj: int, j: int, purpose: int, block: int = 0) -> List[int]:
    return philox4x32_10([j, obj & 0xFFFFFFFF, obj >> 32, purpose << 16 | block], [seed & 0xFFFFFFFF, seed >> 32])


GAUSS_CDF = [0x1fc936cfb902b000, 0x5c5a3878a5513000, 0x90ba6b457f6e7800, 0xb9d6e65c2e45a000, 0xd7212f1da26f6200, 0xea12314c4c373a00,
             0xf53070328c8acd80, 0xfb1cdac9f40b6980, 0xfdfa2ace3c107960, 0xff3c09d0d6606540, 0xffbc45110abb0cdc, 0xffeaa36c3c86f3c9,
             0xfff9db4fc8bf1e60, 0xfffe63d9a0b88f51, 0xffff9da028c31301, 0xffffea9fbab7b7e9, 0xfffffbc5e81f8a58, 0xffffff3d57e1db7b,
             0xffffffe027626c9e, 0xfffffffb4348bc92, 0xffffffff5c0429c2, 0xffffffffebd8d7e9, 0xfffffffffdbfd855, 0xffffffffffc58a4c,
             0xfffffffffffa9c86, 0xffffffffffff8c81, 0xfffffffffffff738, 0xffffffffffffff65, 0xfffffffffffffff7, 0xffffffffffffffff]


def phx_gaussian(w: Sequence[int]) -> int:
    """round(N(0, 3.2^2)) by inversion on integers: magnitude = number of table entries below u, sign = bit 0 of word 2"""
    u = w[0] | w[1] << 32
    k = sum(1 for t in GAUSS_CDF if u > t)
    return -k if w[2] & 1 else k


def draw_encrypt(n: int, seed: int, index: int):
    """(r, e0, e1) of FHESIPubKey::Encrypt (FHE-SI.cpp:14-25) for plaintext `index`"""
    return ([phx_draw(seed, index, j, 0)[0] & 1 for j in range(n)], [phx_gaussian(phx_draw(seed, index, j, 1)) for j in range(n)],
            [phx_gaussian(phx_draw(seed, index, j, 2)) for j in range(n)])


def draw_keygen(n: int, logQ: int, seed: int, index: int):
    """(SampleRandom polynomial modulo 2^logQ, Gaussian error) of key-switch column `index` (FHE-SI.cpp:174-190, Util.cpp:49-55)"""
    poly = []
    for j in range(n):
        u = 0
        for i in range((logQ + 63) // 64):
            w = phx_draw(seed, index, j, 3, i // 2)
            u |= (w[2 * (i % 2)] | w[2 * (i % 2) + 1] << 32) << (64 * i)
        poly.append((u & ((1 << logQ) - 1)) - (1 << (logQ - 1)))
    return poly, [phx_gaussian(phx_draw(seed, index, j, 4)) for j in range(n)]


def draw_hwt(n: int, hwt: int, seed: int, index: int) -> List[int]:
    """sampleHWt (NumbTh.cpp:340-360)"""
    poly, i, t = [0] * n, 0, 0
    hwt = min(hwt, n)
    while i < hwt:
        w = phx_draw(seed, index, t, 5)
        u = (w[0] | w[1] << 32) % n
        if poly[u] == 0:
            poly[u] = 1 if w[2] & 1 else -1
            i += 1
        t += 1
    return poly


# Ciphertext algebra used by Matrix<Ciphertext> / Regression
# --------------------------------------------------------------------------------------------
def ct_add(ctx: Ctx, a_parts: Sequence[Sequence[int]], b_parts: Sequence[Sequence[int]]) -> List[List[int]]:
    """Ciphertext::operator+= on unscaled ciphertexts (Ciphertext.cpp:123-134): parts add + ReduceCoefficients; extra parts of
    the right operand are appended."""
    out = []
    for i in range(max(len(a_parts), len(b_parts))):
        if i < len(a_parts) and i < len(b_parts):
            out.append([reduce_logq(x + y, ctx.logQ) for x, y in zip(a_parts[i], b_parts[i])])
        else:
            out.append(list(a_parts[i] if i < len(a_parts) else b_parts[i]))
    return out


def ct_mul_long(ctx: Ctx, parts: Sequence[Sequence[int]], l: int) -> List[List[int]]:
    """Ciphertext::operator*=(long) on an unscaled ciphertext (Ciphertext.cpp:232-237 -> CiphertextPart::operator*= :21-27)."""
    return [[reduce_logq(c * l, ctx.logQ) for c in part] for part in parts]


def ct_add_const(ctx: Ctx, parts: Sequence[Sequence[int]], other: Sequence[int]) -> List[List[int]]:
    """Ciphertext::operator+=(const ZZX&) on an unscaled ciphertext (Ciphertext.cpp:147-156): scaledConstant[i] = (other[i] << logQ) / p
    -- NTL shifts the magnitude and divides with floor, which is what Python's << and // do on signed integers -- then
    parts[0] += scaledConstant; ReduceCoefficients."""
    sc = [(int(c) << ctx.logQ) // ctx.p for c in other]
    out = [list(part) for part in parts]
    out[0] = [reduce_logq(x + (sc[i] if i < len(sc) else 0), ctx.logQ) for i, x in enumerate(out[0])]
    return out


def ct_mul_poly(ctx: Ctx, parts: Sequence[Sequence[int]], other: Sequence[int]) -> List[List[int]]:
    """Ciphertext::operator*=(const ZZX&) on an unscaled ciphertext (Ciphertext.cpp:245-249) = CiphertextPart::operator*=(const ZZX&)
    (:29-36) on every part: poly *= other over the integers; rem(poly, poly, PhimX); Reduce of every coefficient."""
    return [[reduce_logq(c, ctx.logQ) for c in poly_mul_mod_phi(ctx, list(part), [int(x) for x in other])] for part in parts]


def tprod_mul_long(ctx: Ctx, tprod: Sequence[dict], l: int) -> List[dict]:
    """Ciphertext::operator*=(long) on a scaled-up ciphertext (Ciphertext.cpp:238-241): DoubleCRT *= long."""
    return [dcrt_op_scalar(ctx, t, l, "mul") for t in tprod]


def tprod_add(ctx: Ctx, a: Sequence[dict], b: Sequence[dict]) -> List[dict]:
    """Ciphertext::operator+= on scaled-up ciphertexts (Ciphertext.cpp:135-142)."""
    return [dcrt_op(ctx, x, y, "add") for x, y in zip(a, b)]


def ct_automorph(ctx: Ctx, parts: Sequence[Sequence[int]], k: int) -> List[List[int]]:
    """Ciphertext::operator>>= on an unscaled ciphertext (Ciphertext.cpp:264-269): per part DoubleCRT(poly) >>= k; toPoly
    (CiphertextPart::operator>>=, :54-59).  Coefficients come back centred modulo the whole chain, NOT reduced mod 2^logQ."""
    return [dcrt_to_poly(ctx, dcrt_automorph(ctx, dcrt_from_poly(ctx, part), k)) for part in parts]


def apply_key_switch_parts(ctx: Ctx, ksm: Sequence[Sequence[dict]], parts: Sequence[Sequence[int]]) -> List[List[int]]:
    """KeySwitchSI::ApplyKeySwitch (FHE-SI.cpp:241-260) on an UNSCALED ciphertext: ScaleDown returns at once
    (Ciphertext.cpp:195), ByteDecomp takes the positive residue mod 2^logQ of every coefficient (:94)."""
    digits = byte_decomp(parts, ctx.logQ, ctx.ndigits, ctx.decomp_size)
    bd = [dcrt_from_poly(ctx, d) for d in digits]
    out = []
    for r in range(len(ksm)):
        dp = dot_product(ctx, ksm[r], bd)
        out.append([reduce_logq(c, ctx.logQ) for c in dcrt_to_poly(ctx, dp)])
    return out


def key_switch_init_automorph(ctx: Ctx, t: Sequence[int], k: int, rng: SplitMix64):
    """KeySwitchSI::InitAutomorph (FHE-SI.cpp:229-239): source key (1, t(X^k)), target key t."""
    sample_hwt(rng, 64, ctx.phim)      # `FHESISecKey automorphedKey(context)` (:233) samples and discards a key, as in InitS2
    one = [1] + [0] * (ctx.phim - 1)
    src = []
    for s in (one, list(t)):
        src.append(dcrt_to_poly(ctx, dcrt_automorph(ctx, dcrt_from_poly(ctx, s), k)))
    return key_switch_init(ctx, src, t, rng)


def automorph_generators(m: int, g: int, nslots_usable: int) -> List[int]:
    """The k sequence of Regression::Regression / SumBatchedData (Regression.h:71-80,166-178): g, g^2, g^4, ... mod m,
    one per halving of the usable slot count."""
    ks, k = [], g
    while nslots_usable > 1:
        ks.append(k)
        nslots_usable >>= 1
        k = (k * k) % m
    return ks


def sum_batched_data(ctx: Ctx, auto_ksms, ks: Sequence[int], parts: Sequence[Sequence[int]]) -> List[List[int]]:
    """Regression::SumBatchedData (Regression.h:166-178): ct += KeySwitch_k(ct >>= k) for k = g, g^2, g^4, ..."""
    cur = [list(part) for part in parts]
    for ksm, k in zip(auto_ksms, ks):
        tmp = apply_key_switch_parts(ctx, ksm, ct_automorph(ctx, cur, k))
        cur = ct_add(ctx, cur, tmp)
    return cur


def total_slots(m: int, p: int) -> int:
    """PlaintextSpace::GetTotalSlots (PlaintextSpace.cpp:29-31): number of irreducible factors of Phi_m mod p = phi(m)/ord_m(p)."""
    phim = zms_idx(m)[1]
    d, x = 1, p % m
    while x != 1:
        x = (x * p) % m
        d += 1
    return phim // d


def usable_slots(m: int, p: int) -> int:
    """PlaintextSpace::GetUsableSlots (PlaintextSpace.cpp:38-43): largest power of two <= total slots."""
    u, t = 1, total_slots(m, p)
    while t > 1:
        u <<= 1
        t >>= 1
    return u


# --------------------------------------------------------------------------------------------
# Matrix<Ciphertext> (Matrix.cpp) and Regression::Regress (Regression.h:102-134), literal control flow.
# A ciphertext value is ("parts", [poly, poly, ...]) when unscaled or ("tprod", [dcrt, dcrt, dcrt]) when scaled up.
# --------------------------------------------------------------------------------------------
def _cv_mul(ctx: Ctx, a, b):
    """Ciphertext::operator*=(Ciphertext) (Ciphertext.cpp:167-192); both operands unscaled in every caller."""
    assert a[0] == "parts" and b[0] == "parts"
    return ("tprod", ct_mul(ctx, a[1], b[1]))


def _cv_add(ctx: Ctx, a, b):
    """Ciphertext::operator+= (Ciphertext.cpp:123-145); asserts equal scaling like the reference (:124)."""
    assert a[0] == b[0]
    return ("parts", ct_add(ctx, a[1], b[1])) if a[0] == "parts" else ("tprod", tprod_add(ctx, a[1], b[1]))


def _cv_neg(ctx: Ctx, a):
    """`x *= -1` (Ciphertext.cpp:232-243)."""
    return ("parts", ct_mul_long(ctx, a[1], -1)) if a[0] == "parts" else ("tprod", tprod_mul_long(ctx, a[1], -1))


def matrix_determinant(ctx: Ctx, A, used_rows: List[bool], used_cols: List[bool], dim: int, reduce):
    """Matrix<T>::Determinant (Matrix.cpp:227-263): Laplace expansion along the first unused row; `reduce` on every partial
    determinant of size >= 2; size 1 returns the entry itself."""
    d = len(A)
    row = used_rows.index(False)
    negative, det = False, None
    for col in range(d):
        if used_cols[col]:
            continue
        if dim == 1:
            return A[row][col]
        tmp = A[row][col]
        if negative:
            tmp = _cv_neg(ctx, tmp)
        negative = not negative
        used_rows[row] = used_cols[col] = True
        tmp2 = matrix_determinant(ctx, A, used_rows, used_cols, dim - 1, reduce)
        used_rows[row] = used_cols[col] = False
        tmp = _cv_mul(ctx, tmp, tmp2)
        det = tmp if det is None else _cv_add(ctx, det, tmp)
    return reduce(det) if reduce else det


def matrix_invert(ctx: Ctx, A, reduce):
    """Matrix<T>::Invert (Matrix.cpp:182-216): returns (adjugate, det)."""
    d = len(A)
    adj = [[None] * d for _ in range(d)]
    used_rows, used_cols = [False] * d, [False] * d
    for i in range(d):
        for j in range(d):
            used_rows[i] = used_cols[j] = True
            adj[j][i] = matrix_determinant(ctx, A, used_rows, used_cols, d - 1, reduce)
            used_rows[i] = used_cols[j] = False
            if (i + j) % 2 == 1:
                adj[j][i] = _cv_neg(ctx, adj[j][i])
    det = _cv_mul(ctx, A[0][0], adj[0][0])
    for i in range(1, d):
        det = _cv_add(ctx, det, _cv_mul(ctx, A[0][i], adj[i][0]))
    return adj, (reduce(det) if reduce else det)


def regress(ctx: Ctx, ksm, auto_ksms, ks: Sequence[int], X, y):
    """Regression::Regress (Regression.h:102-134) without the GenerateNoise masking (:136-148).  X: N rows of d unscaled
    ciphertexts (lists of parts), y: N unscaled ciphertexts.  Returns (theta parts list, det parts)."""
    N, d = len(X), len(X[0])
    P = lambda parts: ("parts", [list(q) for q in parts])

    def key_switch(c):
        return ("parts", apply_key_switch(ctx, ksm, c[1])) if c[0] == "tprod" else ("parts", apply_key_switch_parts(ctx, ksm, c[1]))

    def process(c):                                                   # processFunc (:112-115)
        c = key_switch(c)
        return ("parts", sum_batched_data(ctx, auto_ksms, ks, c[1]))

    # last = dataCopy^T * labels  (Matrix::operator*=(vector&), Matrix.cpp:81-98)
    last = []
    for i in range(d):
        acc = _cv_mul(ctx, P(X[0][i]), P(y[0]))
        for j in range(1, N):
            acc = _cv_add(ctx, acc, _cv_mul(ctx, P(X[j][i]), P(y[j])))
        last.append(acc)
    # MultByTranspose (Matrix.cpp:150-174)
    A = [[None] * d for _ in range(d)]
    for i in range(d):
        for j in range(i, d):
            acc = _cv_mul(ctx, P(X[0][i]), P(X[0][j]))
            for k in range(1, N):
                acc = _cv_add(ctx, acc, _cv_mul(ctx, P(X[k][i]), P(X[k][j])))
            A[i][j] = A[j][i] = acc
    last = [process(c) for c in last]
    done = {}
    for i in range(d):                                                # MapAll visits both copies of a symmetric entry; same value
        for j in range(d):
            key = (min(i, j), max(i, j))
            if key not in done:
                done[key] = process(A[i][j])
            A[i][j] = done[key]
    if d == 1:
        return [last[0][1]], A[0][0][1]
    adj, det = matrix_invert(ctx, A, key_switch)
    theta = []
    for i in range(d):                                                # dataCopy *= last (Matrix.cpp:57-79), then MapAll key switch
        acc = _cv_mul(ctx, adj[i][0], last[0])
        for k in range(1, d):
            acc = _cv_add(ctx, acc, _cv_mul(ctx, adj[i][k], last[k]))
        theta.append(key_switch(acc)[1])
    return theta, det[1]


# --------------------------------------------------------------------------------------------
# Wire format (Serialization.h:29-85, Serialization.cpp:3-119, FHEContext.cpp:45-81, FHE-SI.cpp:72-78,137-143,270-276), LP64:
# `unsigned`/uint32_t/int32_t = 4 bytes, `long` = 8 bytes, bool = 1 byte, all little endian (x86-64 raw struct writes).
# --------------------------------------------------------------------------------------------
import struct


def wire_zz(v: int) -> bytes:
    """Export(ofstream&, const ZZ&) (Serialization.cpp:3-13): uint32 NumBytes, bool neg, magnitude bytes little endian
    (NTL: NumBytes(0) == 0; BytesFromZZ writes |val|)."""
    mag = abs(v)
    nb = (mag.bit_length() + 7) // 8
    return struct.pack("<I?", nb, v < 0) + mag.to_bytes(nb, "little")


def wire_zzx(poly: Sequence[int]) -> bytes:
    """Export(ofstream&, const ZZX&) (Serialization.cpp:29-36): int32 degree (-1 for the zero polynomial), coefficients 0..deg."""
    deg = len(poly) - 1
    while deg >= 0 and poly[deg] == 0:
        deg -= 1
    return struct.pack("<i", deg) + b"".join(wire_zz(poly[i]) for i in range(deg + 1))


def wire_vec_long(row: Sequence[int]) -> bytes:
    """Export(ofstream&, const vec_long&) (Serialization.cpp:83-89)."""
    return struct.pack("<I", len(row)) + struct.pack("<%dq" % len(row), *row)


def wire_dcrt(a: dict) -> bytes:
    """Export(ofstream&, const DoubleCRT&) (Serialization.cpp:56-65): uint32 card, then (long index, vec_long row) ascending."""
    return struct.pack("<I", len(a)) + b"".join(struct.pack("<q", i) + wire_vec_long(a[i]) for i in sorted(a))


def wire_vector(items, item_fn) -> bytes:
    """Export(ofstream&, const vector<T>&) (Serialization.h:41-48)."""
    return struct.pack("<I", len(items)) + b"".join(item_fn(x) for x in items)


def wire_ciphertext(parts: Sequence[Sequence[int]]) -> bytes:
    """Export(ofstream&, const Ciphertext&) (Serialization.cpp:109-114) of an unscaled ciphertext: vector<CiphertextPart>."""
    return wire_vector(parts, wire_zzx)


def wire_key_switch(ksm) -> bytes:
    """KeySwitchSI::Export (FHE-SI.cpp:270-272): vector<vector<DoubleCRT>>."""
    return wire_vector(ksm, lambda row: wire_vector(row, wire_dcrt))


def wire_context(ctx: "Ctx", generator: int) -> bytes:
    """FHEcontext::ExportSIContext (FHEContext.cpp:45-60)."""
    out = struct.pack("<II", ctx.m, ctx.logQ) + wire_zz(ctx.p) + struct.pack("<II", generator, ctx.decomp_size)
    out += struct.pack("<I", ctx.L)
    for q, r in zip(ctx.primes, ctx.roots):
        out += struct.pack("<qq", q, r)
    return out


class WireReader:
    """Import side of the same format."""

    def __init__(self, data: bytes):
        self.d, self.o = data, 0

    def take(self, fmt: str):
        v = struct.unpack_from(fmt, self.d, self.o)
        self.o += struct.calcsize(fmt)
        return v if len(v) > 1 else v[0]

    def zz(self) -> int:
        nb, neg = self.take("<I?")
        v = int.from_bytes(self.d[self.o:self.o + nb], "little")
        self.o += nb
        return -v if neg else v

    def zzx(self, n: int = 0) -> List[int]:
        deg = self.take("<i")
        poly = [self.zz() for _ in range(deg + 1)]
        return poly + [0] * max(0, n - len(poly))

    def vec_long(self) -> List[int]:
        n = self.take("<I")
        v = list(struct.unpack_from("<%dq" % n, self.d, self.o))
        self.o += 8 * n
        return v

    def dcrt(self) -> dict:
        out = {}
        for _ in range(self.take("<I")):
            i = self.take("<q")
            out[i] = self.vec_long()
        return out

    def vector(self, item_fn):
        return [item_fn() for _ in range(self.take("<I"))]

    def done(self) -> bool:
        return self.o == len(self.d)


# --------------------------------------------------------------------------------------------
# BGV-style modulus switching (dead code in fhe-si -- no callers -- but part of the DoubleCRT surface)
# --------------------------------------------------------------------------------------------
def dcrt_add_primes_and_scale(ctx: Ctx, a: dict, s1: Sequence[int]) -> dict:
    """DoubleCRT::addPrimesAndScale (DoubleCRT.cpp:162-208): scale the existing rows by F * (F^-1 mod p), F = product of the
    added primes, and append zero rows for them."""
    assert not set(s1) & set(a)
    if not s1:
        return dict(a)
    factor = 1
    for i in s1:
        factor *= ctx.primes[i]
    factor *= pow(factor % ctx.p, -1, ctx.p)
    out = {i: [x * (factor % ctx.primes[i]) % ctx.primes[i] for x in row] for i, row in a.items()}
    for i in s1:
        out[i] = [0] * ctx.phim
    return out


def dcrt_scale_down_to_set(ctx: Ctx, a: dict, s: Sequence[int]) -> dict:
    """DoubleCRT::scaleDownToSet (DoubleCRT.cpp:518-558)."""
    keep = sorted(set(a) & set(s))
    diff = sorted(set(a) - set(s))
    assert keep and diff
    D = 1
    for i in diff:
        D *= ctx.primes[i]
    b = dcrt_op_scalar(ctx, a, D % ctx.p, "mul")                       # *this *= (diffProd % p)
    delta = dcrt_to_poly(ctx, b, idxset=diff)                          # toPoly(delta, diff): centred modulo D
    factor = D * pow(D % ctx.p, -1, ctx.p)
    delta = [d * factor - d for d in delta]
    mod = D * ctx.p
    delta = [(d % mod) - (mod if (d % mod) > mod // 2 else 0) for d in delta]     # ReduceCoefficientsSlow (Util.cpp:35-43)
    b = {i: b[i] for i in keep}                                        # removePrimes(diff)
    dD = dcrt_from_poly(ctx, delta, keep)
    b = dcrt_op(ctx, b, dD, "add")                                     # *this += delta
    return dcrt_div_scalar(ctx, b, D)                                  # *this /= diffProd
