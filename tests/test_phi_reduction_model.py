"""CPU models of the two pieces of host / device logic round 6 added for generic m (no GPU, no library):
(1) Phi_m and Psi_m = (X^m - 1) / Phi_m built from binomials X^e - 1 only -- products, and exact divisions in linear time by the
    recurrence Q[i] = Q[i - e] - A[i] (hostmath.cpp: hm::cyclotomic, hm::cyclotomic_cofactor) -- against the independent Python
    restatement of NumbTh.cpp:142-158 (fhesi_pyref.cyclotomic: dense product and dense exact division);
(2) the reduction modulo Phi_m as two products (bluestein.hip: blue_conv_*):  f Psi = Q (X^m - 1) + r Psi with deg(r Psi) < m, so the
    quotient Q of f by Phi_m is the coefficients m .. of f Psi, and r = (f - Q Phi) mod X^phi(m) -- against a schoolbook remainder, with
    the kernels' coefficients reduced modulo q first (every integer of the products non-negative and below m q^2, the bound the
    three auxiliary primes of the device convolution are sized for)."""
import random

import pytest

import fhesi_pyref as R


def binomial_mul(t, e):                      # t * (X^e - 1)
    r = [0] * (len(t) + e)
    for i, v in enumerate(t):
        r[i + e] += v
        r[i] -= v
    return r


def binomial_div(t, e):                      # t / (X^e - 1), exact
    q = [0] * (len(t) - e)
    for i in range(len(q)):
        q[i] = (q[i - e] if i >= e else 0) - t[i]
    # exactness: the recurrence never looks at the top e coefficients of t; they must agree with the quotient found
    assert all((q[i - e] if 0 <= i - e < len(q) else 0) - (q[i] if i < len(q) else 0) == t[i] for i in range(len(t))), "not divisible"
    return q


def phi_and_psi(m):
    """hm::cyclotomic and hm::cyclotomic_cofactor"""
    plus = [m // d for d in range(1, m + 1) if m % d == 0 and R.mobius(d) == 1]
    minus = [m // d for d in range(1, m + 1) if m % d == 0 and R.mobius(d) == -1]
    phi = [1]
    for e in plus:
        phi = binomial_mul(phi, e)
    for e in minus:
        phi = binomial_div(phi, e)
    psi = [1]
    for e in minus:
        psi = binomial_mul(psi, e)
    for e in plus:
        if e != m:                            # d = 1 is X^m - 1 itself
            psi = binomial_div(psi, e)
    return phi, psi


def poly_mul(a, b):
    r = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                r[i + j] += x * y
    return r


@pytest.mark.parametrize("m", list(range(2, 200)) + [210, 255, 256, 360, 385, 420, 1001, 1155, 2310, 4620])
def test_binomial_cyclotomic_equals_the_dense_restatement(m):
    phi, psi = phi_and_psi(m)
    assert phi == R.cyclotomic(m)
    n = R.zms_idx(m)[1]
    assert len(phi) - 1 == n and len(psi) - 1 == m - n
    prod = poly_mul(phi, psi)
    assert prod == [-1] + [0] * (m - 1) + [1]                    # Phi_m Psi_m = X^m - 1


@pytest.mark.parametrize("m", [9, 12, 15, 36, 45, 105, 360, 1155])
def test_remainder_by_two_products_equals_the_schoolbook_remainder(m):
    rng = random.Random(m)
    phi, psi = phi_and_psi(m)
    n, t = len(phi) - 1, len(psi) - 1
    for q in (97, (1 << 59) + 55, (1 << 60) - 93):              # (the identities hold modulo any q: Phi_m is monic)
        phi_q = [c % q for c in phi]                              # what the device tables hold
        psi_q = [c % q for c in psi]
        for _ in range(3):
            f = [rng.randrange(q) for _ in range(m)]
            f[rng.randrange(m)] = q - 1
            fp = poly_mul(f, psi_q)
            assert max(fp) < m * q * q and len(fp) <= 2 * m - 1    # no wrap in a transform of N >= 2m - 1 points, inside three 60-bit primes
            Q = [c % q for c in fp[m:m + t]] + [0] * max(0, t - (len(fp) - m))
            qp = poly_mul(Q, phi_q)
            assert max(qp) < m * q * q
            r = [(f[j] - qp[j]) % q for j in range(n)]
            # schoolbook: divide by the monic Phi_m from the top
            g = f[:]
            for k in range(m - 1, n - 1, -1):
                c = g[k]
                if c:
                    for j in range(n + 1):
                        g[k - n + j] = (g[k - n + j] - c * phi[j]) % q
            assert r == g[:n]
