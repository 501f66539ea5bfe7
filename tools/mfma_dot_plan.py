#!/usr/bin/env python3
"""The arithmetic of DESIGN.md section 8 (1), checked in numpy before any kernel exists: the key-switch dot product of one coefficient
position modulo one auxiliary prime,   out[ct][c] = sum_k D[ct][k] * K[k][c]  mod p   (D, K below p < 2^30; c = (limb, row)),
as ONE signed-int8 matrix product with int32 accumulation -- the shape `v_mfma_i32_32x32x32_i8` computes -- plus a lane-local recombination.

  D = sum_i (D'_i + 128) 256^i   with D'_i = byte_i(D) - 128 in [-128, 127]           (one xor 0x80808080 per word at run time)
  K = sum_j kappa_j 256^j        with balanced bytes kappa_j in [-128, 127]            (precomputed at key upload; kappa_3 <= 64)
  C'[(i, ct)][(j, c)] = sum_k D'_i[ct][k] kappa_j[k][c]                                (|C'| <= 66 * 128 * 128 < 2^21: int32 is exact)
  s[(j, c)]           = sum_k kappa_j[k][c]                                            (precomputed with the key table)
  out[ct][c] = sum_{i,j} (C'[(i, ct)][(j, c)] + 128 s[(j, c)]) 256^(i+j)  mod p

Rows (i, ct) = 4 x 8 = 32 and one 32-column block per byte plane j of the key put the 16 partial sums of an output into 16 accumulator
registers of one lane (profiles/r02_mfma_i8_layout.txt).  Run: python tools/mfma_dot_plan.py"""
import numpy as np


def balanced_bytes(x, planes=4):
    """x >= 0 (object ints or int64) -> planes arrays of int8-range digits with x = sum_j kappa_j 256^j."""
    x = np.asarray(x, dtype=np.int64).copy()
    out = []
    for j in range(planes):
        b = x & 255
        if j < planes - 1:
            b = np.where(b >= 128, b - 256, b)
        out.append(b)
        x = (x - b) >> 8
    assert np.all(x == 0)
    return out


def main():
    rng = np.random.default_rng(7)
    p, ncol, CT, NC = (1 << 30) - (1 << 15) * 3 + 1, 66, 8, 30          # any modulus below 2^30; 66 columns, 8 ciphertexts, 15 limbs x 2 rows
    for trial in range(50):
        D = rng.integers(0, p, size=(CT, ncol), dtype=np.int64)
        K = rng.integers(0, p, size=(ncol, NC), dtype=np.int64)
        if trial == 0:
            D[:] = p - 1; K[:] = p - 1                                     # extreme values
        if trial == 1:
            D[:] = 0x80808080 % p; K[:] = 0x7f7f7f7f % p                   # byte edges
        want = np.array([[sum(int(D[ct, k]) * int(K[k, c]) for k in range(ncol)) % p for c in range(NC)] for ct in range(CT)])
        Dp = [((D >> (8 * i)) & 255) - 128 for i in range(4)]             # D'_i
        kap = balanced_bytes(K)
        assert all(np.all((a >= -128) & (a <= 127)) for a in Dp + kap) and np.all(kap[3] <= 64)
        A = np.concatenate(Dp, axis=0).astype(np.int8)                     # rows (i, ct): 32 x 66
        B = np.concatenate(kap, axis=1).astype(np.int8)                    # columns (j, c): 66 x 120
        C = A.astype(np.int32) @ B.astype(np.int32)                        # what the matrix core accumulates
        assert np.abs(C).max() <= ncol * 128 * 128 < 1 << 21
        s = B.astype(np.int32).sum(axis=0)                                 # column sums (precomputed)
        got = np.zeros((CT, NC), dtype=object)
        for ct in range(CT):
            for c in range(NC):
                t = [0] * 7                                                # diagonals i + j
                for i in range(4):
                    for j in range(4):
                        t[i + j] += int(C[i * CT + ct, j * NC + c]) + 128 * int(s[j * NC + c])
                assert all(abs(v) < 1 << 25 for v in t)                    # 4 terms below 2^23 each
                got[ct, c] = sum(v << (8 * d) for d, v in enumerate(t)) % p
        assert np.array_equal(got.astype(np.int64), want), trial
    print("ok: 50 random / edge tiles, int32 accumulators exact, diagonals below 2^25")


if __name__ == "__main__":
    main()
