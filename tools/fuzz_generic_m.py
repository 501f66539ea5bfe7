#!/usr/bin/env python3
"""Random generic m (not a power of two, a prime or twice a prime): the reduction modulo Phi_m by two convolutions (bluestein.hip, forced by
FHESI_PHI_CONV=1 below m = 16384) against the long division in LDS and, for every fifth m and for m above 16384, against the oracle.
usage: tools/fuzz_generic_m.py <cases> <seed>      (GPU box)"""
import os
import sys

R_ = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for q in (R_, R_ + "/oracle", R_ + "/tests"):
    sys.path.insert(0, q)
import numpy as np

import fhe_si_amd as F
import fhesi_pyref as R
import oracle_lib as O
import params as P

cases, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)


def generic(m):
    if m & (m - 1) == 0 or R.is_prime(m) or (m % 2 == 0 and (m // 2) % 2 == 1 and R.is_prime(m // 2)):
        return False
    return True


bad = 0
done = 0
while done < cases:
    big = done % 10 == 9
    m = int(rng.integers(16385, 70000)) if big else int(rng.integers(6, 16384))
    if not generic(m):
        continue
    primes, roots = P.first_primes(m, 2)
    os.environ["FHESI_PHI_CONV"] = "1"
    ctx = F.Context(m, primes, roots)
    ev = P.rand_rows(rng, primes, ctx.phim, 2)
    buf = ctx.upload(ev)
    ctx.rows_ntt_inv(buf, 2)
    got = buf.download(ev.shape)
    ok = True
    if not big:
        del os.environ["FHESI_PHI_CONV"]
        ctx2 = F.Context(m, primes, roots)
        b2 = ctx2.upload(ev)
        ctx2.rows_ntt_inv(b2, 2)
        ok = np.array_equal(b2.download(ev.shape), got)
    if big or done % 5 == 0:
        orc = O.Oracle(m, primes, roots)
        if m > 2000:
            orc.set_bluestein_fft(True)
        ok = ok and np.array_equal(got[1, 0], orc.cmod_ifft(0, ev[1, 0]))
    ctx.rows_ntt_fwd(buf, 2)
    ok = ok and np.array_equal(buf.download(ev.shape), ev)
    if not ok:
        bad += 1
        print("MISMATCH m =", m, flush=True)
    done += 1
print(f"{cases} generic m, {bad} mismatches")
