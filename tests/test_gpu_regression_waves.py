"""GPU parity: fhe_si_amd.regression.regress_waves with the device backend (C ABI: fhesi_ct_mul_sum_relin_dev,
fhesi_ct_automorph_key_switch_dev, fhesi_ct_add_dev, fhesi_ct_gather_dev, fhesi_ct_mul_long_dev) against the literal
control flow of Matrix.cpp / Regression.h evaluated by the Python model on the same valid keys and ciphertexts.  Bit-exact."""
import numpy as np
import pytest

import fhe_si_amd as F
import fhesi_pyref as R
import oracle_lib as O
import wave_backends as WB
from fhe_si_amd import regression as G

pytestmark = pytest.mark.gpu


def ksm_rows(ksm, L):
    return np.array([[[d[i] for i in range(L)] for d in ksm[r]] for r in range(2)], dtype=np.uint64)


@pytest.mark.parametrize("d,N", [(2, 2), (3, 1), (4, 1)])
def test_device_waves_equal_literal_control_flow(d, N):
    import torch
    case = WB.regression_case(d=d, N=N, seed=90 + d)
    c = case["ctx"]
    L, nd, nl, n = c.L, c.ndigits, (c.logQ + 63) // 64, c.phim
    ctx = F.Context(c.m, case["primes"], case["roots"])
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm_rows(case["ksm"], L))
    auto = [F.KeySwitchMatrix(ctx, 2, nd).upload(ksm_rows(a, L)) for a in case["auto"]]
    pool = G.ShardedPool(2 * n * nl, 4096, device="cuda:0")
    be = G.DeviceBackend(ctx, c.logQ, c.p, ksk, auto, case["ks"], pool)
    cts = np.stack([np.stack([O.ints_to_limbs(part, nl) for part in ct]) for ct in [case["X"][i][j] for i in range(N) for j in range(d)] + case["y"]])
    first = be.upload(cts)
    X = [[first + i * d + j for j in range(d)] for i in range(N)]
    y = [first + N * d + i for i in range(N)]
    theta, det, stats = G.regress_waves(be, X, y)
    got_det = [O.limbs_to_ints(x) for x in be.download(det)]
    got_theta = [[O.limbs_to_ints(x) for x in be.download(i)] for i in theta]
    theta_ref, det_ref = R.regress(c, case["ksm"], case["auto"], case["ks"], case["X"], case["y"])
    assert got_det == det_ref
    assert got_theta == theta_ref
    assert stats["automorph_key_switches"] == (d + d * (d + 1) // 2) * len(case["ks"])
    torch.cuda.synchronize()
