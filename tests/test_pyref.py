"""CPU: self-consistency of the Python big-int restatement and the reference's Test_AddMul sequence (Test_AddMul.cpp:18-86)."""
import random

import pytest

import fhesi_pyref as R


@pytest.mark.parametrize("m", [9, 15, 16, 22, 32, 46])
def test_bluestein_equals_tdft_and_inverts(m):
    primes = R.add_primes_by_size(m, 100.0)[:2]
    rng = random.Random(m)
    idx, phim = R.zms_idx(m)
    for q in primes:
        root = R.find_root_2m(q, m)
        a = [rng.randrange(q) for _ in range(m)]
        assert R.bluestein_fft(a, m, root, q) == R.tdft(a, m, root * root % q, q)
        x = [rng.randrange(-q * q, q * q) for _ in range(phim)]
        y = R.cmod_fft(x, m, q, root)
        assert R.cmod_ifft(y, m, q, root) == [c % q for c in x]
        if m & (m - 1) == 0:
            assert y == R.negacyclic_ntt_direct([c % q for c in x], m // 2, q, root * root % q)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_addmul_sequence(seed):
    """add; 7-fold add; mul+KS; square+KS; 9-fold add + KS, mul, KS -- success predicate of Test_AddMul.cpp:84-86."""
    m, logQ, p = 22, 80, 23
    _, phim = R.zms_idx(m)
    primes = R.add_primes_by_size(m, R.si_context_size(logQ, p, phim))
    ctx = R.Ctx(m, logQ, p, primes)
    rng = R.SplitMix64(seed)
    t, pk = R.keygen(ctx, rng)
    m1 = [rng.bnd(p) for _ in range(phim)]
    m2 = [rng.bnd(p) for _ in range(phim)]
    c1, c2 = R.encrypt(ctx, pk, m1, rng), R.encrypt(ctx, pk, m2, rng)
    ksm = R.key_switch_init_s2(ctx, t, rng)
    red = lambda parts: [[R.reduce_logq(c, logQ) for c in part] for part in parts]
    add = lambda a, b: red([[x + y for x, y in zip(pa, pb)] for pa, pb in zip(a, b)])
    mulp = lambda a, b: [c % p for c in R.poly_mul_mod_phi(ctx, a, b)]
    assert R.decrypt(ctx, t, add(c1, c2)) == [(a + b) % p for a, b in zip(m1, m2)]
    s7 = c2
    for _ in range(6):
        s7 = add(s7, c2)
    assert R.decrypt(ctx, t, s7) == [7 * b % p for b in m2]
    prod = R.ct_mul_relin(ctx, ksm, c1, c2)
    e_prod = mulp(m1, m2)
    assert R.decrypt(ctx, t, prod) == e_prod
    # square in scaled-up form, 9-fold accumulation before the key switch (Ciphertext.cpp:135-142 is linear)
    tp = R.ct_mul(ctx, prod, prod)
    prod2 = R.apply_key_switch(ctx, ksm, tp)
    e_prod2 = mulp(e_prod, e_prod)
    assert R.decrypt(ctx, t, prod2) == e_prod2
    acc = tp
    for _ in range(8):
        acc = [R.dcrt_op(ctx, x, y, "add") for x, y in zip(acc, tp)]
    nine = R.apply_key_switch(ctx, ksm, acc)
    quad = R.ct_mul_relin(ctx, ksm, nine, prod2)
    assert R.decrypt(ctx, t, quad) == [9 * c % p for c in mulp(e_prod2, e_prod2)]


def test_wire_format_round_trip():
    """Serialization.cpp:3-119 restated: what wire_* writes, WireReader reads back; known byte strings of the small cases."""
    import random
    rng = random.Random(1)
    assert R.wire_zz(0) == bytes.fromhex("0000000000")                     # NumBytes(0) = 0, not negative
    assert R.wire_zz(-258) == bytes.fromhex("02000000010201")              # 2 bytes, negative, magnitude little endian
    assert R.wire_zzx([0, 0]) == bytes.fromhex("ffffffff")                 # zero polynomial: degree -1
    poly = [rng.randrange(-(1 << 90), 1 << 90) for _ in range(10)] + [0, 0]
    rd = R.WireReader(R.wire_zzx(poly))
    assert rd.zzx(12) == poly and rd.done()
    d = {0: [1, 2, 3], 2: [4, 5, (1 << 60) - 1]}
    rd = R.WireReader(R.wire_key_switch([[d, d], [d]]))
    assert rd.vector(lambda: rd.vector(rd.dcrt)) == [[d, d], [d]] and rd.done()
    assert R.wire_vec_long([7]) == bytes.fromhex("01000000" + "0700000000000000")
