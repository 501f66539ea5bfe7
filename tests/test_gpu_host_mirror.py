"""GPU: the C++ mirror of the reference's class surface (fhe-si_amd/host/*.h: FHEcontext / Cmodulus / DoubleCRT / Ciphertext /
FHESISecKey / FHESIPubKey / KeySwitchSI over the C ABI) running the reference's Test_AddMul sequence
(Test_AddMul.cpp:11-113), and bit-exact agreement of its ciphertexts with the committed fixture produced by the
independent Python model from the same documented PRNG stream."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tests", "host")        # the harness programs (tests/host/*.cpp) over the C++ mirror fhe-si_amd/host/*.h
EXE = os.path.join(HOST, "test_addmul")


def build():
    subprocess.check_call(["make", "-C", HOST, "-j8"], stdout=subprocess.DEVNULL)      # normally a no-op: __graft_entry__.build() built them


def test_addmul_sequence_readme_parameters():
    """README smoke parameters `80 23 7` (README:46-47): exit code = number of failed seeds."""
    build()
    r = subprocess.run([EXE, "80", "23", "7", "--tests=8"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "All tests SUCCEEDED!" in r.stdout


def test_addmul_power_of_two_ring():
    build()
    r = subprocess.run([EXE, "100", "257", "3", "--tests=3"], capture_output=True, text=True, timeout=600)     # m = 256
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize("args", [["1"], ["2"], ["5", "47", "5", "90"]])      # Test_General's own p = 2027, g = 3, logQ = 120 (seeds 1, 2); a small ring m = 46
def test_general_sequence(args):
    """Test_General.cpp:16-101 on the mirror: ciphertext x ciphertext + key switch, `+= constant`, `*= constant`, rotation + automorphism key
    switch, negation, and the final combination; every ciphertext decrypts to the plaintext-side value, and the device forms of
    Ciphertext::operator+=(ZZX) / operator*=(ZZX) equal the host forms that follow Ciphertext.cpp:29-36,147-156 literally."""
    build()
    r = subprocess.run([os.path.join(HOST, "test_general"), *args], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "All tests finished. Test SUCCEEDED" in r.stdout
    if len(args) == 1:
        assert "m=2026 phi(m)=1012 logQ=120" in r.stdout


def test_dump_matches_python_model_fixture():
    build()
    fx = [c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "ciphertext.json")))["mul_relin"] if c["m"] == 22][0]
    r = subprocess.run([EXE, str(fx["logQ"]), str(fx["p"]), "7", str(fx["seed"]), "--dump"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["t"] == fx["t"]
    assert [d["c1_0"], d["c1_1"]] == fx["c1"] and [d["c2_0"], d["c2_1"]] == fx["c2"]
    assert [d["res_0"], d["res_1"]] == fx["result"]


def test_modulus_switching_methods_match_python_model():
    """addPrimesAndScale / scaleDownToSet (DoubleCRT.cpp:162-208, 518-558) on the mirrored DoubleCRT."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import fhesi_pyref as R
    build()
    m, logQ, p = 64, 100, 23
    r = subprocess.run([os.path.join(HOST, "test_modswitch"), str(m)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    _, phim = R.zms_idx(m)
    primes = R.add_primes_by_size(m, R.si_context_size(logQ, p, phim))
    ctx = R.Ctx(m, logQ, p, primes)
    L = len(primes)
    assert d["L"] == L and L >= 3
    poly = [int(x) for x in d["poly"]]
    grown = R.dcrt_add_primes_and_scale(ctx, R.dcrt_from_poly(ctx, poly, [0, 1]), list(range(2, L)))
    assert {int(k): [int(x) for x in v] for k, v in d["grown"].items()} == grown
    scaled = R.dcrt_scale_down_to_set(ctx, R.dcrt_from_poly(ctx, poly), [0, 1])
    assert {int(k): [int(x) for x in v] for k, v in d["scaled"].items()} == scaled


REG = os.path.join(HOST, "test_regression")


@pytest.mark.parametrize("p,g,dim,rows,seed", [(23, 7, 1, 2, 5), (23, 7, 2, 2, 1), (23, 7, 3, 2, 2), (17, 3, 3, 3, 3), (257, 3, 2, 2, 4), (47, 5, 4, 1, 6)])
def test_regression_object_path_equals_batched_waves(p, g, dim, rows, seed):
    """Regression::Regress (Regression.h:102-149) three ways -- Matrix<Ciphertext> object at a time (the reference's control flow),
    device waves (RegressBatched), plaintext ring -- see tests/host/test_regression.cpp.  m = 22 and 46 run Bluestein rows,
    m = 16 and 256 the power-of-two NTT."""
    build()
    r = subprocess.run([REG, str(p), str(g), str(dim), str(rows), str(seed), "--at-once"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ciphertexts of both evaluators bit-identical: yes" in r.stdout
    assert "recorded and at-once ciphertexts bit-identical: yes" in r.stdout       # the literal control flow recorded (fhesi_engine.h) = run statement by statement
    assert r.stdout.count("decrypts to the plaintext regression: yes") == 2
    assert "Test SUCCEEDED" in r.stdout


def test_regression_literal_with_recording_off():
    """FHESI_EAGER=1: every statement of the literal control flow runs at once (no recording anywhere in the process); same ciphertexts as the waves."""
    build()
    r = subprocess.run([REG, "23", "7", "3", "2", "2"], capture_output=True, text=True, timeout=900, env=dict(os.environ, FHESI_EAGER="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "object at a time (at once)" in r.stdout and "ciphertexts of both evaluators bit-identical: yes" in r.stdout


@pytest.mark.parametrize("args", [["23", "7", "2", "2", "1"], ["47", "5", "3", "3", "2"], ["257", "3", "3", "2", "3"], ["8423", "7", "4", "2", "1"]])
def test_statistics_moments_and_covariance(args):
    """Statistics::ComputeNthMoment / ComputeCovariance (Statistics.h:48-133) on the mirrored classes (harness: tests/host/statistics_literal.h), driven as
    Test_Statistics.cpp:66-244 drives it (its logQ formula, SetUpSIContext(xi)): recorded and at-once evaluation give bit-identical
    ciphertexts, and mean, second moments, N, N^2 and the covariance matrix decrypt to the same statistics computed in the plaintext ring
    Z_p[X]/Phi_m.  m = 22 / 46 / 8422 are the reference's safe-prime rings, m = 256 a power of two."""
    build()
    r = subprocess.run([os.path.join(HOST, "test_statistics"), *args], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("decrypt to the plaintext statistics: yes") == 2
    assert "recorded and at-once ciphertexts bit-identical: yes" in r.stdout and "Test SUCCEEDED" in r.stdout


@pytest.mark.parametrize("args", [[], ["46", "90", "47", "5"], ["256", "130", "257", "3", "3"]])
def test_recorded_ciphertext_operations_equal_statements_run_at_once(args):
    """The mirror's Ciphertext records operator*= / += / *= long / >>= / ApplyKeySwitch on device-resident values and evaluates them in
    batches (fhe-si_amd/host/fhesi_engine.h); every flow of tests/host/test_lazy.cpp must give the ciphertexts the same statements give
    when each runs at once through the bodies that follow Ciphertext.cpp:123-275 and FHE-SI.cpp:241-260."""
    build()
    r = subprocess.run([os.path.join(HOST, "test_lazy"), *args], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Test SUCCEEDED" in r.stdout and "FAIL" not in r.stdout


def test_wire_format_bytes_match_python_model(tmp_path):
    """ExportSIContext / key Export / Ciphertext Export (FHEContext.cpp:45-60, FHE-SI.cpp:72-74,137-139,270-272, Serialization.cpp)
    written by the C++ mirror == the Python model's rendering of the same objects, byte for byte; then the import round trip."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import fhesi_pyref as R
    build()
    fx = [c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "ciphertext.json")))["mul_relin"] if c["m"] == 22][0]
    m, logQ, p, seed = fx["m"], fx["logQ"], fx["p"], fx["seed"]
    r = subprocess.run([os.path.join(HOST, "test_wire"), str(logQ), str(p), "7", str(seed), str(tmp_path)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "wire roundtrip ok" in r.stdout
    I = lambda v: [int(x) for x in v]
    primes, roots = I(fx["primes"]), I(fx["roots"])
    ctx = R.Ctx(m, logQ, p, primes, roots)
    rd = lambda name: open(os.path.join(str(tmp_path), name), "rb").read()
    assert rd("context.bin") == R.wire_context(ctx, 7)
    t, pk = R.keygen(ctx, R.SplitMix64(seed))
    assert [str(x) for x in t] == fx["t"]
    one = [1] + [0] * (ctx.phim - 1)
    assert rd("sk.bin") == R.wire_vector([R.dcrt_from_poly(ctx, one), R.dcrt_from_poly(ctx, t)], R.wire_dcrt)
    assert rd("pk.bin") == R.wire_vector([R.dcrt_from_poly(ctx, c) for c in pk], R.wire_dcrt)
    ksm = [[{i: I(d[i]) for i in range(ctx.L)} for d in fx["ksm"][r]] for r in range(2)]
    assert rd("ksk.bin") == R.wire_key_switch(ksm)
    assert rd("c1.bin") == R.wire_ciphertext([I(x) for x in fx["c1"]])
    assert rd("c2.bin") == R.wire_ciphertext([I(x) for x in fx["c2"]])
    assert rd("prod.bin") == R.wire_ciphertext([I(x) for x in fx["scaled"]])        # Export scales the product down first


@pytest.mark.parametrize("m", [64, 22, 1024])
def test_single_crt_mirror_class(m):
    """The mirrored SingleCRT class (fhe-si_amd/host/fhesi_doublecrt.h, SingleCRT.h:41-175): conversions, arithmetic and index-set handling
    against big-integer arithmetic on the host and against the mirrored DoubleCRT (tests/host/test_scrt.cpp)."""
    build()
    r = subprocess.run([os.path.join(HOST, "test_scrt"), str(m)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Test SUCCEEDED" in r.stdout


@pytest.mark.parametrize("args", [["--fuzz", "400", "64", "100", "23", "7", "1"], ["--fuzz", "400", "64", "100", "23", "7", "2"], ["--fuzz", "300", "46", "90", "47", "5", "4", "--devices=0,0,0"],
                                  ["--fuzz", "200", "2048", "200", "23", "7", "9"]])
def test_recorded_ciphertext_operations_fuzz(args):
    """Random statements (products + key switch, sums of products, +=, *= long, automorphism + its key switch, += constant, *= polynomial,
    copies, re-encryption, reads, changes of the evaluation threshold) over a pool of ciphertexts kept twice -- recorded on device values
    and run at once on host values: the two pools hold the same bits at every comparison (tests/host/test_lazy.cpp --fuzz)."""
    build()
    r = subprocess.run([os.path.join(HOST, "test_lazy"), *args], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 mismatches" in r.stdout and "Test SUCCEEDED" in r.stdout


@pytest.mark.parametrize("prog,args,expect", [("test_general", ["1"], "Test SUCCEEDED"), ("test_addmul", ["80", "23", "7", "--tests=3"], "All tests SUCCEEDED!"),
                                              ("test_statistics", ["47", "5", "3", "3", "2"], "Test SUCCEEDED"), ("test_wire", ["100", "23", "7", "1", "TMP"], "wire roundtrip ok")])
def test_host_programs_with_recording_off(prog, args, expect, tmp_path):
    """FHESI_EAGER=1: every Ciphertext statement of the drivers runs at once (the bodies that follow Ciphertext.cpp / FHE-SI.cpp statement by
    statement) -- the checker form of the mirror must keep passing the drivers on its own, not only as the other side of a comparison."""
    build()
    argv = [str(tmp_path) if a == "TMP" else a for a in args]
    r = subprocess.run([os.path.join(HOST, prog), *argv], capture_output=True, text=True, timeout=900, env=dict(os.environ, FHESI_EAGER="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert expect in r.stdout
