"""CPU (no GPU): the host logic of the recording evaluator behind the C++ mirror's Ciphertext (fhe-si_amd/host/fhesi_engine.h) -- arena
runs handed out / freed / reused / grown, sharing of equal operations, dependency levelling, evaluation triggers, lifetimes of values and
keys -- on random operation graphs against a direct evaluation, with a host-memory stand-in for the C ABI (tests/host/mock_abi.cpp: toy
arithmetic, test infrastructure only) and under AddressSanitizer + UndefinedBehaviorSanitizer."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tests", "host")


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_engine_host_logic_under_sanitizers(seed):
    subprocess.check_call(["make", "-C", HOST, "test_engine_cpu"], stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(HOST, "test_engine_cpu"), "3000", str(seed)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 mismatches" in r.stdout and "Test SUCCEEDED" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


def test_copy_pool_under_thread_sanitizer():
    """The copy threads of the host-buffer pipeline (fhe-si_amd/csrc/copy_pool.h: parallel memcpy between the caller's pageable buffers and the
    pinned ring, restartable with another thread count) under ThreadSanitizer; every copy compared with its source."""
    subprocess.check_call(["make", "-C", HOST, "test_copy_pool"], stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(HOST, "test_copy_pool"), "60", "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 mismatches" in r.stdout and "Test SUCCEEDED" in r.stdout
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-2000:]
