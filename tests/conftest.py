import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _poisoned_workspaces(request, monkeypatch):
    """Every in-process GPU test runs with FHESI_WS_POISON=1: the library fills each workspace it reserves with 0xA5 bytes before use, so a
    kernel that relies on what an earlier call left there (zeros, mostly) fails parity instead of passing by luck.  (Not the multi-rank
    module: its subprocesses inherit the environment, and their bench lines are read for rates.)"""
    if request.node.get_closest_marker("gpu") and request.module.__name__ != "test_gpu_multirank":
        monkeypatch.setenv("FHESI_WS_POISON", "1")
