"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/fhesi_hip.h declares.
No compute call is made here (there is no GPU and the product path has no CPU fallback)."""
import ctypes
import os
import re

import pytest

import fhe_si_amd as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "fhesi_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fhesi_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_the_header():
    so = F.build_library()
    assert os.path.exists(so)
    lib = ctypes.CDLL(so)
    names = header_functions()
    assert len(names) >= 45
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # the Python binding covers exactly the header
    assert sorted(F.binding.ABI_SYMBOLS) == names


def test_no_oracle_in_the_product_path():
    """The product sources never reference oracle/ (a CPU fallback would void every parity claim)."""
    for base in ("fhe-si_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".cpp", ".h", ".inc", ".hpp")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "fhesi_oracle" not in text and "oracle_lib" not in text and "fhesi_pyref" not in text, os.path.join(dirpath, f)


def test_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(F.FhesiError):
        F.Context(32, [1152921504606844417], [3])
