"""fhe-si_amd: MI355X (gfx950) backend for fhe-si's DoubleCRT hot path.

The product is the C-ABI shared library ``csrc/libfhesi_hip.so`` (declared in ``include/fhesi_hip.h``); this
package is the thin ctypes binding used by tests/ and bench.py, plus the C++ mirror of the reference's class
surface under ``host/``.  The directory name carries a hyphen (it follows the reference's name), so import it
through the repo-root shim:  ``import fhe_si_amd``.
"""
from .binding import (Backend, Context, DoubleCRT, SingleCRT, dcrt_assign_scrt, KeySwitchMatrix, Comm, FhesiError, build_library, library_path,
                      OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_SET)

__all__ = ["Backend", "Context", "DoubleCRT", "SingleCRT", "dcrt_assign_scrt", "KeySwitchMatrix", "Comm", "FhesiError", "build_library", "library_path",
           "OP_ADD", "OP_SUB", "OP_MUL", "OP_DIV", "OP_SET"]
