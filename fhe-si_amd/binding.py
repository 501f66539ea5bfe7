"""ctypes binding of include/fhesi_hip.h (the C ABI of the HIP library).

No compute happens in Python and there is no CPU fallback: if the shared library is missing or a HIP call fails,
the binding raises FhesiError with the library's message.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_DIR, "csrc")
_SO = os.environ.get("FHESI_LIB") or os.path.join(_CSRC, "libfhesi_hip.so")     # FHESI_LIB: dev override (ablation builds)

OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_SET = 0, 1, 2, 3, 4

# every symbol include/fhesi_hip.h declares (tests/test_abi.py checks the library exports all of them)
ABI_SYMBOLS = [
    "fhesi_last_error", "fhesi_device_count", "fhesi_ctx_create", "fhesi_ctx_destroy", "fhesi_ctx_m", "fhesi_ctx_phim",
    "fhesi_ctx_nprimes", "fhesi_ctx_prime", "fhesi_ctx_zms_idx", "fhesi_ctx_phi_m", "fhesi_ctx_sync", "fhesi_ctx_stream",
    "fhesi_timer_start", "fhesi_timer_stop", "fhesi_cmod_fft", "fhesi_cmod_ifft", "fhesi_dcrt_alloc", "fhesi_dcrt_free",
    "fhesi_dcrt_copy", "fhesi_dcrt_index_set", "fhesi_dcrt_equal", "fhesi_dcrt_upload_row", "fhesi_dcrt_download_row",
    "fhesi_dcrt_device_ptr", "fhesi_dcrt_from_poly", "fhesi_dcrt_to_poly", "fhesi_dcrt_op", "fhesi_dcrt_op_scalar",
    "fhesi_dcrt_automorph", "fhesi_dcrt_add_primes", "fhesi_dcrt_remove_primes", "fhesi_dcrt_from_scrt", "fhesi_dcrt_to_scrt",
    "fhesi_rows_ntt_fwd_dev", "fhesi_rows_ntt_inv_dev", "fhesi_rows_op_dev", "fhesi_ksk_create", "fhesi_ksk_free",
    "fhesi_ksk_upload", "fhesi_ksk_device_ptr", "fhesi_ksk_bytes", "fhesi_ct_mul_relin_batch", "fhesi_ct_mul_relin_batch_dev",
    "fhesi_ct_mul_dev", "fhesi_apply_key_switch_dev", "fhesi_dev_alloc", "fhesi_dev_free", "fhesi_dev_upload", "fhesi_dev_download",
    "fhesi_dev_copy", "fhesi_prof_enable", "fhesi_prof_read",
    "fhesi_ct_add_dev", "fhesi_ct_mul_long_dev", "fhesi_rows_mul_long_dev", "fhesi_ct_automorph_dev", "fhesi_ct_automorph_key_switch_dev",
    "fhesi_ct_gather_dev", "fhesi_ct_mul_sum_relin_dev", "fhesi_encrypt_batch", "fhesi_decrypt_batch", "fhesi_dcrt_exp", "fhesi_selftest_aux32",
    "fhesi_ctx_set_option", "fhesi_ctx_get_option", "fhesi_ctx_copy_options", "fhesi_host_alloc", "fhesi_host_free", "fhesi_ksk_key_bits", "fhesi_prof_kernel_name", "fhesi_ksk_mark_dirty", "fhesi_ksk_upload_dev",
    "fhesi_dcrt_add_primes_and_scale", "fhesi_dcrt_scale_down_to_set",
    "fhesi_keyswitch_init_batch", "fhesi_ksk_download", "fhesi_comm_init_all", "fhesi_comm_from_rccl", "fhesi_comm_destroy", "fhesi_comm_rank",
    "fhesi_comm_size", "fhesi_ksk_broadcast", "fhesi_comm_broadcast_dev", "fhesi_comm_exchange", "fhesi_comm_exchange_begin", "fhesi_comm_exchange_end", "fhesi_comm_allreduce_rows", "fhesi_scrt_alloc", "fhesi_scrt_from_poly", "fhesi_scrt_to_poly", "fhesi_scrt_op_scalar", "fhesi_dcrt_assign_scrt", "fhesi_scrt_assign_dcrt",
    "fhesi_ksk_form", "fhesi_ct_add_const_dev", "fhesi_ct_mul_poly_dev",
    "fhesi_encrypt_batch_seeded", "fhesi_keyswitch_init_batch_seeded", "fhesi_dcrt_sample",
    "fhesi_abi_version", "fhesi_host_stage_release",
]
ABI_VERSION = 7          # FHESI_ABI_VERSION of the include/fhesi_hip.h this table was written against (checked in _load)
PROF_CLASSES = {"ntt_fwd": 0, "ntt_inv": 1, "rns_reduce": 2, "tensor": 3, "crt": 4, "digits": 5, "dot": 6, "ew": 7, "ntt_fwd_digits_main": 8}


class FhesiError(RuntimeError):
    pass


def library_path() -> str:
    return _SO


def build_library(force: bool = False) -> str:
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".cpp", ".h", ".inc"))]
    srcs.append(os.path.join(os.path.dirname(_DIR), "include", "fhesi_hip.h"))
    stale = force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _CSRC, "-j8"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None
_vp, _i32, _i64, _u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64


def _load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise FhesiError(f"HIP extension not built: {_SO} is missing (run __graft_entry__.build())")
    # One HIP runtime per process: PyTorch ships its own libamdhip64, the library links /opt/rocm's.  Whichever is loaded first
    # serves both (same soname); loading ours first and torch afterwards leaves torch without a device ("No HIP GPUs are
    # available").  So when torch is installed it is imported first -- it is only ever used for pool memory and collectives
    # (fhe-si_amd/regression.py, bench.py), never for compute.  FHESI_NO_TORCH_PRELOAD=1 skips this.
    if "torch" not in sys.modules and not os.environ.get("FHESI_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    lib = C.CDLL(_SO)
    lib.fhesi_last_error.restype = C.c_char_p
    # a library of another ABI revision would accept this table's calls and shift their arguments: refuse it
    have = lib.fhesi_abi_version() if hasattr(lib, "fhesi_abi_version") else None
    if have != ABI_VERSION:
        raise FhesiError(f"{_SO} has ABI revision {have}, this binding was written against {ABI_VERSION}: rebuild (make -C fhe-si_amd/csrc)")
    sig = {
        "fhesi_device_count": [_vp],
        "fhesi_ctx_create": [_vp, _i64, _i32, _vp, _vp, _i32],
        "fhesi_ctx_destroy": [_vp],
        "fhesi_ctx_prime": [_vp, _i32, _vp, _vp],
        "fhesi_ctx_zms_idx": [_vp, _vp],
        "fhesi_ctx_phi_m": [_vp, _vp],
        "fhesi_ctx_sync": [_vp],
        "fhesi_timer_start": [_vp],
        "fhesi_timer_stop": [_vp, _vp],
        "fhesi_cmod_fft": [_vp, _i32, _vp, _i32, _i64, _vp],
        "fhesi_cmod_ifft": [_vp, _i32, _vp, _vp],
        "fhesi_dcrt_alloc": [_vp, _vp, _i32, _vp],
        "fhesi_dcrt_free": [_vp],
        "fhesi_dcrt_copy": [_vp, _vp],
        "fhesi_dcrt_index_set": [_vp, _vp, _vp],
        "fhesi_dcrt_equal": [_vp, _vp, _vp],
        "fhesi_dcrt_upload_row": [_vp, _i32, _vp],
        "fhesi_dcrt_download_row": [_vp, _i32, _vp],
        "fhesi_dcrt_from_poly": [_vp, _vp, _i32, _i64],
        "fhesi_dcrt_to_poly": [_vp, _vp, _i32, _i32, _vp, _i32],
        "fhesi_dcrt_op": [_vp, _vp, _i32],
        "fhesi_dcrt_op_scalar": [_vp, _vp, _i32, _i32],
        "fhesi_dcrt_automorph": [_vp, _i64],
        "fhesi_dcrt_exp": [_vp, _i64],
        "fhesi_selftest_aux32": [_vp],
        "fhesi_dcrt_add_primes": [_vp, _vp, _i32],
        "fhesi_dcrt_remove_primes": [_vp, _vp, _i32],
        "fhesi_dcrt_from_scrt": [_vp, _vp],
        "fhesi_dcrt_to_scrt": [_vp, _vp],
        "fhesi_rows_ntt_fwd_dev": [_vp, _vp, _i64],
        "fhesi_rows_ntt_inv_dev": [_vp, _vp, _i64],
        "fhesi_rows_op_dev": [_vp, _vp, _vp, _i64, _i32],
        "fhesi_ksk_create": [_vp, _i32, _i32, _vp],
        "fhesi_ksk_free": [_vp],
        "fhesi_ksk_upload": [_vp, _vp],
        "fhesi_ct_mul_relin_batch": [_vp, _vp, _i32, _u64, _i32, _vp, _vp, _vp, _i32, _i64],
        "fhesi_ct_mul_relin_batch_dev": [_vp, _vp, _i32, _u64, _i32, _vp, _vp, _vp, _i32, _i64],
        "fhesi_ct_mul_dev": [_vp, _u64, _vp, _vp, _i32, _i64, _vp],
        "fhesi_apply_key_switch_dev": [_vp, _vp, _i32, _i32, _vp, _i64, _vp, _i32],
        "fhesi_ct_add_dev": [_vp, _i32, _vp, _vp, _i32, _i32, _i64],
        "fhesi_ct_mul_long_dev": [_vp, _i32, _vp, _i64, _i32, _i32, _i64],
        "fhesi_rows_mul_long_dev": [_vp, _vp, _i64, _i64],
        "fhesi_ct_automorph_dev": [_vp, _i64, _vp, _i32, _i32, _i64, _vp, _i32],
        "fhesi_ct_automorph_key_switch_dev": [_vp, _vp, _i32, _i32, _i64, _vp, _i32, _i64, _vp, _i32],
        "fhesi_ct_gather_dev": [_vp, _vp, _vp, _i64, _i64, _vp],
        "fhesi_ct_mul_sum_relin_dev": [_vp, _vp, _i32, _u64, _i32, _vp, _i32, _vp, _vp, _vp, _i64, _vp],
        "fhesi_encrypt_batch": [_vp, _vp, _vp, _i32, _u64, _vp, _vp, _i64, _vp, _i32],
        "fhesi_decrypt_batch": [_vp, _vp, _i32, _u64, _vp, _i32, _i64, _vp],
        "fhesi_dev_alloc": [_vp, C.c_size_t, _vp],
        "fhesi_dev_free": [_vp, _vp],
        "fhesi_dev_upload": [_vp, _vp, _vp, C.c_size_t],
        "fhesi_dev_download": [_vp, _vp, _vp, C.c_size_t],
        "fhesi_dev_copy": [_vp, _vp, _vp, C.c_size_t],
        "fhesi_prof_enable": [_vp, _i32],
        "fhesi_prof_read": [_vp, _i32, _vp, _vp, _vp],
        "fhesi_prof_kernel_name": [_vp, _i32, _vp, C.c_size_t],
        "fhesi_ctx_set_option": [_vp, C.c_char_p, _i64],
        "fhesi_ctx_get_option": [_vp, C.c_char_p, _vp],
        "fhesi_ctx_copy_options": [_vp, _vp],
        "fhesi_host_alloc": [_vp, C.c_size_t, _vp],
        "fhesi_host_free": [_vp, _vp],
        "fhesi_host_stage_release": [_vp],
        "fhesi_ksk_key_bits": [_vp, _vp, _vp],
        "fhesi_ksk_mark_dirty": [_vp],
        "fhesi_comm_init_all": [_i32, _vp, _vp],
        "fhesi_comm_from_rccl": [_vp, _vp],
        "fhesi_comm_destroy": [_vp],
        "fhesi_ksk_broadcast": [_vp, _vp, _i32],
        "fhesi_comm_broadcast_dev": [_vp, _vp, _vp, C.c_size_t, _i32],
        "fhesi_comm_exchange": [_vp, _vp, _vp, _vp],
        "fhesi_comm_exchange_begin": [_vp, _vp, _vp, _vp],
        "fhesi_comm_exchange_end": [_vp, _vp],
        "fhesi_comm_allreduce_rows": [_vp, _vp, _vp, _i64],
        "fhesi_ksk_download": [_vp, _vp],
        "fhesi_keyswitch_init_batch": [_vp, _vp, _i32, _vp, _i32, _i32, _vp, _i32, _vp],
        "fhesi_scrt_alloc": [_vp, _vp, _i32, _vp],
        "fhesi_scrt_from_poly": [_vp, _vp, _i32, _i64],
        "fhesi_scrt_to_poly": [_vp, _vp, _i32, _vp, _i32],
        "fhesi_scrt_op_scalar": [_vp, _vp, _i32, _i32],
        "fhesi_dcrt_assign_scrt": [_vp, _vp],
        "fhesi_scrt_assign_dcrt": [_vp, _vp, _vp, _i32],
        "fhesi_dcrt_add_primes_and_scale": [_vp, _vp, _i32, _u64, _vp],
        "fhesi_dcrt_scale_down_to_set": [_vp, _vp, _i32, _u64],
        "fhesi_ksk_upload_dev": [_vp, _vp],
        "fhesi_ksk_form": [_vp, _vp, _vp, _vp],
        "fhesi_encrypt_batch_seeded": [_vp, _vp, _vp, _i32, _u64, _u64, _u64, _vp, _i64, _vp, _i32],
        "fhesi_keyswitch_init_batch_seeded": [_vp, _vp, _i32, _vp, _i32, _i32, _u64, _u64, _u64],
        "fhesi_dcrt_sample": [_vp, _i32, _i64, _u64, _u64],
        "fhesi_ct_add_const_dev": [_vp, _i32, _u64, _vp, _i32, _i32, _i64, _vp, _i32],
        "fhesi_ct_mul_poly_dev": [_vp, _i32, _vp, _i32, _i32, _i64, _vp, _i32],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    for name in ("fhesi_ctx_m", "fhesi_ctx_phim"):
        getattr(lib, name).argtypes = [_vp]
        getattr(lib, name).restype = _i64
    lib.fhesi_ctx_nprimes.argtypes = [_vp]
    lib.fhesi_ctx_nprimes.restype = _i32
    for name in ("fhesi_ctx_stream", "fhesi_dcrt_device_ptr", "fhesi_ksk_device_ptr"):
        getattr(lib, name).argtypes = [_vp]
        getattr(lib, name).restype = _vp
    for name in ("fhesi_comm_rank", "fhesi_comm_size"):
        getattr(lib, name).argtypes = [_vp]
        getattr(lib, name).restype = _i32
    lib.fhesi_ksk_bytes.argtypes = [_vp]
    lib.fhesi_ksk_bytes.restype = C.c_size_t
    _lib = lib
    return lib


def _ck(rc: int):
    if rc != 0:
        raise FhesiError(_load().fhesi_last_error().decode())


def _p(a: np.ndarray):
    return a.ctypes.data_as(_vp)


class Backend:
    @staticmethod
    def lib():
        return _load()

    @staticmethod
    def device_count() -> int:
        n = _i32(0)
        _ck(_load().fhesi_device_count(C.byref(n)))
        return n.value


class DevBuf:
    """Plain HBM buffer owned through the C ABI."""

    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx, self.nbytes = ctx, nbytes
        self.ptr = _vp()
        _ck(_load().fhesi_dev_alloc(ctx.h, nbytes, C.byref(self.ptr)))

    def upload(self, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        _ck(_load().fhesi_dev_upload(self.ctx.h, self.ptr, _p(arr), arr.nbytes))
        return self

    def download(self, shape, dtype=np.uint64) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        _ck(_load().fhesi_dev_download(self.ctx.h, _p(out), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            _load().fhesi_dev_free(self.ctx.h, self.ptr)
            self.ptr = _vp()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """FHEcontext + Cmodulus chain on one GPU (fhesi_ctx_create)."""

    def __init__(self, m: int, primes, roots, device: int = 0):
        q = np.array([int(x) for x in primes], dtype=np.uint64)
        r = np.array([int(x) for x in roots], dtype=np.uint64)
        self.h = _vp()
        _ck(_load().fhesi_ctx_create(C.byref(self.h), m, len(q), _p(q), _p(r), device))
        self.m, self.primes, self.roots = m, [int(x) for x in q], [int(x) for x in r]
        self.L = len(self.primes)
        self.phim = _load().fhesi_ctx_phim(self.h)

    def close(self):
        if getattr(self, "h", None):
            _load().fhesi_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _ck(_load().fhesi_ctx_sync(self.h))

    def zms_idx(self) -> np.ndarray:
        out = np.zeros(self.m, dtype=np.int32)
        _ck(_load().fhesi_ctx_zms_idx(self.h, _p(out)))
        return out

    def phi_m(self) -> np.ndarray:
        out = np.zeros(self.phim + 1, dtype=np.int64)
        _ck(_load().fhesi_ctx_phi_m(self.h, _p(out)))
        return out

    def timer_start(self):
        _ck(_load().fhesi_timer_start(self.h))

    def timer_stop(self) -> float:
        ms = C.c_float(0)
        _ck(_load().fhesi_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def prof_enable(self, on: bool = True):
        _ck(_load().fhesi_prof_enable(self.h, int(on)))

    def prof_read(self, cls: str):
        """-> (launches, units, total_ms) for one kernel class of PROF_CLASSES."""
        n, u, ms = _i64(0), C.c_double(0), C.c_double(0)
        _ck(_load().fhesi_prof_read(self.h, PROF_CLASSES[cls], C.byref(n), C.byref(u), C.byref(ms)))
        return n.value, u.value, ms.value

    def prof_kernel_name(self, cls: str) -> str:
        """Demangled name of the kernel the last profiled launch of that class ran ('' if none)."""
        buf = C.create_string_buffer(512)
        _ck(_load().fhesi_prof_kernel_name(self.h, PROF_CLASSES[cls], buf, 512))
        return buf.value.decode()

    def set_option(self, name: str, value: int):
        _ck(_load().fhesi_ctx_set_option(self.h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = _i64(0)
        _ck(_load().fhesi_ctx_get_option(self.h, name.encode(), C.byref(v)))
        return v.value

    def selftest_aux32(self):
        """Diagnostic of the 32-bit auxiliary transforms of the key switch (n = 2^14): raises FhesiError on failure."""
        _ck(_load().fhesi_selftest_aux32(self.h))

    def dev_copy(self, dst_ptr: int, src_ptr: int, nbytes: int):
        _ck(_load().fhesi_dev_copy(self.h, _vp(dst_ptr), _vp(src_ptr), nbytes))

    # Cmodulus::FFT / iFFT
    def cmod_fft(self, prime: int, limbs: np.ndarray) -> np.ndarray:
        limbs = np.ascontiguousarray(limbs, dtype=np.uint64)
        y = np.zeros(self.phim, dtype=np.uint64)
        _ck(_load().fhesi_cmod_fft(self.h, prime, _p(limbs), limbs.shape[1], limbs.shape[0], _p(y)))
        return y

    def cmod_ifft(self, prime: int, y: np.ndarray) -> np.ndarray:
        y = np.ascontiguousarray(y, dtype=np.uint64)
        x = np.zeros(self.phim, dtype=np.uint64)
        _ck(_load().fhesi_cmod_ifft(self.h, prime, _p(y), _p(x)))
        return x

    def alloc(self, nbytes: int) -> DevBuf:
        return DevBuf(self, nbytes)

    def upload(self, arr: np.ndarray) -> DevBuf:
        arr = np.ascontiguousarray(arr)
        return DevBuf(self, max(arr.nbytes, 8)).upload(arr)

    # batched device-resident rows [count][L][phim]
    def rows_ntt_fwd(self, buf: DevBuf, count: int):
        _ck(_load().fhesi_rows_ntt_fwd_dev(self.h, buf.ptr, count))

    def rows_ntt_inv(self, buf: DevBuf, count: int):
        _ck(_load().fhesi_rows_ntt_inv_dev(self.h, buf.ptr, count))

    def rows_op(self, dst: DevBuf, src: DevBuf, count: int, op: int):
        _ck(_load().fhesi_rows_op_dev(self.h, dst.ptr, src.ptr, count, op))

    def ct_mul_dev(self, p: int, a: DevBuf, b: DevBuf, nlimbs: int, count: int, tprod: DevBuf):
        _ck(_load().fhesi_ct_mul_dev(self.h, p, a.ptr, b.ptr, nlimbs, count, tprod.ptr))

    def apply_key_switch_dev(self, ksk: "KeySwitchMatrix", logQ: int, tprod: DevBuf, count: int, out: DevBuf, nlimbs: int, decomp_bytes: int = 3):
        _ck(_load().fhesi_apply_key_switch_dev(self.h, ksk.h, logQ, decomp_bytes, tprod.ptr, count, out.ptr, nlimbs))

    # ---- Encrypt / Decrypt batches (FHE-SI.cpp:10-36, 93-119); randomness supplied by the caller
    def encrypt_batch(self, pk0: "DoubleCRT", pk1: "DoubleCRT", logQ: int, p: int, rand: np.ndarray, msg: np.ndarray, out: DevBuf, nlimbs: int):
        """rand: [count][3][phim] int64 = (r, e0, e1); msg: [count][phim] int64; out: device [count][2][phim][nlimbs]."""
        rand = np.ascontiguousarray(rand, dtype=np.int64)
        msg = np.ascontiguousarray(msg, dtype=np.int64)
        assert rand.shape[0] == msg.shape[0] and rand.shape[1] == 3
        _ck(_load().fhesi_encrypt_batch(self.h, pk0.h, pk1.h, logQ, p, _p(rand), _p(msg), msg.shape[0], out.ptr, nlimbs))

    def encrypt_batch_seeded(self, pk0: "DoubleCRT", pk1: "DoubleCRT", logQ: int, p: int, seed: int, first_index: int, msg: np.ndarray, out: DevBuf, nlimbs: int):
        """FHESIPubKey::Encrypt with r, e0, e1 drawn on the device from (seed, first_index + i) -- philox.h."""
        msg = np.ascontiguousarray(msg, dtype=np.int64)
        _ck(_load().fhesi_encrypt_batch_seeded(self.h, pk0.h, pk1.h, logQ, p, seed, first_index, _p(msg), msg.shape[0], out.ptr, nlimbs))

    def decrypt_batch(self, sk1: "DoubleCRT", logQ: int, p: int, ct: DevBuf, nlimbs: int, count: int) -> np.ndarray:
        msg = np.zeros((count, self.phim), dtype=np.int64)
        _ck(_load().fhesi_decrypt_batch(self.h, sk1.h, logQ, p, ct.ptr, nlimbs, count, _p(msg)))
        return msg

    # ---- ciphertext algebra between multiplications (Matrix<Ciphertext> / Regression), batches resident in HBM
    def ct_add_dev(self, logQ: int, dst: DevBuf, src: DevBuf, nparts: int, nlimbs: int, count: int):
        _ck(_load().fhesi_ct_add_dev(self.h, logQ, dst.ptr, src.ptr, nparts, nlimbs, count))

    def ct_mul_long_dev(self, logQ: int, ct: DevBuf, l: int, nparts: int, nlimbs: int, count: int):
        _ck(_load().fhesi_ct_mul_long_dev(self.h, logQ, ct.ptr, l, nparts, nlimbs, count))

    def ct_add_const_dev(self, logQ: int, p: int, ct: DevBuf, nparts: int, nlimbs: int, count: int, poly: np.ndarray):
        """Ciphertext::operator+=(const ZZX&) on unscaled ciphertexts (Ciphertext.cpp:147-156); poly [npoly][phim] int64, npoly 1 or count."""
        poly = np.ascontiguousarray(poly, dtype=np.int64).reshape(-1, self.phim)
        _ck(_load().fhesi_ct_add_const_dev(self.h, logQ, p, ct.ptr, nparts, nlimbs, count, _p(poly), poly.shape[0]))

    def ct_mul_poly_dev(self, logQ: int, ct: DevBuf, nparts: int, nlimbs: int, count: int, poly: np.ndarray):
        """Ciphertext::operator*=(const ZZX&) on unscaled ciphertexts (Ciphertext.cpp:245-249, :29-36)."""
        poly = np.ascontiguousarray(poly, dtype=np.int64).reshape(-1, self.phim)
        _ck(_load().fhesi_ct_mul_poly_dev(self.h, logQ, ct.ptr, nparts, nlimbs, count, _p(poly), poly.shape[0]))

    def rows_mul_long_dev(self, rows: DevBuf, l: int, count: int):
        _ck(_load().fhesi_rows_mul_long_dev(self.h, rows.ptr, l, count))

    def ct_automorph_dev(self, k: int, src: DevBuf, nparts: int, nlimbs_in: int, count: int, out: DevBuf, nlimbs_out: int):
        _ck(_load().fhesi_ct_automorph_dev(self.h, k, src.ptr, nparts, nlimbs_in, count, out.ptr, nlimbs_out))

    def ct_automorph_key_switch_dev(self, ksk: "KeySwitchMatrix", logQ: int, k: int, src: DevBuf, nlimbs_in: int, count: int, out: DevBuf,
                                    nlimbs: int, decomp_bytes: int = 3):
        _ck(_load().fhesi_ct_automorph_key_switch_dev(self.h, ksk.h, logQ, decomp_bytes, k, src.ptr, nlimbs_in, count, out.ptr, nlimbs))

    def ct_gather_dev(self, pool: DevBuf, idx, words: int, out: DevBuf):
        ia = np.ascontiguousarray(idx, dtype=np.int32)
        _ck(_load().fhesi_ct_gather_dev(self.h, pool.ptr, _p(ia), len(ia), words, out.ptr))

    def ct_mul_sum_relin_dev(self, ksk: "KeySwitchMatrix", logQ: int, p: int, pool: DevBuf, nlimbs: int, a_idx, b_idx, seg, out: DevBuf,
                             decomp_bytes: int = 3):
        """out[g] = KeySwitch(sum_{t in [seg[g], seg[g+1])} pool[a_idx[t]] * pool[b_idx[t]]): one wave of Matrix<Ciphertext> products."""
        ia, ib = np.ascontiguousarray(a_idx, dtype=np.int32), np.ascontiguousarray(b_idx, dtype=np.int32)
        sg = np.ascontiguousarray(seg, dtype=np.int32)
        assert len(ia) == len(ib) == int(sg[-1]) and sg[0] == 0
        _ck(_load().fhesi_ct_mul_sum_relin_dev(self.h, ksk.h, logQ, p, decomp_bytes, pool.ptr, nlimbs, _p(ia), _p(ib), _p(sg), len(sg) - 1, out.ptr))

    def ct_mul_relin_dev(self, ksk: "KeySwitchMatrix", logQ: int, p: int, a: DevBuf, b: DevBuf, out: DevBuf, nlimbs: int, count: int, decomp_bytes: int = 3):
        _ck(_load().fhesi_ct_mul_relin_batch_dev(self.h, ksk.h, logQ, p, decomp_bytes, a.ptr, b.ptr, out.ptr, nlimbs, count))

    def ct_mul_relin(self, ksk: "KeySwitchMatrix", logQ: int, p: int, a: np.ndarray, b: np.ndarray, decomp_bytes: int = 3, out: np.ndarray = None) -> np.ndarray:
        """a, b: [count][2][phim][nlimbs] uint64 two's complement -> same shape (host buffers: fhesi_ct_mul_relin_batch).  `out`: a result
        array to reuse (a fresh numpy array costs a page fault per 4 KiB on first touch); arrays from host_array() are pinned and take the
        DMA path without the staging copy."""
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        if out is None:
            out = np.empty_like(a)
        elif out.shape != a.shape or out.dtype != np.uint64 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous uint64 array of the operands' shape")
        _ck(_load().fhesi_ct_mul_relin_batch(self.h, ksk.h, logQ, p, decomp_bytes, _p(a), _p(b), _p(out), a.shape[-1], a.shape[0]))
        return out

    def host_array(self, shape, dtype=np.uint64) -> np.ndarray:
        """numpy array over PINNED host memory (fhesi_host_alloc); freed when the array (and its views) are collected."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        ptr = _vp()
        _ck(_load().fhesi_host_alloc(self.h, n, C.byref(ptr)))
        buf = (C.c_char * max(n, 1)).from_address(ptr.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        import weakref
        # (the finalizer holds no context handle: fhesi_host_free does not dereference it, so the array may outlive the Context -- at
        # interpreter shutdown the order of collection is arbitrary)
        lib, addr = _load(), ptr.value
        weakref.finalize(buf, lambda: lib.fhesi_host_free(None, C.c_void_p(addr)))
        return arr

    def release_host_staging(self):
        """hand back the pinned + device staging ring the host-buffer calls keep between uses"""
        _ck(_load().fhesi_host_stage_release(self.h))


class DoubleCRT:
    """One DoubleCRT object resident in HBM (DoubleCRT.h:83-365 through the C ABI)."""

    def __init__(self, ctx: Context, index_set=None):
        self.ctx = ctx
        self.h = _vp()
        if index_set is None:
            _ck(_load().fhesi_dcrt_alloc(ctx.h, None, 0, C.byref(self.h)))
        else:
            ia = np.array(list(index_set), dtype=np.int32)
            if len(ia) == 0:
                raise FhesiError("DoubleCRT: empty index set")
            _ck(_load().fhesi_dcrt_alloc(ctx.h, _p(ia), len(ia), C.byref(self.h)))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                _load().fhesi_dcrt_free(self.h)
                self.h = None
        except Exception:
            pass

    @classmethod
    def from_poly(cls, ctx: Context, limbs: np.ndarray, index_set=None) -> "DoubleCRT":
        d = cls(ctx, index_set)
        d.assign_poly(limbs)
        return d

    def index_set(self):
        n = _i32(0)
        buf = np.zeros(self.ctx.L, dtype=np.int32)
        _ck(_load().fhesi_dcrt_index_set(self.h, _p(buf), C.byref(n)))
        return [int(x) for x in buf[:n.value]]

    def assign_poly(self, limbs: np.ndarray):
        limbs = np.ascontiguousarray(limbs, dtype=np.uint64)
        _ck(_load().fhesi_dcrt_from_poly(self.h, _p(limbs), limbs.shape[1], limbs.shape[0]))

    def assign(self, other: "DoubleCRT"):
        _ck(_load().fhesi_dcrt_copy(self.h, other.h))

    def copy(self) -> "DoubleCRT":
        d = DoubleCRT(self.ctx, self.index_set())
        d.assign(self)
        return d

    def to_poly(self, nlimbs: int, index_set=None, positive: bool = False) -> np.ndarray:
        out = np.zeros((self.ctx.phim, nlimbs), dtype=np.uint64)
        if index_set is None:
            _ck(_load().fhesi_dcrt_to_poly(self.h, None, 0, int(positive), _p(out), nlimbs))
        else:
            ia = np.array(list(index_set), dtype=np.int32)
            if len(ia) == 0:      # empty intersection -> zero polynomial (DoubleCRT.cpp:354-357)
                return out
            _ck(_load().fhesi_dcrt_to_poly(self.h, _p(ia), len(ia), int(positive), _p(out), nlimbs))
        return out

    def row(self, prime: int) -> np.ndarray:
        out = np.zeros(self.ctx.phim, dtype=np.uint64)
        _ck(_load().fhesi_dcrt_download_row(self.h, prime, _p(out)))
        return out

    def sample(self, kind: int, param: int, seed: int, index: int):
        """DoubleCRT::sampleHWt(param) (kind 0) / sampleGaussian() (kind 1) drawn on the device from (seed, index) -- philox.h."""
        _ck(_load().fhesi_dcrt_sample(self.h, kind, param, seed, index))
        return self

    def set_row(self, prime: int, row: np.ndarray):
        row = np.ascontiguousarray(row, dtype=np.uint64)
        _ck(_load().fhesi_dcrt_upload_row(self.h, prime, _p(row)))

    def rows(self) -> np.ndarray:
        return np.stack([self.row(i) for i in self.index_set()])

    def op(self, other: "DoubleCRT", op: int):
        _ck(_load().fhesi_dcrt_op(self.h, other.h, op))
        return self

    def op_scalar(self, num: int, op: int, nlimbs: int = 4):
        mod = 1 << (64 * nlimbs)
        v = num % mod
        s = np.array([(v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(nlimbs)], dtype=np.uint64)
        _ck(_load().fhesi_dcrt_op_scalar(self.h, _p(s), nlimbs, op))
        return self

    def automorph(self, k: int):
        _ck(_load().fhesi_dcrt_automorph(self.h, k))
        return self

    def exp(self, e: int):
        """DoubleCRT::Exp (DoubleCRT.cpp:423-434): element-wise PowerMod."""
        _ck(_load().fhesi_dcrt_exp(self.h, e))
        return self

    def add_primes(self, idx):
        ia = np.array(list(idx), dtype=np.int32)
        _ck(_load().fhesi_dcrt_add_primes(self.h, _p(ia), len(ia)))

    def remove_primes(self, idx):
        ia = np.array(list(idx), dtype=np.int32)
        _ck(_load().fhesi_dcrt_remove_primes(self.h, _p(ia), len(ia)))

    def add_primes_and_scale(self, idx, p: int) -> float:
        """DoubleCRT::addPrimesAndScale (DoubleCRT.cpp:162-208); returns the logarithm of the scaling factor."""
        ia = np.array(list(idx), dtype=np.int32)
        lf = C.c_double(0.0)
        _ck(_load().fhesi_dcrt_add_primes_and_scale(self.h, _p(ia), len(ia), p, C.byref(lf)))
        return lf.value

    def scale_down_to_set(self, idx, p: int):
        """DoubleCRT::scaleDownToSet (DoubleCRT.cpp:518-558)."""
        ia = np.array(list(idx), dtype=np.int32)
        _ck(_load().fhesi_dcrt_scale_down_to_set(self.h, _p(ia), len(ia), p))

    def equals(self, other: "DoubleCRT") -> bool:
        eq = _i32(0)
        _ck(_load().fhesi_dcrt_equal(self.h, other.h, C.byref(eq)))
        return bool(eq.value)

    def from_scrt(self, coeff_rows: np.ndarray):
        coeff_rows = np.ascontiguousarray(coeff_rows, dtype=np.uint64)
        _ck(_load().fhesi_dcrt_from_scrt(self.h, _p(coeff_rows)))

    def to_scrt(self) -> np.ndarray:
        out = np.zeros((len(self.index_set()), self.ctx.phim), dtype=np.uint64)
        _ck(_load().fhesi_dcrt_to_scrt(self.h, _p(out)))
        return out


class SingleCRT:
    """One SingleCRT object resident in HBM (SingleCRT.h:41-175 through the C ABI): coefficient residues per prime."""

    def __init__(self, ctx: Context, index_set=None):
        self.ctx = ctx
        self.h = _vp()
        if index_set is None:
            _ck(_load().fhesi_scrt_alloc(ctx.h, None, 0, C.byref(self.h)))
        else:
            ia = np.array(list(index_set), dtype=np.int32)
            if len(ia) == 0:
                raise FhesiError("SingleCRT: empty index set")
            _ck(_load().fhesi_scrt_alloc(ctx.h, _p(ia), len(ia), C.byref(self.h)))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                _load().fhesi_dcrt_free(self.h)
                self.h = None
        except Exception:
            pass

    index_set = DoubleCRT.index_set
    row = DoubleCRT.row
    set_row = DoubleCRT.set_row
    rows = DoubleCRT.rows
    remove_primes = DoubleCRT.remove_primes

    def assign_poly(self, limbs: np.ndarray):
        limbs = np.ascontiguousarray(limbs, dtype=np.uint64)
        _ck(_load().fhesi_scrt_from_poly(self.h, _p(limbs), limbs.shape[1], limbs.shape[0]))
        return self

    def to_poly(self, nlimbs: int, index_set=None) -> np.ndarray:
        out = np.zeros((self.ctx.phim, nlimbs), dtype=np.uint64)
        if index_set is None:
            _ck(_load().fhesi_scrt_to_poly(self.h, None, 0, _p(out), nlimbs))
        else:
            ia = np.array(list(index_set), dtype=np.int32)
            if len(ia) == 0:
                return out
            _ck(_load().fhesi_scrt_to_poly(self.h, _p(ia), len(ia), _p(out), nlimbs))
        return out

    def assign(self, other: "SingleCRT"):
        _ck(_load().fhesi_dcrt_copy(self.h, other.h))

    def op(self, other: "SingleCRT", op: int):
        _ck(_load().fhesi_dcrt_op(self.h, other.h, op))
        return self

    def op_scalar(self, num: int, op: int, nlimbs: int = 4):
        mod = 1 << (64 * nlimbs)
        v = num % mod
        s = np.array([(v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(nlimbs)], dtype=np.uint64)
        _ck(_load().fhesi_scrt_op_scalar(self.h, _p(s), nlimbs, op))
        return self

    def equals(self, other) -> bool:
        eq = _i32(0)
        _ck(_load().fhesi_dcrt_equal(self.h, other.h, C.byref(eq)))
        return bool(eq.value)

    def assign_dcrt(self, d: DoubleCRT, index_set=None):
        """DoubleCRT::toSingleCRT (DoubleCRT.cpp:498-515)."""
        if index_set is None:
            _ck(_load().fhesi_scrt_assign_dcrt(self.h, d.h, None, 0))
        else:
            ia = np.array(list(index_set), dtype=np.int32)
            _ck(_load().fhesi_scrt_assign_dcrt(self.h, d.h, _p(ia), len(ia)))
        return self


def dcrt_assign_scrt(d: DoubleCRT, s: SingleCRT):
    """DoubleCRT::operator=(const SingleCRT&) (DoubleCRT.cpp:484-496)."""
    _ck(_load().fhesi_dcrt_assign_scrt(d.h, s.h))
    return d


class Comm:
    """One rank of a multi-GPU group (fhesi_comm: an RCCL communicator; a loopback group when ranks share a device).  Collective
    methods must be called by every rank of the group concurrently (one host thread per rank; ctypes releases the GIL)."""

    def __init__(self, handle):
        self.h = handle

    @staticmethod
    def init_all(devices):
        devs = np.array(list(devices), dtype=np.int32)
        hs = (_vp * len(devs))()
        _ck(_load().fhesi_comm_init_all(len(devs), _p(devs), hs))
        return [Comm(_vp(h)) for h in hs]

    @property
    def rank(self) -> int:
        return _load().fhesi_comm_rank(self.h)

    @property
    def size(self) -> int:
        return _load().fhesi_comm_size(self.h)

    def ksk_broadcast(self, ksk: "KeySwitchMatrix", root: int = 0):
        _ck(_load().fhesi_ksk_broadcast(ksk.h, self.h, root))

    def exchange(self, ctx: Context, base: DevBuf, offsets_words):
        off = np.ascontiguousarray(offsets_words, dtype=np.int64)
        _ck(_load().fhesi_comm_exchange(ctx.h, self.h, base.ptr, _p(off)))

    def exchange_begin(self, ctx: Context, base: DevBuf, offsets_words):
        """the exchange enqueued on the communicator's own stream behind the context's stream; returns without waiting for the GPU"""
        off = np.ascontiguousarray(offsets_words, dtype=np.int64)
        _ck(_load().fhesi_comm_exchange_begin(ctx.h, self.h, base.ptr, _p(off)))

    def exchange_end(self, ctx: Context):
        """every exchange begun since the last end has landed"""
        _ck(_load().fhesi_comm_exchange_end(ctx.h, self.h))

    def allreduce_rows(self, ctx: Context, rows: DevBuf, count: int):
        _ck(_load().fhesi_comm_allreduce_rows(ctx.h, self.h, rows.ptr, count))

    def destroy(self):
        if self.h:
            _load().fhesi_comm_destroy(self.h)
            self.h = None


class KeySwitchMatrix:
    """KeySwitchSI::keySwitchMatrix resident in HBM: [2][ncomp*ndigits][L][phim]."""

    def __init__(self, ctx: Context, ncomp: int, ndigits: int):
        self.ctx, self.ncomp, self.ndigits = ctx, ncomp, ndigits
        self.h = _vp()
        _ck(_load().fhesi_ksk_create(ctx.h, ncomp, ndigits, C.byref(self.h)))

    def upload(self, rows: np.ndarray):
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        assert rows.nbytes == self.nbytes, (rows.nbytes, self.nbytes)
        _ck(_load().fhesi_ksk_upload(self.h, _p(rows)))
        return self

    @property
    def nbytes(self) -> int:
        return _load().fhesi_ksk_bytes(self.h)

    @property
    def device_ptr(self) -> int:
        return _load().fhesi_ksk_device_ptr(self.h)

    def download(self) -> np.ndarray:
        out = np.zeros((2, self.ncomp * self.ndigits, self.ctx.L, self.ctx.phim), dtype=np.uint64)
        _ck(_load().fhesi_ksk_download(self.h, _p(out)))
        return out

    def init_batch(self, src, dst_t: "DoubleCRT", logQ: int, a: np.ndarray, err: np.ndarray, decomp_bytes: int = 3):
        """KeySwitchSI::Init (FHE-SI.cpp:153-209) for all columns at once: src = the source key's DoubleCRT components, a = the random
        polynomials [ncol][phim][nlimbs], err = the Gaussian errors [ncol][phim], drawn by the caller in the reference's order."""
        a = np.ascontiguousarray(a, dtype=np.uint64)
        err = np.ascontiguousarray(err, dtype=np.int64)
        assert a.shape[0] == err.shape[0] == self.ncomp * self.ndigits
        hs = (_vp * len(src))(*[d.h for d in src])
        _ck(_load().fhesi_keyswitch_init_batch(self.h, hs, len(src), dst_t.h, logQ, decomp_bytes, _p(a), a.shape[-1], _p(err)))
        return self

    def init_batch_seeded(self, src, dst_t: "DoubleCRT", logQ: int, seed: int, public_seed: int, first_index: int, decomp_bytes: int = 3):
        """KeySwitchSI::Init with the column randomness drawn on the device -- philox.h: the public polynomials a from (public_seed, first_index +
        column), the secret errors from (seed, first_index + column).  No default index: an (seed, index) pair must never be used twice."""
        if public_seed == seed:
            raise ValueError("public_seed must differ from the secret seed")
        hs = (_vp * len(src))(*[d.h for d in src])
        _ck(_load().fhesi_keyswitch_init_batch_seeded(self.h, hs, len(src), dst_t.h, logQ, decomp_bytes, seed, public_seed, first_index))
        return self

    FORMS = {-1: "none yet", 0: "per chain prime", 1: "four 30-bit auxiliary primes, limbs", 2: "two largest chain primes, limbs", 3: "two largest chain primes, residues"}

    def form(self):
        """(form, rows, limb_bits) of the last key switch with this matrix (fhesi_ksk_form): which exact form of the dot product ran."""
        f, r, b = C.c_int32(), C.c_int32(), C.c_int32()
        _ck(_load().fhesi_ksk_form(self.h, C.byref(f), C.byref(r), C.byref(b)))
        return f.value, r.value, b.value

    def key_bits(self):
        """(centred, nb) of the last table built from this matrix (fhesi_ksk_key_bits): centred limbs of a generated matrix, and the measured
        size of its integer coefficients."""
        c, b = C.c_int32(), C.c_int32()
        _ck(_load().fhesi_ksk_key_bits(self.h, C.byref(c), C.byref(b)))
        return bool(c.value), b.value

    def mark_dirty(self):
        """The rows were written through device_ptr (e.g. by a collective): derived tables are rebuilt at the next key switch."""
        _ck(_load().fhesi_ksk_mark_dirty(self.h))

    def upload_dev(self, src_ptr: int):
        """Whole matrix from another HBM buffer of the same device (the staging tensor of an RCCL broadcast)."""
        _ck(_load().fhesi_ksk_upload_dev(self.h, _vp(src_ptr)))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                _load().fhesi_ksk_free(self.h)
                self.h = None
        except Exception:
            pass
