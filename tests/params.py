"""Shared parameter builders for the tests (chains follow FHEContext.cpp:83-115 through the Python restatement)."""
from __future__ import annotations

import functools

import numpy as np

import fhesi_pyref as R


@functools.lru_cache(maxsize=None)
def chain_for(m: int, logQ: int, p: int, xi: int = 1, sp_nbits: int = 60):
    _, phim = R.zms_idx(m)
    primes = R.add_primes_by_size(m, R.si_context_size(logQ, p, phim, xi), sp_nbits)
    roots = [R.find_root_2m(q, m) for q in primes]
    return tuple(primes), tuple(roots)


@functools.lru_cache(maxsize=None)
def first_primes(m: int, count: int, sp_nbits: int = 60):
    """first `count` primes = 1 mod 2m descending from 2^sp_nbits (rule of FHEContext.cpp:92-108), deterministic roots."""
    primes = []
    p = (1 << sp_nbits) - 1
    p -= p % (2 * m)
    p += 2 * m + 1
    while len(primes) < count:
        p -= 2 * m
        if R.is_prime(p):
            primes.append(p)
    roots = [R.find_root_2m(q, m) for q in primes]
    return tuple(primes), tuple(roots)


def rand_rows(rng: np.random.Generator, primes, n: int, count: int = 1) -> np.ndarray:
    """[count][L][n] uniform residues."""
    out = np.empty((count, len(primes), n), dtype=np.uint64)
    for i, q in enumerate(primes):
        out[:, i, :] = rng.integers(0, q, size=(count, n), dtype=np.uint64)
    return out


def rand_limbs(rng: np.random.Generator, shape, nlimbs: int, bits: int) -> np.ndarray:
    """signed values uniform in [-2^(bits-1), 2^(bits-1)) as two's complement limbs, shape + [nlimbs]."""
    total = int(np.prod(shape))
    raw = rng.integers(0, 1 << 63, size=(total, nlimbs), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(total, nlimbs), dtype=np.uint64)
    out = np.zeros((total, nlimbs), dtype=np.uint64)
    full, rem = divmod(bits, 64)
    out[:, :full] = raw[:, :full]
    if rem:
        v = raw[:, full] & np.uint64((1 << rem) - 1)
        sign = (v >> np.uint64(rem - 1)) & np.uint64(1)
        ext = np.where(sign == 1, np.uint64(((1 << 64) - 1) ^ ((1 << rem) - 1)), np.uint64(0))
        out[:, full] = v | ext
        for k in range(full + 1, nlimbs):
            out[:, k] = np.where(sign == 1, np.uint64((1 << 64) - 1), np.uint64(0))
    else:
        sign = out[:, full - 1] >> np.uint64(63)
        for k in range(full, nlimbs):
            out[:, k] = np.where(sign == 1, np.uint64((1 << 64) - 1), np.uint64(0))
    return out.reshape(tuple(shape) + (nlimbs,))
