"""Checker-backed backend for fhe_si_amd.regression.regress_waves: every ciphertext operation is evaluated by the Python
big-int model (oracle/fhesi_pyref.py) on a host-memory pool.  Test infrastructure only -- it lets the CPU suite exercise the
wave schedule and its N>1 sharding (gloo) without a GPU."""
import numpy as np

import fhesi_pyref as R
import oracle_lib as O


class PyrefBackend:
    def __init__(self, rctx, ksm, auto_ksms, auto_ks, pool, overlap=1):
        self.c, self.ksm, self.auto_ksms, self.auto_ks, self.pool, self.overlap = rctx, ksm, list(auto_ksms), list(auto_ks), pool, overlap
        self.nl = (rctx.logQ + 63) // 64
        assert pool.words == 2 * rctx.phim * self.nl

    def _get(self, idx):
        a = self.pool.t[idx].numpy().view(np.uint64).reshape(2, self.c.phim, self.nl)
        return [O.limbs_to_ints(a[0]), O.limbs_to_ints(a[1])]

    def _put(self, idx, parts):
        a = np.stack([O.ints_to_limbs(parts[0], self.nl), O.ints_to_limbs(parts[1], self.nl)])
        self.pool.t[idx].copy_(self.pool.torch.from_numpy(a.view(np.int64).reshape(-1)))

    def upload(self, cts):
        first = self.pool.alloc(len(cts))
        for i, parts in enumerate(cts):
            self._put(first + i, parts)
        return first

    def download(self, idx):
        return self._get(idx)

    def run_wave(self, w):
        first = self.pool.alloc(w.groups)

        def compute(lo, hi):
            for g in range(lo, hi):
                tp = None
                for t in range(w.seg[g], w.seg[g + 1]):
                    prod = R.ct_mul(self.c, self._get(w.a[t]), self._get(w.b[t]))
                    tp = prod if tp is None else R.tprod_add(self.c, tp, prod)
                self._put(first + g, R.apply_key_switch(self.c, self.ksm, tp))
        self.pool.run_sharded(first, w.groups, compute, self.overlap)
        return first

    def sum_batched(self, first, count):
        def compute(lo, hi):
            for i in range(lo, hi):
                self._put(first + i, R.sum_batched_data(self.c, self.auto_ksms, self.auto_ks, self._get(first + i)))
        self.pool.run_sharded(first, count, compute, self.overlap)

    def negated(self, idx):
        first = self.pool.alloc(len(idx))
        for i, j in enumerate(idx):
            self._put(first + i, R.ct_mul_long(self.c, self._get(j), -1))
        return first


def regression_case(m=22, p=23, g=7, d=3, N=2, seed=77):
    """Valid keys, encrypted random data, and the literal evaluation of Regression::Regress by the Python model."""
    import math
    n0 = (p - 1) // 2 - 1
    xi = max(N, d)
    lgQ = 4.5 * math.log(n0) + max(1, d - 1) * (math.log(1280) + 2 * math.log(n0) + math.log(xi))     # Test_Regression.cpp:107-108
    logQ = int(math.ceil(lgQ / math.log(2) + 24.7))
    _, phim = R.zms_idx(m)
    primes = R.add_primes_by_size(m, R.si_context_size(logQ, p, phim, xi))
    roots = [R.find_root_2m(q, m) for q in primes]
    ctx = R.Ctx(m, logQ, p, primes, roots)
    rng = R.SplitMix64(seed)
    t, pk = R.keygen(ctx, rng)
    ksm = R.key_switch_init_s2(ctx, t, rng)
    ks = R.automorph_generators(m, g, R.usable_slots(m, p))
    auto = [R.key_switch_init_automorph(ctx, t, k, rng) for k in ks]
    msgX = [[[rng.bnd(p) for _ in range(phim)] for _ in range(d)] for _ in range(N)]
    msgY = [[rng.bnd(p) for _ in range(phim)] for _ in range(N)]
    X = [[R.encrypt(ctx, pk, msgX[i][j], rng) for j in range(d)] for i in range(N)]
    y = [R.encrypt(ctx, pk, msgY[i], rng) for i in range(N)]
    return dict(ctx=ctx, t=t, ksm=ksm, ks=ks, auto=auto, X=X, y=y, msgX=msgX, msgY=msgY, primes=primes, roots=roots)
