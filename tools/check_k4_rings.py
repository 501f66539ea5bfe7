import sys, os
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import fhe_si_amd as F, params as P, fhesi_pyref as R
for m, logQ, p in ((65266, 512, 65267), (32602, 512, 32603)):
    primes, roots = P.chain_for(m, logQ, p)
    ctx = F.Context(m, primes, roots)
    n, nd, nl = ctx.phim, R.ndigits(logQ), (logQ + 63) // 64
    one = np.zeros((n, 1), dtype=np.uint64); one[0, 0] = 1
    t = F.DoubleCRT(ctx).sample(0, 64, 5, 1); t2 = t.copy(); t2.op(t, 2)
    ksk = F.KeySwitchMatrix(ctx, 3, nd).init_batch_seeded([F.DoubleCRT.from_poly(ctx, one), t, t2], t, logQ, 5, 6, 100, 3)
    rng = np.random.default_rng(3)
    count = 29
    a = P.rand_limbs(rng, (count, 2, n), nl, logQ); b = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    da, db, dout = ctx.upload(a), ctx.upload(b), ctx.alloc(a.nbytes)
    ctx.prof_enable(True)
    ctx.ct_mul_relin_dev(ksk, logQ, p, da, db, dout, nl, count); ctx.sync()
    name = ctx.prof_kernel_name("dot"); ctx.prof_enable(False)
    got = dout.download((count, 2, n, nl))
    ctx.set_option("dot32_k4", 0)
    ctx.ct_mul_relin_dev(ksk, logQ, p, da, db, dout, nl, count)
    ref = dout.download((count, 2, n, nl))
    print(m, name, ksk.form(), "equal:", bool(np.array_equal(got, ref)))
