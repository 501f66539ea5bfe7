import sys, time, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle')
import numpy as np
import fhe_si_amd as F
import params as P
t0=time.time()
m, logQ, p = 4096, 128, 23
primes, roots = P.chain_for(m, logQ, p)
ctx = F.Context(m, primes, roots, device=0)
print("ctx", time.time()-t0); t0=time.time()
comms = F.Comm.init_all([0])
print("comm init_all([0])", time.time()-t0); t0=time.time()
nd=6
ksk = F.KeySwitchMatrix(ctx, 3, nd)
comms[0].ksk_broadcast(ksk, 0)
print("broadcast", time.time()-t0); t0=time.time()
comms[0].destroy()
print("destroy", time.time()-t0)
