#!/usr/bin/env python3
"""The cumulative table of |round(N(0, 3.2^2))| used by the counter-based samplers (fhe-si_amd/csrc/philox.h, oracle/fhesi_oracle.c,
oracle/fhesi_pyref.py): entry k = 2^64 - floor(2^64 erfc((k + 1/2) / (sigma sqrt 2))), the last one saturated.  The committed constants
ARE the definition (the low bits carry the rounding of a double-precision erfc); this script documents where they came from."""
import math
s, T, k = 3.2, [], 0
while True:
    e = math.erfc((k + 0.5) / (s * math.sqrt(2)))
    v = (1 << 64) - int(e * (1 << 64))
    if v >= (1 << 64) - 1 or e * (1 << 64) < 1:
        T.append((1 << 64) - 1)
        break
    T.append(v)
    k += 1
print(len(T))
print(", ".join(hex(x) for x in T))
