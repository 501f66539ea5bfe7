#!/usr/bin/env python3
"""Table A of DESIGN.md section 4.2 from the committed profile files of one round:
   tools/design_tables.py r06_f      (reads profiles/<tag>_metric_kernel_times.csv, profiles/pmc_*.json, profiles/sq_main_kernels.json)
Per kernel of the metric step (1024 multiplications per launch): rocprof min / median / mean ms, SURVEY 8(d)'s algorithmic GB, the PMC bytes,
the fraction of 8 TB/s on the algorithmic bytes at the MEDIAN launch, VALU busy, VALU wave-instructions per launch and lane-instructions per
multiplication with their share of the step."""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
# algorithmic bytes per launch of 1024 multiplications (n = 2^14, nl = 8, nd = 22, NP = 35 tensor primes, 7 centred limbs), SURVEY 8(d):
#   digit rows: 1024*66*4 rows written (2^14*4 B) + the 3 scaled-down polynomials of 8 limbs read once (K2')
#   plain rows: read + written; tensor inverse: 4 source rows read, 3 written per (ciphertext, prime); dot: digit rows + outputs + keys once
N, B = 1 << 14, 1024
ALG = {"ntt32_fwd_kernel3<true, 0, false, Aux32Primes": (B * 66 * 4 * N * 4 + B * 3 * 8 * 8 * N, "digit rows (ByteDecomp + forward): rows written + source once"),
       "ntt32_fwd_kernel3<false, 0, false, T32Primes": (B * 4 * 35 * 2 * N * 4, "tensor forward rows"),
       "dot32_kernel4<7": ((B * 66 * 4 + B * 2 * 7 * 4 + 66 * 2 * 7 * 4) * N * 4, "key-switch dot product"),
       "ntt32_inv_kernel3<false, true, T32Primes": (B * 35 * (4 + 3) * N * 4, "tensor product + inverse rows"),
       "rns32_reduce_kernel<8": (B * 4 * (8 * 8 * N + 35 * N * 4), "operands -> residues"),
       "crt32_scale_kernel<512, false": (B * 3 * (35 * N * 4 + 8 * 8 * N), "residues -> round(x / 2^logQ)"),
       "ntt32_inv_kernel3<true, false, Aux32Primes": (B * 2 * 7 * 4 * 2 * N * 4, "key-switch inverse rows"),
       "ks_recombine_centred_kernel<8, 7": (B * 2 * (7 * 4 * N * 4 + 8 * 8 * N), "recombination")}
PMC = {"ntt32_fwd_kernel3<true": "pmc_ntt_fwd.json", "ntt32_fwd_kernel3<false": "pmc_t32_fwd.json", "dot32_kernel4": "pmc_dot_aux.json", "ntt32_inv_kernel3<false": "pmc_t32_inv.json",
       "rns32_reduce": "pmc_t32_rns.json", "crt32_scale": "pmc_t32_crt.json", "ntt32_inv_kernel3<true": "pmc_ntt_inv.json", "ks_recombine": "pmc_recombine.json"}
times = {}
for r in csv.DictReader(open(os.path.join(ROOT, "profiles", f"{tag}_metric_kernel_times.csv"))):
    times[r["kernel"]] = r
sq = json.load(open(os.path.join(ROOT, "profiles", "sq_main_kernels.json")))["kernels"]
rows, tot_i, tot_ms = [], 0.0, 0.0
for key, (alg, what) in ALG.items():
    t = next((v for k, v in times.items() if key in k), None)
    q = next((v for k, v in sq.items() if key in k), None)
    pm = None
    for pk, pf in PMC.items():
        if key.startswith(pk):
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", pf)))["hbm_bytes_per_launch"]
            except Exception:
                pm = None
    if not t:
        continue
    med = float(t["median_ms"])
    wi = q["valu_wave_instr_per_launch"] if q else None
    rows.append((key, what, float(t["min_ms"]), med, float(t["mean_ms"]), alg / 1e9, pm / 1e9 if pm else None, alg / (med * 1e-3) / 8e12, q["valu_busy"] if q else None, wi))
    tot_i += wi or 0
    tot_ms += med
print("| kernel | ms per 1024: min / median / mean | alg. GB | moved GB (PMC) | frac of 8 TB/s (alg., median) | VALU busy | VALU wave-instr (G) | lane-instr per mult (M) | share |")
print("|---|---|---|---|---|---|---|---|---|")
for key, what, mn, med, mean, alg, pm, frac, busy, wi in rows:
    print(f"| `{key}…>` {what} | {mn:.2f} / {med:.2f} / {mean:.2f} | {alg:.2f} | {pm:.2f} | {frac:.2f} | {busy:.2f} | {wi / 1e9:.2f} | {wi * 64 / 1024 / 1e6:.0f} | {wi / tot_i * 100:.0f} % |" if wi and pm else
          f"| `{key}…>` {what} | {mn:.2f} / {med:.2f} / {mean:.2f} | {alg:.2f} | {'—' if not pm else '%.2f' % pm} | {frac:.2f} | {'—' if busy is None else busy} | — | — | — |")
alg_tot = sum(r[5] for r in rows)
print(f"| **step** (Σ medians) | {tot_ms:.2f} | {alg_tot:.1f} | | {alg_tot * 1e9 / (tot_ms * 1e-3) / 8e12:.2f} | | {tot_i / 1e9:.2f} | {tot_i * 64 / 1024 / 1e6:.0f} | |")
