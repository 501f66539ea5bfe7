#!/usr/bin/env python3
"""Extended walk of tests/test_gpu_shape_fuzz.py: the same two test bodies on seeds beyond the committed parametrisation.
usage: python3 tools/fuzz_shapes.py FIRST LAST   (runs on a GPU box; prints one line per failure and a summary)"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import conftest  # noqa: F401  (paths)
import test_gpu_shape_fuzz as T
first, last = int(sys.argv[1]), int(sys.argv[2])
bad = 0
t0 = time.time()
for seed in range(first, last):
    for name, fn, ok in (("uniform", T.test_mul_relin_on_random_shapes, True), ("generated", T.test_mul_relin_on_random_shapes_with_generated_keys, seed % 3 == 0)):
        if not ok:
            continue
        try:
            fn(seed)
        except Exception:
            bad += 1
            print("FAIL", name, seed, T._case(seed) if name == "uniform" else "", flush=True)
            traceback.print_exc(limit=2)
print(f"seeds {first}..{last - 1}: {bad} failures, {time.time() - t0:.0f} s", flush=True)
