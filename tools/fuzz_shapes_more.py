"""More seeds of tests/test_gpu_shape_fuzz.py than the suite runs (GPU box, from the repo root): python tools/fuzz_shapes_more.py  -- seeds 24..423;
prints every failing shape.  Round 3: 400 shapes, 0 failures."""
import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.path.join(os.getcwd(), "oracle")); sys.path.insert(0, os.getcwd())
import test_gpu_shape_fuzz as T
bad = 0
for seed in range(24, 424):
    try:
        T.test_mul_relin_on_random_shapes(seed)
    except Exception as e:
        bad += 1
        print("seed", seed, T._case(seed), type(e).__name__, str(e)[:300], flush=True)
print("done, failures:", bad)
