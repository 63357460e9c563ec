"""One-off extension of tests/test_gpu_fuzz.py: the same gate on seeds the suite does not hold.
   usage: python tools/fuzz_sweep.py <first seed> <count>"""
import os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_fuzz import test_random_scene_matches_oracle as gate
first, count = int(sys.argv[1]), int(sys.argv[2])
bad = []
t0 = time.perf_counter()
for k, seed in enumerate(range(first, first + count)):
    try:
        gate(seed)
    except AssertionError:
        bad.append(seed)
        print("seed", seed, "FAILED:", traceback.format_exc().splitlines()[-3:], flush=True)
    if (k + 1) % 25 == 0:
        print(f"{k + 1} scenes, {len(bad)} failures, {time.perf_counter() - t0:.1f} s", flush=True)
print("done:", count, "scenes, failures:", bad, flush=True)
sys.exit(1 if bad else 0)
