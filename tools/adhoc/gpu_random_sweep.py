"""ad hoc: the randomised differential test over a long seed range on the GPU (not part of the suite)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_differential_random_cpu import compare, random_case
from tests.common import gpu_batch
lo, hi = int(sys.argv[1]), int(sys.argv[2])
t = time.time(); bad = []
for seed in range(lo, hi):
    try:
        compare(lambda rhs, y0, p, t0, t1, **kw: gpu_batch(rhs, y0, p, t0, t1, **kw), seed)
        if random_case(seed)[5]["method"] in ("DOPRI5", "DOP853"):
            compare(lambda rhs, y0, p, t0, t1, **kw: gpu_batch(rhs, y0, p, t0, t1, variant=3, **kw), seed)
    except AssertionError as e:
        bad.append((seed, str(e)[:300]))
        if len(bad) >= 5: break
    if seed % 250 == 0: print("seed", seed, "elapsed", round(time.time() - t, 1), "bad", len(bad), flush=True)
print("done", lo, hi, "elapsed", round(time.time() - t, 1), "mismatches", len(bad), flush=True)
for b in bad: print(b)
