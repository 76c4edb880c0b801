#!/usr/bin/env python3
"""Offline randomised differential sweep on the GPU: tests/test_differential_random_cpu.random_case() for a range of
seeds, the product path (default policy and, for DOPRI5 / DOP853, the lane-cooperative kernels) against the oracle, bit
for bit.  Usage (on the GPU box, from the repo root):  python tests/sweeps/gpu_random_sweep.py FIRST_SEED COUNT [report.json] [fma]
(a fourth argument `fma` runs the FMA arithmetic mode against liboracle_fma.so).
Prints a progress line every 500 seeds; exits non-zero on the first mismatch (the assertion names the seed)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.common import gpu_batch  # noqa: E402
from tests.test_differential_random_cpu import compare, random_case  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
FMA = len(sys.argv) > 4 and sys.argv[4] == "fma"
t0 = time.time()
n_coop = 0
for k, seed in enumerate(range(first, first + count)):
    compare(lambda rhs, y0, p, a, b, **kw: gpu_batch(rhs, y0, p, a, b, fast=FMA, **kw), seed, fma=FMA)
    if random_case(seed)[5]["method"] in ("DOPRI5", "DOP853"):
        compare(lambda rhs, y0, p, a, b, **kw: gpu_batch(rhs, y0, p, a, b, variant=3, fast=FMA, **kw), seed, fma=FMA)
        n_coop += 1
    if (k + 1) % 500 == 0:
        print(f"{k + 1} seeds ok ({n_coop} also through the cooperative kernels), {time.time() - t0:.0f} s", flush=True)
rep = {"fp_mode": "fma" if FMA else "strict", "first_seed": first, "seeds": count, "also_cooperative": n_coop, "mismatches": 0, "seconds": round(time.time() - t0, 1)}
print(json.dumps(rep))
if len(sys.argv) > 3:
    json.dump(rep, open(sys.argv[3], "w"))
