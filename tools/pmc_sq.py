#!/usr/bin/env python3
"""Aggregates a rocprofv3 --pmc SQ_* pass (counter_collection.csv) per kernel name.

Usage: pmc_sq.py <sq_counter_collection.csv> <out.json>
Per kernel: dispatches and the per-dispatch mean of every counter, plus a few quotients (VALU instructions per wave,
share of wave time that is issue-stall / parked; SQ_WAVE_CYCLES, SQ_WAIT_* and SQ_ACTIVE_INST_* count quad-cycles)."""
import csv, json, os, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_sha import kernel_sources_sha256   # noqa: E402

src, out = sys.argv[1:3]
solves = int(sys.argv[3]) if len(sys.argv) > 3 else None   # complete solves the profiled command ran (bench.py: 3 passes x (steps + warmup))
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(src)):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in acc.items():
    if "chunk_kernel" not in k and "init_kernel" not in k:
        continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    e = {"dispatches": len(next(iter(cs.values()))), "mean_per_dispatch": m}
    if solves:
        e["profiled_solves"] = solves
        e["per_solve"] = {c: sum(v) / solves for c, v in cs.items()}
    if m.get("SQ_WAVES"):
        e["valu_insts_per_wave"] = m.get("SQ_INSTS_VALU", 0.0) / m["SQ_WAVES"]
        e["salu_insts_per_wave"] = m.get("SQ_INSTS_SALU", 0.0) / m["SQ_WAVES"]
    if m.get("SQ_WAVE_CYCLES"):
        wc = m["SQ_WAVE_CYCLES"]
        e["share_of_wave_cycles"] = {c: m[c] / wc for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY") if c in m}
    res[k] = e
res["kernel_sources_sha256"] = kernel_sources_sha256()   # the kernels these counters belong to (bench.py checks it)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
