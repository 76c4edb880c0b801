#!/usr/bin/env python3
"""Post-process rocprofv3 --pmc runs into profiles/pmc_hbm_bytes_per_launch.json.

Usage: pmc_hbm.py <calib_fetch.csv> <calib_write.csv> <bench_fetch.csv> <bench_write.csv> <out.json>
Each csv is a rocprofv3 `*_counter_collection.csv` (one row per dispatch and counter).
FETCH_SIZE / WRITE_SIZE are in KiB-units per the guide (hbm_bytes = counter * 1024); the calibration kernel
(tools/hbm_calib.hip, the integrator's 8-B-per-lane SoA pattern, known byte count) gives the correction factors.
"""
import csv, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_sha import kernel_sources_sha256   # noqa: E402

def per_kernel(path, counter):
    d = {}
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        d.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return d

cf, cw, bf, bw, out = sys.argv[1:6]
known = float(sys.argv[6]) if len(sys.argv) > 6 else (1 << 24) * 16 * 8.0
cal_f = per_kernel(cf, "FETCH_SIZE"); cal_w = per_kernel(cw, "WRITE_SIZE")
kf = [k for k in cal_f if "calib" in k][0]; kw = [k for k in cal_w if "calib" in k][0]
f_meas = sum(cal_f[kf]) / len(cal_f[kf]) * 1024.0
w_meas = sum(cal_w[kw]) / len(cal_w[kw]) * 1024.0
corr_f, corr_w = known / f_meas, known / w_meas
res = {"calibration": {"known_bytes_each_way": known, "FETCH_SIZE_bytes_raw": f_meas, "WRITE_SIZE_bytes_raw": w_meas,
                       "fetch_correction": corr_f, "write_correction": corr_w,
                       "pattern": "8 B per lane, SoA-coalesced (tools/hbm_calib.hip)"}}
bf_, bw_ = per_kernel(bf, "FETCH_SIZE"), per_kernel(bw, "WRITE_SIZE")
kern = {}
for k in bf_:
    if "chunk_kernel" in k or "init_kernel" in k or "chunk_body" in k:
        f = sum(bf_[k]) / len(bf_[k]) * 1024.0 * corr_f
        w = sum(bw_.get(k, [0.0])) / max(len(bw_.get(k, [0.0])), 1) * 1024.0 * corr_w
        kern[k] = {"launches": len(bf_[k]), "fetch_bytes_per_launch": f, "write_bytes_per_launch": w, "hbm_bytes_per_launch": f + w}
res["kernels"] = kern
ck = [k for k in kern if "chunk_kernel" in k]   # chunk_kernel_t<..> and coop_chunk_kernel<..>: every stepping launch
if ck:
    n = sum(kern[k]["launches"] for k in ck)
    res["hbm_bytes_per_launch"] = sum(kern[k]["hbm_bytes_per_launch"] * kern[k]["launches"] for k in ck) / n
    res["kernel"] = " + ".join(ck)
    res["stepping_launches"] = n
res["kernel_sources_sha256"] = kernel_sources_sha256()   # the kernels these counters belong to (bench.py checks it)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
