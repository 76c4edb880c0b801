#!/bin/bash
# Copies what tools/profile_c2.sh, profile_sq.sh and profile_logged.sh left under gpurun_out/ into profiles/ under the round's names.
#   bash tools/collect_profiles.sh r04
set -e
R=${1:?round prefix, e.g. r04}
cd "$(dirname "$0")/.."
G=gpurun_out P=profiles
last_line() { python3 -c "import sys; open(sys.argv[2], 'w').write(open(sys.argv[1]).read().strip().splitlines()[-1] + '\n')" "$1" "$2"; }
for fp in strict fma; do
    [ -d $G/prof_c2_$fp ] || continue
    cp $G/prof_c2_$fp/kt_kernel_stats.csv $P/${R}_kernel_stats_bench_c2_$fp.csv
    cp $G/prof_c2_$fp/kt_domain_stats.csv $P/${R}_domain_stats_c2_$fp.csv
    last_line $G/prof_c2_$fp/bench_kt.json $P/${R}_bench_c2_${fp}_under_rocprof.json
    cp $G/prof_c2_$fp/pmc_hbm_bytes_per_launch.json $P/${R}_pmc_hbm_bytes_per_launch_$fp.json
    cp $G/prof_c2_$fp/sq_counters_c2.json $P/${R}_sq_counters_c2_$fp.json
done
[ -f $G/prof_c2_strict/calib1.log ] && cat $G/prof_c2_strict/calib1.log $G/prof_c2_strict/calib2.log | grep -v "^[WEI]20" > $P/${R}_hbm_calib.log
for w in c3 c5; do for fp in strict fma; do
    [ -f $G/prof_${w}_$fp/sq_counters_$w.json ] || continue
    cp $G/prof_${w}_$fp/sq_counters_$w.json $P/${R}_sq_counters_${w}_$fp.json
    cp $G/prof_${w}_$fp/kt_kernel_stats.csv $P/${R}_kernel_stats_bench_${w}_$fp.csv
done; done
for w in c2_strict c3_strict c2_fma; do
    [ -f $G/prof_logged_$w/timing.json ] || continue
    cp $G/prof_logged_$w/timing.json $P/${R}_logged_timing_$w.json
    cp $G/prof_logged_$w/kt_kernel_stats.csv $P/${R}_kernel_stats_logged_$w.csv
done
python3 - <<PY
import glob, json, sys
sys.path.insert(0, "tools")
from kernel_sha import kernel_sources_sha256
sha = kernel_sources_sha256()
for f in sorted(glob.glob("profiles/${R}_*.json")):
    try:
        d = json.load(open(f))
    except Exception:
        continue
    h = d.get("kernel_sources_sha256") if isinstance(d, dict) else None
    print(("ok   " if h == sha else "STALE" if h else "nohash"), f)
PY
