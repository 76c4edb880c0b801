#!/bin/bash
# Reproduces profiles/r01_*: rocprofv3 kernel-trace stats of bench.py (C2) and the HBM-byte counters, collected in
# separate --pmc passes as /opt/skills/guides/MI355X_MICROARCH.md prescribes.  Run on the GPU box from the repo root:
#   gpurun --timeout 900 -- 'bash tools/profile_c2.sh [strict|fma]'
set -u
FP=${1:-strict}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_c2_$FP
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
make -C $R/tools > $O/make.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- python3 $R/bench.py --steps 20 --no-cpu-baseline --no-fast --fp $FP > $O/bench_kt.json 2> $O/kt.err; echo kt rc=$?
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -o calib_fetch -- $R/tools/hbm_calib 16777216 3 > $O/calib1.log 2>&1; echo c1 rc=$?
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O -o calib_write -- $R/tools/hbm_calib 16777216 3 > $O/calib2.log 2>&1; echo c2 rc=$?
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -o bench_fetch -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fast --fp $FP > $O/b1.log 2>&1; echo b1 rc=$?
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O -o bench_write -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fast --fp $FP > $O/b2.log 2>&1; echo b2 rc=$?
python3 $R/tools/pmc_hbm.py $O/calib_fetch_counter_collection.csv $O/calib_write_counter_collection.csv \
        $O/bench_fetch_counter_collection.csv $O/bench_write_counter_collection.csv $O/pmc_hbm_bytes_per_launch.json; echo pmc rc=$?
ls $O | head -40
# SQ counters (their own pass): instructions per wave and where the wave cycles go
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O -o bench_sq -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fast --fp $FP > $O/b3.log 2>&1; echo b3 rc=$?
python3 $R/tools/pmc_sq.py $O/bench_sq_counter_collection.csv $O/sq_counters_c2.json 21 > $O/sq.log 2>&1; echo sq rc=$?   # 3 passes x (5 + 2) solves
