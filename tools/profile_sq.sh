#!/bin/bash
# SQ-counter pass for any bench workload: bash tools/profile_sq.sh c3 [strict|fma]   (on the GPU box, from the repo root)
set -u
W=${1:-c3}
FP=${2:-strict}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_${W}_$FP
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O -o sq -- python3 $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-fast --fp $FP > $O/b.log 2>&1; echo sq rc=$?
python3 $R/tools/pmc_sq.py $O/sq_counter_collection.csv $O/sq_counters_$W.json 12 > $O/sq.log 2>&1; echo post rc=$?   # 3 passes x (3 + 1) solves
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- python3 $R/bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-fast --fp $FP > $O/bench_kt.json 2> $O/kt.err; echo kt rc=$?
