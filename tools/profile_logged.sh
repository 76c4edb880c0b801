#!/bin/bash
# rocprofv3 kernel-trace stats of the one-pass accepted-step log (tools/time_logged.py) on C2 / C3.  Run on the GPU box:
#   gpurun --timeout 900 -- 'bash tools/profile_logged.sh c2 strict'
set -u
WL=${1:-c2}
FP=${2:-strict}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_logged_${WL}_$FP
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/time_logged.py $WL --fp $FP --solves 10 > $O/timing.json 2> $O/timing.err; echo timing rc=$?
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- python3 $R/tools/time_logged.py $WL --fp $FP --solves 10 --only one > $O/kt_run.json 2> $O/kt.err; echo kt rc=$?
cat $O/timing.json
ls $O
