#!/bin/bash
# tools/r5_small_prof.sh -- where a small solve's time goes: kernel trace of 25-camera and 6-camera solves
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5small
rm -rf $O; mkdir -p $O
python3 tools/ba_small_run.py 25 1500 30 | tee $O/plain25.txt
python3 tools/ba_small_run.py 6 300 30 | tee $O/plain6.txt
rocprofv3 --kernel-trace --stats -d $O/t25 -o t --output-format csv -- python3 tools/ba_small_run.py 25 1500 20 > $O/t25.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/t6 -o t --output-format csv -- python3 tools/ba_small_run.py 6 300 20 > $O/t6.log 2>&1
for t in t25 t6; do f=$(find $O/$t -name "*kernel_stats.csv" | head -1); echo "== $t"; head -32 $f | cut -d, -f1-5 | cut -c1-110; cp $f $O/${t}_kernel_stats.csv; done
find $O -name "*kernel_trace.csv" -delete
