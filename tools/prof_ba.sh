#!/bin/bash
# tools/prof_ba.sh TAG NC NPTS NBLK -- kernel-time profile of one BA config (rocprofv3 --kernel-trace --stats), summary to gpurun_out/
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-ba5}; NC=${2:-1000}; NP=${3:-100000}; NBLK=${4:-79}
O=gpurun_out/prof_$TAG
rm -rf "$O"; mkdir -p "$O"
rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python3 tools/ba_run.py $NC $NP 2 > $O/trace.log 2>&1
grep "^run" $O/trace.log
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${TAG}_kernel_stats.csv
TIMELINE_STEPS=${TIMELINE_STEPS:-} python3 tools/chol_timeline2.py $(find $O -name "*kernel_trace.csv" | head -1) $NBLK
find $O -name "*kernel_trace.csv" -size +8M -delete
head -12 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-120
