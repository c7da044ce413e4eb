#!/bin/bash
# tools/r5_chol_trace.sh TAG -- kernel traces of the cfg-5 factorisation's tail, diagonal blocks resident and as launches
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5chol_${1:-trace}
mkdir -p $O
export RCN_LIB=tools/librcn_diag.so RCN_POLL_MODE=1
rocprofv3 --kernel-trace -d $O/tsrv -o t --output-format csv -- python3 tools/ba_run.py 1000 100000 2 > $O/tsrv.log 2>&1
f=$(find $O/tsrv -name "*kernel_trace.csv" | head -1); python3 tools/chol_trace_tail.py $f 500 > $O/tail_server.txt; rm -rf $O/tsrv
export RCN_CHOL_DIAG_SERVER=0
rocprofv3 --kernel-trace -d $O/tlch -o t --output-format csv -- python3 tools/ba_run.py 1000 100000 2 > $O/tlch.log 2>&1
f=$(find $O/tlch -name "*kernel_trace.csv" | head -1); python3 tools/chol_trace_tail.py $f 500 > $O/tail_launches.txt; rm -rf $O/tlch
wc -l $O/tail_server.txt $O/tail_launches.txt
