#!/bin/bash
# tools/r5_chol_ab12.sh TAG -- round 5: why the resident diagonal workgroup slows the kernels beside it (how it polls, its stream's priority)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5chol_${1:-ab12}
mkdir -p $O
run() {   # label, env assignments...
  local label=$1; shift
  echo "== $label" | tee -a $O/ab.txt
  env RCN_LIB=tools/librcn_diag.so "$@" timeout -k 10 200 python3 tools/ba_run.py 1000 100000 5 2>$O/err_$label.txt | grep "^run [1-4]" | sed -e 's/.*(\([0-9.]* it\/s\)).*chol \([0-9.]*\) tri.*/\2 ms chol  \1/' | tr '\n' ';' | tee -a $O/ab.txt
  echo | tee -a $O/ab.txt
}
run server X=1
run server_count RCN_POLL_MODE=1
run server_count_sleep8 RCN_POLL_MODE=1 RCN_POLL_SLEEPS=8
run server_clock_sleep8 RCN_POLL_SLEEPS=8
run server_prio0 RCN_DIAG_STREAM_PRIO=0
run server_prio0_count RCN_DIAG_STREAM_PRIO=0 RCN_POLL_MODE=1
run launches RCN_CHOL_DIAG_SERVER=0
run launches_count RCN_CHOL_DIAG_SERVER=0 RCN_POLL_MODE=1
run server_hosttime RCN_CHOL_HOSTTIME=1
grep "enqueued" $O/err_server_hosttime.txt | tail -3 | tee -a $O/ab.txt
RCN_POLL_MODE=1 timeout -k 10 300 python3 tools/chol_device_timeline.py > $O/chol_timeline_server_count.txt 2>&1; tail -1 $O/chol_timeline_server_count.txt
