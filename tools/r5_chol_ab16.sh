#!/bin/bash
# tools/r5_chol_ab16.sh TAG -- round 5: the fifth stream as a CU-masked stream (all CUs): does a hardware queue of its own take the slowdown away?
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5chol_${1:-ab16}
mkdir -p $O
run() {   # label, env assignments...
  local label=$1; shift
  echo "== $label" | tee -a $O/ab.txt
  env RCN_LIB=tools/librcn_diag.so "$@" timeout -k 10 200 python3 tools/ba_run.py 1000 100000 5 2>$O/err_$label.txt | grep "^run [1-4]" | sed -e 's/.*(\([0-9.]* it\/s\)).*chol \([0-9.]*\) tri.*/\2 ms chol  \1/' | tr '\n' ';' | tee -a $O/ab.txt
  echo | tee -a $O/ab.txt
}
run default X=1
run chain_fullmask RCN_CHOL_CHAIN_STREAM=3
run server_fullmask RCN_CHOL_DIAG_SERVER=1 RCN_DIAG_STREAM_PRIO=2
run server_fullmask_count RCN_CHOL_DIAG_SERVER=1 RCN_DIAG_STREAM_PRIO=2 RCN_POLL_MODE=1
RCN_CHOL_DIAG_SERVER=1 RCN_DIAG_STREAM_PRIO=2 timeout -k 10 300 python3 tools/chol_device_timeline.py > $O/chol_timeline_server_fullmask.txt 2>&1; tail -1 $O/chol_timeline_server_fullmask.txt
