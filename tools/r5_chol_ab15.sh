#!/bin/bash
# tools/r5_chol_ab15.sh TAG -- round 5: a fifth stream made everything slower twice -- streams sharing a hardware queue?  GPU_MAX_HW_QUEUES (the HIP runtime's, default 4)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5chol_${1:-ab15}
mkdir -p $O
run() {   # label, env assignments...
  local label=$1; shift
  echo "== $label" | tee -a $O/ab.txt
  env RCN_LIB=tools/librcn_diag.so "$@" timeout -k 10 200 python3 tools/ba_run.py 1000 100000 5 2>$O/err_$label.txt | grep "^run [1-4]" | sed -e 's/.*(\([0-9.]* it\/s\)).*chol \([0-9.]*\) tri.*/\2 ms chol  \1/' | tr '\n' ';' | tee -a $O/ab.txt
  echo | tee -a $O/ab.txt
}
run default X=1
run q8_default GPU_MAX_HW_QUEUES=8
run q8_chain_hi GPU_MAX_HW_QUEUES=8 RCN_CHOL_CHAIN_STREAM=1
run q8_server GPU_MAX_HW_QUEUES=8 RCN_CHOL_DIAG_SERVER=1
run q8_server_count GPU_MAX_HW_QUEUES=8 RCN_CHOL_DIAG_SERVER=1 RCN_POLL_MODE=1
run q8_server_chain GPU_MAX_HW_QUEUES=8 RCN_CHOL_DIAG_SERVER=1 RCN_CHOL_CHAIN_STREAM=1
run q2_default GPU_MAX_HW_QUEUES=2
GPU_MAX_HW_QUEUES=8 RCN_CHOL_DIAG_SERVER=1 timeout -k 10 300 python3 tools/chol_device_timeline.py > $O/chol_timeline_q8_server.txt 2>&1; tail -1 $O/chol_timeline_q8_server.txt
GPU_MAX_HW_QUEUES=8 timeout -k 10 300 python3 tools/chol_device_timeline.py > $O/chol_timeline_q8_default.txt 2>&1; tail -1 $O/chol_timeline_q8_default.txt
echo "== cfg4" | tee -a $O/ab.txt
RCN_LIB=tools/librcn_diag.so GPU_MAX_HW_QUEUES=8 timeout -k 10 200 python3 tools/ba_run.py 200 20000 5 2>/dev/null | tail -2 | tee -a $O/ab.txt
RCN_LIB=tools/librcn_diag.so GPU_MAX_HW_QUEUES=8 RCN_CHOL_DIAG_SERVER=1 timeout -k 10 200 python3 tools/ba_run.py 200 20000 5 2>/dev/null | tail -2 | tee -a $O/ab.txt
