#!/bin/bash
# tools/r5_chol_ab17.sh TAG -- round 5: four compute pipes?  Three streams plus the new one (no stream of its own for the first super-step's panel product)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5chol_${1:-ab17}
mkdir -p $O
run() {   # label, env assignments...
  local label=$1; shift
  echo "== $label" | tee -a $O/ab.txt
  env RCN_LIB=tools/librcn_diag.so "$@" timeout -k 10 200 python3 tools/ba_run.py 1000 100000 5 2>$O/err_$label.txt | grep "^run [1-4]" | sed -e 's/.*(\([0-9.]* it\/s\)).*chol \([0-9.]*\) tri.*/\2 ms chol  \1/' | tr '\n' ';' | tee -a $O/ab.txt
  echo | tee -a $O/ab.txt
}
run default X=1
run pg0 RCN_CHOL_PGSTREAM=0
run pg0_chain_hi RCN_CHOL_PGSTREAM=0 RCN_CHOL_CHAIN_STREAM=1
run pg0_server RCN_CHOL_PGSTREAM=0 RCN_CHOL_DIAG_SERVER=1
run pg0_server_count RCN_CHOL_PGSTREAM=0 RCN_CHOL_DIAG_SERVER=1 RCN_POLL_MODE=1
RCN_CHOL_PGSTREAM=0 RCN_CHOL_DIAG_SERVER=1 RCN_POLL_MODE=1 timeout -k 10 300 python3 tools/chol_device_timeline.py > $O/chol_timeline_pg0_server.txt 2>&1; tail -1 $O/chol_timeline_pg0_server.txt
echo "== cfg4" | tee -a $O/ab.txt
RCN_LIB=tools/librcn_diag.so timeout -k 10 200 python3 tools/ba_run.py 200 20000 5 2>/dev/null | tail -1 | tee -a $O/ab.txt
RCN_LIB=tools/librcn_diag.so RCN_CHOL_PGSTREAM=0 RCN_CHOL_DIAG_SERVER=1 RCN_POLL_MODE=1 timeout -k 10 200 python3 tools/ba_run.py 200 20000 5 2>/dev/null | tail -1 | tee -a $O/ab.txt
