#!/bin/bash
# tools/r5_chol_ab18.sh TAG -- round 5: is it the caller's stream that is special, or the fifth stream?  (chain and panels swap handles; one stream destroyed before the new one is made)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5chol_${1:-ab18}
mkdir -p $O
run() {   # label, env assignments...
  local label=$1; shift
  echo "== $label" | tee -a $O/ab.txt
  env RCN_LIB=tools/librcn_diag.so "$@" timeout -k 10 200 python3 tools/ba_run.py 1000 100000 5 2>$O/err_$label.txt | grep "^run [1-4]" | sed -e 's/.*(\([0-9.]* it\/s\)).*chol \([0-9.]*\) tri.*/\2 ms chol  \1/' | tr '\n' ';' | tee -a $O/ab.txt
  echo | tee -a $O/ab.txt
}
run default X=1
run swap_chain_panel RCN_CHOL_CHAIN_STREAM=4
run pg0_nopanel2_chain_hi RCN_CHOL_PGSTREAM=0 RCN_NO_PANEL2=1 RCN_CHOL_CHAIN_STREAM=1
run pg0_nopanel2_server RCN_CHOL_PGSTREAM=0 RCN_NO_PANEL2=1 RCN_CHOL_DIAG_SERVER=1 RCN_POLL_MODE=1
RCN_CHOL_CHAIN_STREAM=4 timeout -k 10 300 python3 tools/chol_device_timeline.py > $O/chol_timeline_swap.txt 2>&1; tail -1 $O/chol_timeline_swap.txt
