#!/bin/bash
# tools/r5_chol_ab14.sh TAG -- round 5: the chain on a highest-priority stream of the library's own (does the diagonal kernel still queue behind a bulk launch's dispatch?)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5chol_${1:-ab14}
mkdir -p $O
run() {   # label, env assignments...
  local label=$1; shift
  echo "== $label" | tee -a $O/ab.txt
  env RCN_LIB=tools/librcn_diag.so "$@" timeout -k 10 200 python3 tools/ba_run.py 1000 100000 5 2>$O/err_$label.txt | grep "^run [1-4]" | sed -e 's/.*(\([0-9.]* it\/s\)).*chol \([0-9.]*\) tri.*/\2 ms chol  \1/' | tr '\n' ';' | tee -a $O/ab.txt
  echo | tee -a $O/ab.txt
}
run default X=1
run chain_hi RCN_CHOL_CHAIN_STREAM=1
run chain_own_normal RCN_CHOL_CHAIN_STREAM=2
run default_again X=1
run chain_hi_again RCN_CHOL_CHAIN_STREAM=1
RCN_CHOL_CHAIN_STREAM=1 timeout -k 10 300 python3 tools/chol_device_timeline.py > $O/chol_timeline_chain_hi.txt 2>&1; tail -1 $O/chol_timeline_chain_hi.txt
echo "== cfg4" | tee -a $O/ab.txt
RCN_LIB=tools/librcn_diag.so timeout -k 10 200 python3 tools/ba_run.py 200 20000 5 2>/dev/null | tail -2 | tee -a $O/ab.txt
RCN_LIB=tools/librcn_diag.so RCN_CHOL_CHAIN_STREAM=1 timeout -k 10 200 python3 tools/ba_run.py 200 20000 5 2>/dev/null | tail -2 | tee -a $O/ab.txt
