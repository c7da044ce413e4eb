#!/bin/bash
# tools/r5_chol_ab21.sh TAG -- round 5: the chain's next tiles carved out of the right-looking regime's updates (RCN_CHOL_CARVE = rows from which on)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5chol_${1:-ab21}
mkdir -p $O
run() {   # label, env assignments...
  local label=$1; shift
  echo "== $label" | tee -a $O/ab.txt
  env RCN_LIB=tools/librcn_diag.so "$@" timeout -k 10 200 python3 tools/ba_run.py 1000 100000 5 2>$O/err_$label.txt | grep "^run [1-4]" | sed -e 's/.*(\([0-9.]* it\/s\)).*chol \([0-9.]*\) tri.*/\2 ms chol  \1/' | tr '\n' ';' | tee -a $O/ab.txt
  echo | tee -a $O/ab.txt
}
run carve0 X=1
run carve26 RCN_CHOL_CARVE=26
run carve30 RCN_CHOL_CARVE=30
run carve34 RCN_CHOL_CARVE=34
run carve42 RCN_CHOL_CARVE=42
run carve99 RCN_CHOL_CARVE=99
run carve0_again X=1
RCN_CHOL_CARVE=34 timeout -k 10 300 python3 tools/chol_device_timeline.py > $O/chol_timeline_carve34.txt 2>&1; tail -1 $O/chol_timeline_carve34.txt
echo "== cfg4" | tee -a $O/ab.txt
RCN_LIB=tools/librcn_diag.so timeout -k 10 200 python3 tools/ba_run.py 200 20000 5 2>/dev/null | tail -1 | tee -a $O/ab.txt
RCN_LIB=tools/librcn_diag.so RCN_CHOL_CARVE=99 timeout -k 10 200 python3 tools/ba_run.py 200 20000 5 2>/dev/null | tail -1 | tee -a $O/ab.txt
echo "== soak (diagnostic build, RCN_CHOL_CARVE=99), 100..400 cameras" | tee -a $O/ab.txt
RCN_LIB=tools/librcn_diag.so RCN_CHOL_CARVE=99 timeout -k 10 150 python3 tools/soak_ba_large.py 45 91 100 400 2>&1 | tail -1 | tee -a $O/ab.txt
