#!/bin/bash
# tools/r5_chol_ab7.sh TAG -- round 5: a chain-bound super-step's small operations on the chain's own stream
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5chol_${1:-ab7}
mkdir -p $O
run() {   # label, env assignments...
  local label=$1; shift
  echo "== $label" | tee -a $O/ab.txt
  env RCN_LIB=tools/librcn_diag.so "$@" timeout -k 10 200 python3 tools/ba_run.py 1000 100000 5 2>$O/err_$label.txt | grep "^run [1-4]" | sed -e 's/.*(\([0-9.]* it\/s\)).*chol \([0-9.]*\) tri.*/\2 ms chol  \1/' | tr '\n' ';' | tee -a $O/ab.txt
  echo | tee -a $O/ab.txt
}
run default X=1
run min24_serial44 RCN_CHOL_TL_MIN=24 RCN_CHOL_TL_SERIAL=44
run min16_serial44 RCN_CHOL_TL_MIN=16 RCN_CHOL_TL_SERIAL=44
run min12_serial44 RCN_CHOL_TL_MIN=12 RCN_CHOL_TL_SERIAL=44
run min16_serial99 RCN_CHOL_TL_MIN=16 RCN_CHOL_TL_SERIAL=99
run min16_serial36 RCN_CHOL_TL_MIN=16 RCN_CHOL_TL_SERIAL=36
run min16_serial44_r16 RCN_CHOL_TL_MIN=16 RCN_CHOL_TL_SERIAL=44 RCN_RESERVED_CUS=16
RCN_CHOL_TL_MIN=16 RCN_CHOL_TL_SERIAL=44 timeout -k 10 300 python3 tools/chol_device_timeline.py > $O/chol_timeline_min16_s44.txt 2>&1; tail -1 $O/chol_timeline_min16_s44.txt
echo "== soak (diagnostic build, RCN_CHOL_TL_MIN=8 RCN_CHOL_TL_SERIAL=99), 100..400 cameras" | tee -a $O/ab.txt
RCN_LIB=tools/librcn_diag.so RCN_CHOL_TL_MIN=8 RCN_CHOL_TL_SERIAL=99 timeout -k 10 150 python3 tools/soak_ba_large.py 45 51 100 400 2>&1 | tail -1 | tee -a $O/ab.txt
