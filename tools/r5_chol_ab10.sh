#!/bin/bash
# tools/r5_chol_ab10.sh TAG -- round 5: the right-looking regime's thresholds again, now that the panel stream is unmasked
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5chol_${1:-ab10}
mkdir -p $O
run() {   # label, env assignments...
  local label=$1; shift
  echo "== $label" | tee -a $O/ab.txt
  env RCN_LIB=tools/librcn_diag.so "$@" timeout -k 10 200 python3 tools/ba_run.py 1000 100000 5 2>$O/err_$label.txt | grep "^run [1-4]" | sed -e 's/.*(\([0-9.]* it\/s\)).*chol \([0-9.]*\) tri.*/\2 ms chol  \1/' | tr '\n' ';' | tee -a $O/ab.txt
  echo | tee -a $O/ab.txt
}
run default X=1
run pipe16 RCN_CHOL_PIPE_MIN=16
run pipe24 RCN_CHOL_PIPE_MIN=24
run pipe48 RCN_CHOL_PIPE_MIN=48
run pipe99 RCN_CHOL_PIPE_MIN=99
run pair16 RCN_CHOL_PAIR_MIN=16
run pair32 RCN_CHOL_PAIR_MIN=32
run nopairs RCN_CHOL_GROUP=1
run default_again X=1
