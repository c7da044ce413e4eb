#!/bin/bash
# tools/r5_chol_ab8.sh TAG -- round 5: the chain's 2 g-row window (head rows and next super-diagonal block through the per-step latency kernels)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5chol_${1:-ab8}
mkdir -p $O
run() {   # label, env assignments...
  local label=$1; shift
  echo "== $label" | tee -a $O/ab.txt
  env RCN_LIB=tools/librcn_diag.so "$@" timeout -k 10 200 python3 tools/ba_run.py 1000 100000 5 2>$O/err_$label.txt | grep "^run [1-4]" | sed -e 's/.*(\([0-9.]* it\/s\)).*chol \([0-9.]*\) tri.*/\2 ms chol  \1/' | tr '\n' ';' | tee -a $O/ab.txt
  echo | tee -a $O/ab.txt
}
timeout -k 10 600 python -m pytest tests/test_ba_gpu.py -x -q 2>&1 | tail -3 | tee -a $O/ab.txt
run default X=1
run no_window RCN_CHOL_WINDOW=0
run min32 RCN_CHOL_TL_MIN=32
run min24 RCN_CHOL_TL_MIN=24
run min16 RCN_CHOL_TL_MIN=16
run min12 RCN_CHOL_TL_MIN=12
run min8 RCN_CHOL_TL_MIN=8
run min16_r16 RCN_CHOL_TL_MIN=16 RCN_RESERVED_CUS=16
RCN_CHOL_TL_MIN=16 timeout -k 10 300 python3 tools/chol_device_timeline.py > $O/chol_timeline_min16.txt 2>&1; tail -1 $O/chol_timeline_min16.txt
echo "== soak (diagnostic build, RCN_CHOL_TL_MIN=8), 100..400 cameras" | tee -a $O/ab.txt
RCN_LIB=tools/librcn_diag.so RCN_CHOL_TL_MIN=8 timeout -k 10 150 python3 tools/soak_ba_large.py 45 61 100 400 2>&1 | tail -1 | tee -a $O/ab.txt
echo "== soak: product build, 450..900 cameras" | tee -a $O/ab.txt
timeout -k 10 200 python3 tools/soak_ba_large.py 45 62 450 900 2>&1 | tail -1 | tee -a $O/ab.txt
