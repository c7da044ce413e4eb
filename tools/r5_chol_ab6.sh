#!/bin/bash
# tools/r5_chol_ab6.sh TAG -- round 5: reserved CUs with the latency-form head kernels; the far-start test on its own
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5chol_${1:-ab6}
mkdir -p $O
run() {   # label, env assignments...
  local label=$1; shift
  echo "== $label" | tee -a $O/ab.txt
  env RCN_LIB=tools/librcn_diag.so "$@" timeout -k 10 200 python3 tools/ba_run.py 1000 100000 5 2>$O/err_$label.txt | grep "^run [1-4]" | sed -e 's/.*(\([0-9.]* it\/s\)).*chol \([0-9.]*\) tri.*/\2 ms chol  \1/' | tr '\n' ';' | tee -a $O/ab.txt
  echo | tee -a $O/ab.txt
}
timeout -k 10 600 python -m pytest tests/test_ba_gpu.py -q -k "far_start" 2>&1 | tail -30 | tee -a $O/ab.txt
timeout -k 10 600 python -m pytest tests/test_ba_gpu.py -q -k "far_start" 2>&1 | tail -3 | tee -a $O/ab.txt
run default X=1
run r16 RCN_RESERVED_CUS=16
run r16_min24 RCN_RESERVED_CUS=16 RCN_CHOL_TL_MIN=24
run r24_min24 RCN_RESERVED_CUS=24 RCN_CHOL_TL_MIN=24
run r16_min16 RCN_RESERVED_CUS=16 RCN_CHOL_TL_MIN=16
run nomask RCN_NO_CU_MASK=1
run nomask_min24 RCN_NO_CU_MASK=1 RCN_CHOL_TL_MIN=24
echo "== soak: product build, 100..400 cameras (the right-looking regime alone)" | tee -a $O/ab.txt
timeout -k 10 200 python3 tools/soak_ba_large.py 60 41 100 400 2>&1 | tail -1 | tee -a $O/ab.txt
