#!/bin/bash
# tools/r5_chol_ab3.sh TAG -- round 5: what the bulk update loses beside the other streams (diagnostic build, one process per setting)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5chol_${1:-ab3}
mkdir -p $O
run() {   # label, env assignments...
  local label=$1; shift
  echo "== $label" | tee -a $O/ab.txt
  env RCN_LIB=tools/librcn_diag.so "$@" timeout -k 10 200 python3 tools/ba_run.py 1000 100000 5 2>$O/err_$label.txt | grep "^run [1-4]" | sed -e 's/.*(\([0-9.]* it\/s\)).*chol \([0-9.]*\) tri.*/\2 ms chol  \1/' | tr '\n' ';' | tee -a $O/ab.txt
  echo | tee -a $O/ab.txt
}
run default X=1
run pg_noprio RCN_CHOL_PG_PRIO=0
run r16 RCN_RESERVED_CUS=16
run safe RCN_CHOL_SAFE=1
run tl8_min48 RCN_CHOL_TL=8 RCN_CHOL_TL_MIN=48
timeout -k 10 300 python3 tools/chol_device_timeline.py > $O/chol_timeline.txt 2>&1; tail -1 $O/chol_timeline.txt
RCN_CHOL_SAFE=1 timeout -k 10 300 python3 tools/chol_device_timeline.py > $O/chol_timeline_safe.txt 2>&1; tail -1 $O/chol_timeline_safe.txt
RCN_CHOL_SAFE=1 RCN_CHOL_TL=0 timeout -k 10 300 python3 tools/chol_device_timeline.py > $O/chol_timeline_safe_tl0.txt 2>&1; tail -1 $O/chol_timeline_safe_tl0.txt
echo "== soak: two-level regime forced at small sizes (diagnostic build, RCN_CHOL_TL_MIN=8), 100..400 cameras" | tee -a $O/ab.txt
RCN_LIB=tools/librcn_diag.so RCN_CHOL_TL_MIN=8 timeout -k 10 150 python3 tools/soak_ba_large.py 60 11 100 400 2>&1 | tail -2 | tee -a $O/ab.txt
echo "== soak: product build, 450..900 cameras" | tee -a $O/ab.txt
timeout -k 10 200 python3 tools/soak_ba_large.py 90 12 450 900 2>&1 | tail -2 | tee -a $O/ab.txt
timeout -k 10 600 python -m pytest tests/test_ba_gpu.py tests/test_ba_session_gpu.py -x -q 2>&1 | tail -3 | tee -a $O/ab.txt
