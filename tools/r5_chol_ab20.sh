#!/bin/bash
# tools/r5_chol_ab20.sh TAG -- round 5: bulk updates of the right-looking regime behind the next diagonal block (the diagonal kernel no longer queues behind a bulk launch's dispatch)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5chol_${1:-ab20}
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ba_gpu.py -x -q > $O/test_ba_gpu.txt 2>&1 || { tail -40 $O/test_ba_gpu.txt; echo "test_ba_gpu failed"; exit 1; }
tail -2 $O/test_ba_gpu.txt
run() {   # label, env assignments...
  local label=$1; shift
  echo "== $label" | tee -a $O/ab.txt
  env RCN_LIB=tools/librcn_diag.so "$@" timeout -k 10 200 python3 tools/ba_run.py 1000 100000 5 2>$O/err_$label.txt | grep "^run [1-4]" | sed -e 's/.*(\([0-9.]* it\/s\)).*chol \([0-9.]*\) tri.*/\2 ms chol  \1/' | tr '\n' ';' | tee -a $O/ab.txt
  echo | tee -a $O/ab.txt
}
run behind X=1
run in_front RCN_CHOL_BULK_BEHIND=0
run behind_again X=1
run in_front_again RCN_CHOL_BULK_BEHIND=0
run behind_nopairs RCN_CHOL_GROUP=1
run behind_min24 RCN_CHOL_TL_MIN=24
timeout -k 10 300 python3 tools/chol_device_timeline.py > $O/chol_timeline_behind.txt 2>&1; tail -1 $O/chol_timeline_behind.txt
echo "== cfg4" | tee -a $O/ab.txt
RCN_LIB=tools/librcn_diag.so timeout -k 10 200 python3 tools/ba_run.py 200 20000 5 2>/dev/null | tail -2 | tee -a $O/ab.txt
RCN_LIB=tools/librcn_diag.so RCN_CHOL_BULK_BEHIND=0 timeout -k 10 200 python3 tools/ba_run.py 200 20000 5 2>/dev/null | tail -2 | tee -a $O/ab.txt
echo "== soak (diagnostic build, RCN_CHOL_TL_MIN=8), 100..400 cameras" | tee -a $O/ab.txt
RCN_LIB=tools/librcn_diag.so RCN_CHOL_TL_MIN=8 timeout -k 10 150 python3 tools/soak_ba_large.py 45 81 100 400 2>&1 | tail -1 | tee -a $O/ab.txt
echo "== soak: product build, 100..400 cameras" | tee -a $O/ab.txt
timeout -k 10 150 python3 tools/soak_ba_large.py 45 82 100 400 2>&1 | tail -1 | tee -a $O/ab.txt
echo "== soak: product build, 450..900 cameras" | tee -a $O/ab.txt
timeout -k 10 200 python3 tools/soak_ba_large.py 45 83 450 900 2>&1 | tail -1 | tee -a $O/ab.txt
