#!/bin/bash
# tools/r5_chol_ab11.sh TAG -- round 5: the diagonal blocks in ONE resident workgroup (k_chol_diag_server) against a launch per block
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5chol_${1:-ab11}
mkdir -p $O
timeout -k 10 120 ./tools/chol_kernels_check > $O/kernels_check.txt 2>&1 || { tail -20 $O/kernels_check.txt; echo "kernel check failed"; exit 1; }
tail -2 $O/kernels_check.txt
timeout -k 10 900 python -m pytest tests/test_ba_gpu.py -x -q > $O/test_ba_gpu.txt 2>&1 || { tail -40 $O/test_ba_gpu.txt; echo "test_ba_gpu failed"; exit 1; }
tail -3 $O/test_ba_gpu.txt
run() {   # label, env assignments...
  local label=$1; shift
  echo "== $label" | tee -a $O/ab.txt
  env RCN_LIB=tools/librcn_diag.so "$@" timeout -k 10 200 python3 tools/ba_run.py 1000 100000 5 2>$O/err_$label.txt | grep "^run [1-4]" | sed -e 's/.*(\([0-9.]* it\/s\)).*chol \([0-9.]*\) tri.*/\2 ms chol  \1/' | tr '\n' ';' | tee -a $O/ab.txt
  echo | tee -a $O/ab.txt
}
run server X=1
run launches RCN_CHOL_DIAG_SERVER=0
run server_min24 RCN_CHOL_TL_MIN=24
run server_min16 RCN_CHOL_TL_MIN=16
run server_nopairs RCN_CHOL_GROUP=1
run server_again X=1
timeout -k 10 300 python3 tools/chol_device_timeline.py > $O/chol_timeline_server.txt 2>&1; tail -1 $O/chol_timeline_server.txt
echo "== cfg4" | tee -a $O/ab.txt
RCN_LIB=tools/librcn_diag.so timeout -k 10 200 python3 tools/ba_run.py 200 20000 5 2>/dev/null | tail -2 | tee -a $O/ab.txt
RCN_LIB=tools/librcn_diag.so RCN_CHOL_DIAG_SERVER=0 timeout -k 10 200 python3 tools/ba_run.py 200 20000 5 2>/dev/null | tail -2 | tee -a $O/ab.txt
echo "== soak (diagnostic build, RCN_CHOL_TL_MIN=8), 100..400 cameras" | tee -a $O/ab.txt
RCN_LIB=tools/librcn_diag.so RCN_CHOL_TL_MIN=8 timeout -k 10 150 python3 tools/soak_ba_large.py 45 71 100 400 2>&1 | tail -1 | tee -a $O/ab.txt
echo "== soak: product build, 100..400 cameras" | tee -a $O/ab.txt
timeout -k 10 150 python3 tools/soak_ba_large.py 45 72 100 400 2>&1 | tail -1 | tee -a $O/ab.txt
echo "== soak: product build, 450..900 cameras" | tee -a $O/ab.txt
timeout -k 10 200 python3 tools/soak_ba_large.py 45 73 450 900 2>&1 | tail -1 | tee -a $O/ab.txt
