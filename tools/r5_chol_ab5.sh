#!/bin/bash
# tools/r5_chol_ab5.sh TAG -- round 5: fused tail with prompt publication + two inverse buffers; latency-form head kernels; how far down the two-level regime pays
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5chol_${1:-ab5}
mkdir -p $O
timeout -k 10 120 ./tools/chol_kernels_check > $O/kernels_check.txt 2>&1 || { tail -20 $O/kernels_check.txt; echo "kernel check failed"; exit 1; }
tail -1 $O/kernels_check.txt
run() {   # label, env assignments...
  local label=$1; shift
  echo "== $label" | tee -a $O/ab.txt
  env RCN_LIB=tools/librcn_diag.so "$@" timeout -k 10 200 python3 tools/ba_run.py 1000 100000 5 2>$O/err_$label.txt | grep "^run [1-4]" | sed -e 's/.*(\([0-9.]* it\/s\)).*chol \([0-9.]*\) tri.*/\2 ms chol  \1/' | tr '\n' ';' | tee -a $O/ab.txt
  echo | tee -a $O/ab.txt
}
timeout -k 10 600 python -m pytest tests/test_ba_gpu.py -x -q 2>&1 | tail -3 | tee -a $O/ab.txt
run default X=1
run no_fuse RCN_CHOL_FUSE_TAIL=0
run head_pipe RCN_CHOL_HEAD_SMALL=0
run no_fuse_head_pipe RCN_CHOL_FUSE_TAIL=0 RCN_CHOL_HEAD_SMALL=0
run tl_min32 RCN_CHOL_TL_MIN=32
run tl_min24 RCN_CHOL_TL_MIN=24
run tl_min16 RCN_CHOL_TL_MIN=16
run tl_min24_nofuse RCN_CHOL_TL_MIN=24 RCN_CHOL_FUSE_TAIL=0
run tl8_min24 RCN_CHOL_TL=8 RCN_CHOL_TL_MIN=24
timeout -k 10 300 python3 tools/chol_device_timeline.py > $O/chol_timeline.txt 2>&1; tail -1 $O/chol_timeline.txt
RCN_CHOL_TL_MIN=24 timeout -k 10 300 python3 tools/chol_device_timeline.py > $O/chol_timeline_min24.txt 2>&1; tail -1 $O/chol_timeline_min24.txt
echo "== soak: two-level regime forced at small sizes (diagnostic build, RCN_CHOL_TL_MIN=8), 100..400 cameras" | tee -a $O/ab.txt
RCN_LIB=tools/librcn_diag.so RCN_CHOL_TL_MIN=8 timeout -k 10 150 python3 tools/soak_ba_large.py 45 31 100 400 2>&1 | tail -1 | tee -a $O/ab.txt
echo "== soak: product build, 450..900 cameras" | tee -a $O/ab.txt
timeout -k 10 200 python3 tools/soak_ba_large.py 45 32 450 900 2>&1 | tail -1 | tee -a $O/ab.txt
