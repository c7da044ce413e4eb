#!/bin/bash
# tools/r5_small_ab.sh TAG -- round 5: small solves (the reference's own regime) with the step's scalars through the pinned mirror vs copy + synchronisation
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5small_${1:-ab}
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ba_gpu.py tests/test_ba_session_gpu.py -x -q > $O/tests.txt 2>&1 || { tail -40 $O/tests.txt; echo "BA tests failed"; exit 1; }
tail -2 $O/tests.txt
for rep in 1 2; do
for nc in 3 6 12 25; do
  np=$((nc * 60))
  echo -n "mirror  : " | tee -a $O/ab.txt; RCN_LIB=tools/librcn_diag.so timeout -k 10 100 python3 tools/ba_small_run.py $nc $np 40 2>/dev/null | tail -1 | tee -a $O/ab.txt
  echo -n "copy    : " | tee -a $O/ab.txt; RCN_LIB=tools/librcn_diag.so RCN_BA_MIRROR=0 timeout -k 10 100 python3 tools/ba_small_run.py $nc $np 40 2>/dev/null | tail -1 | tee -a $O/ab.txt
done
done
