#!/bin/bash
# tools/r5_chol_run.sh -- round 5: the factorisation's new kernels and schedule on the GPU box (checks first, then timings)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5chol_${1:-a}
mkdir -p $O
timeout -k 10 120 ./tools/chol_kernels_check > $O/kernels_check.txt 2>&1 || { tail -20 $O/kernels_check.txt; echo "kernel check failed"; exit 1; }
tail -3 $O/kernels_check.txt
timeout -k 10 900 python -m pytest tests/test_ba_gpu.py -x -q > $O/test_ba_gpu.txt 2>&1 || { tail -40 $O/test_ba_gpu.txt; echo "test_ba_gpu failed"; exit 1; }
tail -3 $O/test_ba_gpu.txt
timeout -k 10 300 python3 tools/ba_run.py 1000 100000 5 > $O/ba_run_cfg5.txt 2>&1 || { tail -20 $O/ba_run_cfg5.txt; exit 1; }
cat $O/ba_run_cfg5.txt
timeout -k 10 300 python3 tools/ba_run.py 200 20000 5 > $O/ba_run_cfg4.txt 2>&1; tail -3 $O/ba_run_cfg4.txt
timeout -k 10 300 python3 tools/chol_device_timeline.py > $O/chol_timeline.txt 2>&1; tail -2 $O/chol_timeline.txt
timeout -k 10 300 ./tools/gemm_nt_bench > $O/gemm_nt_bench.txt 2>&1; grep "full\|rolled" $O/gemm_nt_bench.txt | head -20
