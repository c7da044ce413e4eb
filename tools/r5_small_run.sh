#!/bin/bash
# tools/r5_small_run.sh TAG -- the reference's own problem sizes: suites, then solve times at 3 .. 25 cameras
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5small_${1:-a}
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ba_gpu.py tests/test_ba_session_gpu.py tests/test_cfg1_plumbing.py tests/test_cpp_adapter.py -x -q 2>&1 | tail -4 | tee $O/tests.txt
for sh in "3 60" "6 300" "9 500" "12 700" "25 1500"; do timeout -k 10 120 python3 tools/ba_small_run.py $sh 30 | tee -a $O/times.txt; done
timeout -k 10 200 python3 tools/ba_run.py 1000 100000 4 2>/dev/null | tail -2 | tee -a $O/times.txt
timeout -k 10 200 python3 tools/ba_run.py 200 20000 4 2>/dev/null | tail -2 | tee -a $O/times.txt
