#!/bin/bash
# tools/r5_match_run.sh TAG -- round 5, matcher side: same-box A/B of the pipeline-chunk size on the cfg-3 grid, the fold ablations of K1 at D = 128 / 32, the matcher suites
cd "$(dirname "$0")/.." || exit 1
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r5match_${1:-a}
mkdir -p $O
timeout -k 10 600 python3 tools/chunk_sweep.py 1000 4096 67108864 134217728 268435456 536870912 4294967296 134217728 > $O/chunk_sweep.txt 2>&1; grep chunk_rows $O/chunk_sweep.txt
timeout -k 10 300 bash tools/fold_ablation.sh > $O/fold_ablation.txt 2>&1; cat $O/fold_ablation.txt
timeout -k 10 900 python -m pytest tests/test_match_gpu.py tests/test_match_tiers_gpu.py tests/test_cfg3_gpu.py -x -q > $O/tests.txt 2>&1; tail -5 $O/tests.txt
