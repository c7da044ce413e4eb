#!/bin/bash
# tools/r5_eval_ab.sh TAG -- round 5: a streaming kernel changed without moving a bit: tests, the five-solve hash, kernel times
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5eval_${1:-a}
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ba_gpu.py tests/test_ba_session_gpu.py -x -q > $O/tests.txt 2>&1 || { tail -40 $O/tests.txt; echo "BA tests failed"; exit 1; }
tail -2 $O/tests.txt
python3 tools/ba_bits_hash.py 2>/dev/null | tail -1 | tee $O/hash.txt
bash tools/r5_stream_prof.sh eval_${1:-a} > $O/prof.txt 2>&1; grep -v "^W2026\|^E2026" $O/prof.txt | tail -20
python3 bench.py --help > /dev/null 2>&1
