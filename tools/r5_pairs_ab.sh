#!/bin/bash
# tools/r5_pairs_ab.sh -- the pair-list kernels at 1000 cameras (rocprofv3 stats of a solve that builds them) + tests + the five-solve hash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5pairs
rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_ba_gpu.py tests/test_ba_session_gpu.py -x -q > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -1 $O/tests.txt
python3 tools/ba_bits_hash.py 2>/dev/null | tail -1
rocprofv3 --kernel-trace --stats -d $O/t -o t --output-format csv -- python3 tools/ba_run.py 1000 100000 2 > $O/t.log 2>&1
f=$(find $O/t -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_pair" in r["Name"] or "k_scan" in r["Name"]:
        print("  %-22s %3s calls %9.1f us" % (r["Name"].split("(")[0], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf $O/t
python3 tools/ba_rebuild_run.py 2>/dev/null | tail -2
