#!/bin/bash
# tools/r5_stream_prof.sh TAG "ENV=.. ENV=.." ... -- the streaming side of a cfg-5 LM iteration, kernel by kernel (rocprofv3 --stats), per setting
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5stream_${1:-a}; shift
mkdir -p $O
export RCN_LIB=tools/librcn_diag.so
n=0
for setting in "default" "$@"; do
  n=$((n+1))
  echo "== $setting" | tee -a $O/stats.txt
  if [ "$setting" != default ]; then export $setting; fi
  rocprofv3 --kernel-trace --stats -d $O/t$n -o t --output-format csv -- python3 tools/ba_run.py 1000 100000 3 > $O/t$n.log 2>&1
  grep "^run 2" $O/t$n.log | tee -a $O/stats.txt
  f=$(find $O/t$n -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY' | tee -a $O/stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0.0
for r in rows:
    nm = r["Name"].split("(")[0].replace("void ", "")
    if any(k in nm for k in ("k_gemm", "k_ring_gate", "k_chol", "k_sinv", "k_pair", "k_scan", "rocclr", "k_ba_cam_rot")):
        continue
    calls, avg = int(r["Calls"]), float(r["AverageNs"]) / 1e3
    per_it = avg * calls / 9.0            # three solves of three iterations
    tot += per_it
    print("  %-34s %5d calls  %8.1f us  (%7.1f us per iteration)" % (nm[:34], calls, avg, per_it))
print("  streaming kernels per iteration: %.1f us" % tot)
PY
  rm -rf $O/t$n
  if [ "$setting" != default ]; then unset ${setting%%=*}; fi
done
