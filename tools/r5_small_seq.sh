#!/bin/bash
# tools/r5_small_seq.sh -- the launch sequence of one 25-camera and one 6-camera solve (kernel + memory-copy trace)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5small_seq
rm -rf $O; mkdir -p $O
for nc in 25 6; do
  np=$((nc * 60))
  rocprofv3 --kernel-trace --memory-copy-trace -d $O/t$nc -o t --output-format csv -- python3 tools/ba_small_run.py $nc $np 4 > $O/t$nc.log 2>&1
  k=$(find $O/t$nc -name "*kernel_trace.csv" | head -1); m=$(find $O/t$nc -name "*memory_copy_trace.csv" | head -1)
  python3 tools/small_trace_seq.py $k $m > $O/seq$nc.txt 2>$O/seq$nc.err
  wc -l $O/seq$nc.txt; tail -2 $O/t$nc.log
  rm -rf $O/t$nc
done
