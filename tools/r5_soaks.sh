#!/bin/bash
# tools/r5_soaks.sh -- the round's randomised differential runs on the final sources (outputs under gpurun_out/r5soak/)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5soak
rm -rf $O; mkdir -p $O
timeout -k 10 400 python3 tools/soak_match.py 150 141 > $O/soak_match.txt 2>&1; tail -1 $O/soak_match.txt
timeout -k 10 200 python3 tools/soak_ba.py 60 142 > $O/soak_ba.txt 2>&1; tail -1 $O/soak_ba.txt
timeout -k 10 250 python3 tools/soak_ba_large.py 90 143 100 400 > $O/soak_ba_large_100_400.txt 2>&1; tail -1 $O/soak_ba_large_100_400.txt
timeout -k 10 300 python3 tools/soak_ba_large.py 120 144 450 900 > $O/soak_ba_large_450_900.txt 2>&1; tail -1 $O/soak_ba_large_450_900.txt
RCN_LIB=tools/librcn_diag.so RCN_CHOL_TL_MIN=8 timeout -k 10 200 python3 tools/soak_ba_large.py 60 145 100 400 > $O/soak_ba_large_two_level_small.txt 2>&1; tail -1 $O/soak_ba_large_two_level_small.txt
