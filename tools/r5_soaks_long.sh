#!/bin/bash
# tools/r5_soaks_long.sh -- longer randomised runs on the final sources (other seeds than tools/r5_soaks.sh)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5soak_long
rm -rf $O; mkdir -p $O
timeout -k 10 700 python3 tools/soak_ba_large.py 480 244 450 900 > $O/soak_ba_large_450_900.txt 2>&1; tail -1 $O/soak_ba_large_450_900.txt
timeout -k 10 400 python3 tools/soak_ba_large.py 240 243 100 400 > $O/soak_ba_large_100_400.txt 2>&1; tail -1 $O/soak_ba_large_100_400.txt
timeout -k 10 400 python3 tools/soak_match.py 200 241 > $O/soak_match.txt 2>&1; tail -1 $O/soak_match.txt
