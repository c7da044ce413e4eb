#!/bin/bash
# tools/chol_ab.sh ENV v1 v2 ... -- A/B of one diagnostic-build switch on the cfg-5 solve, ONE PROCESS PER VALUE (tools/chol_sweep.py keeps
# several contexts alive in one process: once a switch adds a stream, their hardware queues are shared and its figures are not the switch's).
# Two passes over the values; prints chol ms per iteration and it/s of runs 1.. of each process.
cd "$(dirname "$0")/.." || exit 1
export RCN_LIB=tools/librcn_diag.so
env=$1; shift
for pass in 1 2; do
  for v in "$@"; do
    echo "== $env=$v (pass $pass)"
    env "$env=$v" timeout -k 10 120 python3 tools/ba_run.py 1000 100000 5 2>/dev/null | grep "^run [1-4]" | sed -e 's/.*(\([0-9.]* it\/s\)).*chol \([0-9.]*\) tri.*/\2 ms chol  \1/' | tr '\n' ';'
    echo
  done
done
