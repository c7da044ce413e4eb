#!/bin/bash
# tools/r5_chol_ab.sh TAG -- round 5: A/B of the factorisation's schedule switches (diagnostic build, one process per setting)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5chol_${1:-ab}
mkdir -p $O
timeout -k 10 120 ./tools/chol_kernels_check > $O/kernels_check.txt 2>&1 || { tail -20 $O/kernels_check.txt; echo "kernel check failed"; exit 1; }
run() {   # label, env assignments...
  local label=$1; shift
  echo "== $label" | tee -a $O/ab.txt
  env RCN_LIB=tools/librcn_diag.so "$@" timeout -k 10 200 python3 tools/ba_run.py 1000 100000 5 2>$O/err_$label.txt | grep "^run [1-4]" | sed -e 's/.*(\([0-9.]* it\/s\)).*chol \([0-9.]*\) tri.*/\2 ms chol  \1/' | tr '\n' ';' | tee -a $O/ab.txt
  echo | tee -a $O/ab.txt
  grep -h "factorise:" $O/err_$label.txt | tail -2 | tee -a $O/ab.txt
}
run default X=1
run tl0 RCN_CHOL_TL=0
run gate_in_kernel RCN_CHOL_GATE_IN_KERNEL=1
run pg_on_b RCN_CHOL_PGSTREAM=0
run hosttime RCN_CHOL_HOSTTIME=1
run tl8 RCN_CHOL_TL=8
run tl8_min20 RCN_CHOL_TL=8 RCN_CHOL_TL_MIN=20
run tl4_min20 RCN_CHOL_TL_MIN=20
run tl4_min40 RCN_CHOL_TL_MIN=40
run tl0_hosttime RCN_CHOL_TL=0 RCN_CHOL_HOSTTIME=1
echo "== cfg4 (product build)" | tee -a $O/ab.txt
timeout -k 10 200 python3 tools/ba_run.py 200 20000 5 2>/dev/null | tail -2 | tee -a $O/ab.txt
echo "== cfg5 (product build)" | tee -a $O/ab.txt
timeout -k 10 200 python3 tools/ba_run.py 1000 100000 5 2>/dev/null | tail -3 | tee -a $O/ab.txt
timeout -k 10 300 python3 tools/chol_device_timeline.py > $O/chol_timeline.txt 2>&1; tail -1 $O/chol_timeline.txt
RCN_CHOL_TL=8 timeout -k 10 300 python3 tools/chol_device_timeline.py > $O/chol_timeline_tl8.txt 2>&1; tail -1 $O/chol_timeline_tl8.txt
