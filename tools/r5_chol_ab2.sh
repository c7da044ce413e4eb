#!/bin/bash
# tools/r5_chol_ab2.sh TAG -- round 5: where the panel stream's kernels may run (diagnostic build, one process per setting)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5chol_${1:-ab2}
mkdir -p $O
run() {   # label, env assignments...
  local label=$1; shift
  echo "== $label" | tee -a $O/ab.txt
  env RCN_LIB=tools/librcn_diag.so "$@" timeout -k 10 200 python3 tools/ba_run.py 1000 100000 5 2>$O/err_$label.txt | grep "^run [1-4]" | sed -e 's/.*(\([0-9.]* it\/s\)).*chol \([0-9.]*\) tri.*/\2 ms chol  \1/' | tr '\n' ';' | tee -a $O/ab.txt
  echo | tee -a $O/ab.txt
}
run default X=1
run tl0 RCN_CHOL_TL=0
run mode0 RCN_PANEL_MODE=0
run mode0_tl0 RCN_PANEL_MODE=0 RCN_CHOL_TL=0
run mode2_r24 RCN_PANEL_MODE=2 RCN_RESERVED_CUS=24
run mode2_r24_tl0 RCN_PANEL_MODE=2 RCN_RESERVED_CUS=24 RCN_CHOL_TL=0
run mode1_r16 RCN_RESERVED_CUS=16
run mode1_r16_tl0 RCN_RESERVED_CUS=16 RCN_CHOL_TL=0
run tl8 RCN_CHOL_TL=8
run tl8_r16 RCN_CHOL_TL=8 RCN_RESERVED_CUS=16
run tl4_min40 RCN_CHOL_TL_MIN=40
run tl4_min48 RCN_CHOL_TL_MIN=48
run tl8_min40 RCN_CHOL_TL=8 RCN_CHOL_TL_MIN=40
echo "== cfg4 default / tl irrelevant" | tee -a $O/ab.txt
RCN_LIB=tools/librcn_diag.so timeout -k 10 200 python3 tools/ba_run.py 200 20000 5 2>/dev/null | tail -2 | tee -a $O/ab.txt
RCN_LIB=tools/librcn_diag.so RCN_PANEL_MODE=0 timeout -k 10 200 python3 tools/ba_run.py 200 20000 5 2>/dev/null | tail -2 | tee -a $O/ab.txt
timeout -k 10 300 python3 tools/chol_device_timeline.py > $O/chol_timeline.txt 2>&1; tail -1 $O/chol_timeline.txt
RCN_CHOL_TL=8 timeout -k 10 300 python3 tools/chol_device_timeline.py > $O/chol_timeline_tl8.txt 2>&1; tail -1 $O/chol_timeline_tl8.txt
RCN_CHOL_TL=0 timeout -k 10 300 python3 tools/chol_device_timeline.py > $O/chol_timeline_tl0.txt 2>&1; tail -1 $O/chol_timeline_tl0.txt
