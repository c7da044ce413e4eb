#!/bin/bash
# tools/r5_chol_ab9.sh TAG -- round 5: keep the panel stream off the diagonal kernel's CUs (three tiers of CU masks)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5chol_${1:-ab9}
mkdir -p $O
run() {   # label, env assignments...
  local label=$1; shift
  echo "== $label" | tee -a $O/ab.txt
  env RCN_LIB=tools/librcn_diag.so "$@" timeout -k 10 200 python3 tools/ba_run.py 1000 100000 5 2>$O/err_$label.txt | grep "^run [1-4]" | sed -e 's/.*(\([0-9.]* it\/s\)).*chol \([0-9.]*\) tri.*/\2 ms chol  \1/' | tr '\n' ';' | tee -a $O/ab.txt
  echo | tee -a $O/ab.txt
}
run default X=1
run mode2_r16 RCN_PANEL_MODE=2 RCN_RESERVED_CUS=16
run mode2_r24 RCN_PANEL_MODE=2 RCN_RESERVED_CUS=24
run mode2_r32 RCN_PANEL_MODE=2 RCN_RESERVED_CUS=32
run mode2_r16_min24 RCN_PANEL_MODE=2 RCN_RESERVED_CUS=16 RCN_CHOL_TL_MIN=24
run mode2_r24_min24 RCN_PANEL_MODE=2 RCN_RESERVED_CUS=24 RCN_CHOL_TL_MIN=24
run mode2_r24_min16 RCN_PANEL_MODE=2 RCN_RESERVED_CUS=24 RCN_CHOL_TL_MIN=16
RCN_PANEL_MODE=2 RCN_RESERVED_CUS=24 timeout -k 10 300 python3 tools/chol_device_timeline.py > $O/chol_timeline_mode2_r24.txt 2>&1; tail -1 $O/chol_timeline_mode2_r24.txt
echo "== cfg4" | tee -a $O/ab.txt
RCN_LIB=tools/librcn_diag.so timeout -k 10 200 python3 tools/ba_run.py 200 20000 5 2>/dev/null | tail -2 | tee -a $O/ab.txt
RCN_LIB=tools/librcn_diag.so RCN_PANEL_MODE=2 RCN_RESERVED_CUS=24 timeout -k 10 200 python3 tools/ba_run.py 200 20000 5 2>/dev/null | tail -2 | tee -a $O/ab.txt
