#!/bin/bash
# tools/fold_ablation.sh -- K1 at the SIFT and ORB shapes with the shipping fold (3 vector operations per accumulator element), a values-only fold
# (RCN_COARSE_ABL=2: 2 operations, results wrong by construction), the running minimum alone (RCN_COARSE_ABL=6: ONE operation -- round 5) and no fold at all
# (RCN_COARSE_ABL=1): what a cheaper exact fold could buy at most.
cd "$GRAFT_REPO_ROOT" || exit 1
export RCN_LIB=$PWD/tools/librcn_diag.so
for shape in "100 1500 128 12" "100 1500 32 12"; do
  for abl in 0 2 6 1; do
    echo -n "shape $shape  RCN_COARSE_ABL=$abl  "; RCN_COARSE_ABL=$abl python3 tools/k1_run.py $shape 2>&1 | tail -1
  done
done
