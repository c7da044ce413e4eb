#!/bin/bash
# tools/profile_round.sh [TAG] -- every rocprofv3 pass behind profiles/<TAG>_* (TAG = r05 by default; run on the MI355X box from
# the repo root:  gpurun -- 'RCN_GIT_HEAD=$(git rev-parse HEAD) bash tools/profile_round.sh r05' -- .git does not travel, the head is handed over).  Counter passes are separate runs with --pmc only (no trace domains),
# the program directly after `--`.  Raw output goes to gpurun_out/prof_<TAG>/, tools/pmc_report.py condenses it.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
R=${1:-r05}
LEGS=${2:-all}      # all | match | ba  (the two halves fit a 20-minute GPU call each; tools/pmc_report.py then runs on the merged output, here or on the CPU box)
O=gpurun_out/prof_$R
[ "$LEGS" = all ] && rm -rf "$O"
mkdir -p "$O"
run() { echo "== $*"; "$@" || echo "   (rc=$?)"; }
if [ "$LEGS" != ba ]; then
# ---- kernel time: the whole default bench (cfg3 headline + cfg2 + BA legs), no CPU baseline
run rocprofv3 --kernel-trace --stats -d $O/trace_bench -o t --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/trace_bench.log 2>&1
# ---- kernel time of the dominant kernel at the headline shape alone (cfg3: every launch in this trace is a 1000 x 4096 x 256 launch)
run rocprofv3 --kernel-trace --stats -d $O/trace_k1_cfg3 -o t --output-format csv -- python3 tools/k1_run.py 1000 4096 256 2 > $O/trace_k1_cfg3.log 2>&1
# ---- K1 counters at cfg2 shape (100 x 2048 x 256) and at the SIFT shape (100 x 1500 x 128)
for shape in "100 2048 256 6" "100 1500 128 6"; do
  tag=$(echo $shape | tr ' ' '_')
  run rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/k1_sq_$tag -o p --output-format csv -- python3 tools/k1_run.py $shape > $O/k1_sq_$tag.log 2>&1
  run rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM -d $O/k1_lds_$tag -o p --output-format csv -- python3 tools/k1_run.py $shape > $O/k1_lds_$tag.log 2>&1
  run rocprofv3 --pmc FETCH_SIZE -d $O/k1_fetch_$tag -o p --output-format csv -- python3 tools/k1_run.py $shape > $O/k1_fetch_$tag.log 2>&1
  run rocprofv3 --pmc WRITE_SIZE -d $O/k1_write_$tag -o p --output-format csv -- python3 tools/k1_run.py $shape > $O/k1_write_$tag.log 2>&1
done
# ---- K1 traffic at the headline workload (cfg3: 1000 x 4096 x 256, one launch of ~3.2 s)
run rocprofv3 --pmc FETCH_SIZE -d $O/k1_fetch_1000_4096_256 -o p --output-format csv -- python3 tools/k1_run.py 1000 4096 256 1 > $O/k1_fetch_cfg3.log 2>&1
run rocprofv3 --pmc WRITE_SIZE -d $O/k1_write_1000_4096_256 -o p --output-format csv -- python3 tools/k1_run.py 1000 4096 256 1 > $O/k1_write_cfg3.log 2>&1
run rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/k1_sq_1000_4096_256 -o p --output-format csv -- python3 tools/k1_run.py 1000 4096 256 1 > $O/k1_sq_cfg3.log 2>&1
run rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM -d $O/k1_lds_1000_4096_256 -o p --output-format csv -- python3 tools/k1_run.py 1000 4096 256 1 > $O/k1_lds_cfg3.log 2>&1
# ---- the top-2 fold as a ceiling at the SIFT shape (diagnostic build): shipping fold (3 vector operations per accumulator element), values only (2), none
for abl in 0 2 6 1; do
  RCN_LIB=$PWD/tools/librcn_diag.so RCN_COARSE_ABL=$abl rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES -d $O/fold_abl$abl -o p --output-format csv -- python3 tools/k1_run.py 100 1500 128 6 > $O/fold_abl$abl.log 2>&1 || echo "   (rc=$?)"
done
fi
if [ "$LEGS" != match ]; then
# ---- BA cfg 5: kernel time and the factorisation chain's counters
run rocprofv3 --kernel-trace --stats -d $O/trace_ba5 -o t --output-format csv -- python3 tools/ba_run.py 1000 100000 2 > $O/trace_ba5.log 2>&1
run rocprofv3 --kernel-trace --stats -d $O/trace_ba4 -o t --output-format csv -- python3 tools/ba_run.py 200 20000 2 > $O/trace_ba4.log 2>&1
run rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/ba5_sq -o p --output-format csv -- python3 tools/ba_run.py 1000 100000 1 > $O/ba5_sq.log 2>&1
run rocprofv3 --pmc FETCH_SIZE -d $O/ba5_fetch -o p --output-format csv -- python3 tools/ba_run.py 1000 100000 1 > $O/ba5_fetch.log 2>&1
run rocprofv3 --pmc WRITE_SIZE -d $O/ba5_write -o p --output-format csv -- python3 tools/ba_run.py 1000 100000 1 > $O/ba5_write.log 2>&1
fi
# keep what the report needs small: counter CSVs and the *_kernel_stats.csv of the traces
find $O -name "*kernel_trace.csv" -size +4M -delete
du -sh $O
find $O -name "*kernel_trace.csv" -size +1M -delete
if [ "$LEGS" = all ]; then mkdir -p gpurun_out/profiles_$R && python3 tools/pmc_report.py $O gpurun_out/profiles_$R $R && ls -la gpurun_out/profiles_$R; fi
