#!/bin/bash
# tools/diag_gemm_clock.sh -- effective clock and MFMA-pipe share of the bulk f64 kernel's variants (tools/gemm_nt_bench) and of the bare MFMA loop
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/gclk; rm -rf $O; mkdir -p $O
./tools/gemm_nt_bench > $O/plain.txt 2>&1
./tools/mfma_f64_peak > $O/peak.txt 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS -d $O/gnb -o p --output-format csv -- ./tools/gemm_nt_bench > $O/gnb.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA -d $O/peak -o p --output-format csv -- ./tools/mfma_f64_peak > $O/peak.log 2>&1
python3 tools/diag_clock_report.py $O > $O/report.txt 2>&1
cat $O/plain.txt $O/peak.txt $O/report.txt
