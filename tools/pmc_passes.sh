#!/bin/bash
# Runs the rocprofv3 passes the profiles/ summaries come from over one python program (dev aid, GPU box):
#   tools/pmc_passes.sh <out dir under gpurun_out/> <program.py> [args]
# Passes (each its own run; --pmc is never combined with a trace domain):
#   trace   --kernel-trace --stats
#   fetch   FETCH_SIZE            write  WRITE_SIZE   (together they exceed the TCC slots)
#   sq1     GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
#   sq2     GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$out
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/trace -o run -- python3 $R/"$@" > $R/$out/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$out/fetch -o run -- python3 $R/"$@" > $R/$out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$out/write -o run -- python3 $R/"$@" > $R/$out/write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $R/$out/sq1 -o run -- python3 $R/"$@" > $R/$out/sq1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $R/$out/sq2 -o run -- python3 $R/"$@" > $R/$out/sq2.log 2>&1 || echo "sq2 pass failed"
cd $R
python3 tools/pmc_summary.py $out > $out/summary.txt
python3 tools/pmc_clock_summary.py $out/sq1 > $out/clock.txt
echo done $out
