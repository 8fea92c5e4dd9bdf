#!/bin/bash
# SQ / GRBM counters per kernel of ANY target script, three rocprofv3 --pmc passes (counters in their own runs,
# --kernel-trace only beside them), summarised per kernel family by tools/sq_parse.py:
#   bash tools/sq_counters.sh <tag> <family-regex> -- python3 tools/<target>.py [args]
# -> gpurun_out/sq_<tag>.txt.  The program after -- must be the python interpreter itself (no env / bash hop:
# the profiler's preloaded library initialises the GPU before the program starts).
set -e
TAG=$1; FAM=$2; shift 2; [ "$1" == "--" ] && shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
  --kernel-trace -d $OUT/sq1 -o sq1 --output-format csv -- "$@" > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES \
  --kernel-trace -d $OUT/sq2 -o sq2 --output-format csv -- "$@" > $OUT/sq2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM \
  --kernel-trace -d $OUT/sq3 -o sq3 --output-format csv -- "$@" > $OUT/sq3.log 2>&1 || true
python3 $R/tools/sq_parse.py "$FAM" $OUT "$*" > $R/gpurun_out/sq_$TAG.txt
cat $R/gpurun_out/sq_$TAG.txt
