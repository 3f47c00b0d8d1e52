#!/bin/bash
# Counter passes (separate runs, no tracing) for one workload:  tools/pmc_detail.sh <tag> <script.py> [args...]
# Pass A: SQ occupancy / issue mix; pass B: SQ waits + LDS; passes C, D: FETCH_SIZE, WRITE_SIZE (TCC slots do not fit together).
export TMPDIR=/tmp
R=$PWD
TAG=$1; shift
OUT=$R/gpurun_out/pmc_$TAG
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $OUT/A -- python3 "$@" > $OUT.A.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/B -- python3 "$@" > $OUT.B.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/C -- python3 "$@" > $OUT.C.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/D -- python3 "$@" > $OUT.D.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/K -- python3 "$@" > $OUT.K.log 2>&1
python3 profiles/summarize.py $OUT "$TAG: $*" > $R/gpurun_out/pmc_$TAG.txt
tail -3 $OUT.K.log
