#!/bin/bash
# SQ counter passes A and B of tools/pmc_detail.sh only:  tools/pmc_ab.sh <tag> <script.py> [args...]
export TMPDIR=/tmp
R=$PWD
TAG=$1; shift
OUT=$R/gpurun_out/pmc_$TAG
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $OUT/A -- python3 "$@" > $OUT.A.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/B -- python3 "$@" > $OUT.B.log 2>&1
python3 profiles/summarize.py $OUT "$TAG: $*" > $R/gpurun_out/pmc_$TAG.txt
