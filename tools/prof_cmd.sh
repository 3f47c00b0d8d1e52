#!/bin/bash
# rocprofv3 passes of ANY python command of this repo (run through gpurun): kernel trace + two SQ counter passes + the
# L2 <-> fabric request counters, each in a run of its own, summarised into gpurun_out/<tag>_pmc.txt.
#   tools/prof_cmd.sh <tag> <script.py> [args ...]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG
CMD="python3 $R/$*"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/K -- $CMD > $R/gpurun_out/${TAG}_stdout.txt 2> /dev/null )
( cd /tmp && timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $OUT/A -- $CMD > /dev/null 2>&1 )
( cd /tmp && timeout 400 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/B -- $CMD > /dev/null 2>&1 )
( cd /tmp && timeout 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $OUT/C -- $CMD > /dev/null 2>&1 )
python3 profiles/summarize.py $OUT "$TAG: $*" | cut -c1-900 > $R/gpurun_out/${TAG}_pmc.txt
grep -v "^at::" $R/gpurun_out/${TAG}_pmc.txt | head -60
