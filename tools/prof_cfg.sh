#!/bin/bash
# rocprofv3 passes of one bench.py configuration (run through gpurun): kernel trace + two SQ counter passes (separate runs,
# as the guide prescribes), summarised into gpurun_out/<tag>_pmc.txt.   tools/prof_cfg.sh <tag> <config> [VAR=value ...]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
TAG=$1; CFG=$2; shift; shift
for kv in "$@"; do export "$kv"; done
OUT=$R/gpurun_out/prof_$TAG
CMD="python3 $R/bench.py --config $CFG --no_cpu_baseline --no_other_configs --steps 4 --warmup 2"
# (the kernel-trace pass keeps bench.py's own line: its live HIP-event figure for the dominant launch and the profiler's
# average come from the SAME process)
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/K -- $CMD > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> /dev/null )
( cd /tmp && timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $OUT/A -- $CMD > /dev/null 2>&1 )
( cd /tmp && timeout 400 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/B -- $CMD > /dev/null 2>&1 )
if [ -n "$HBM" ]; then
( cd /tmp && timeout 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $OUT/C -- $CMD > /dev/null 2>&1 )
( cd /tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $OUT/D -- $CMD > /dev/null 2>&1 )
fi
# (every kernel of the run is listed, torch's `at::` elementwise / copy kernels included: they are part of what ran)
python3 profiles/summarize.py $OUT "$TAG: bench.py --config $CFG $*" | cut -c1-900 > $R/gpurun_out/${TAG}_pmc.txt
grep -E "^k_fwd|^k_gat|^kernel" $R/gpurun_out/${TAG}_pmc.txt | cut -c1-600
