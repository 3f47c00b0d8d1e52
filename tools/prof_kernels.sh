#!/bin/bash
# ONE rocprofv3 pass (kernel trace + stats; no counters) of one bench.py configuration, summarised with EVERY kernel of the
# run listed (torch's at:: kernels included) into gpurun_out/<tag>_kernels.txt.   tools/prof_kernels.sh <tag> <config> [VAR=value ...]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
TAG=$1; CFG=$2; shift; shift
for kv in "$@"; do export "$kv"; done
OUT=$R/gpurun_out/prof_$TAG
CMD="python3 $R/bench.py --config $CFG --no_cpu_baseline --no_other_configs --steps ${STEPS:-2} --warmup ${WARMUP:-1}"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/K -- $CMD > $R/gpurun_out/${TAG}_bench.json 2> /dev/null )
python3 profiles/summarize.py $OUT "$TAG: bench.py --config $CFG $*" | cut -c1-160 > $R/gpurun_out/${TAG}_kernels.txt
head -40 $R/gpurun_out/${TAG}_kernels.txt
