#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_heat2d_c -- python3 $R/bench.py --config 3b --no_cpu_baseline --steps 20 --warmup 3 > $R/gpurun_out/prof_heat2d_c.log 2>&1
cd $R
python3 profiles/summarize.py gpurun_out/prof_heat2d_c "r03 (final) heat2d: bench.py --config 3b --steps 20 --warmup 3" | head -24 > gpurun_out/r03_c_heat2d_kernel_stats.txt
cat gpurun_out/r03_c_heat2d_kernel_stats.txt
