#!/bin/bash
# Issue / wait counters of heat2d's forward kernel: the plain kernel and the tiled one (shared network evaluations).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for mode in 0 1; do
  OUT=$R/gpurun_out/pmc_heat2d_share$mode
  ( cd /tmp && ODIL_TRACE_SHARE=$mode timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $OUT/A -- python3 $R/bench.py --config 3b --no_cpu_baseline --steps 4 --warmup 2 > /dev/null 2>&1 )
  ( cd /tmp && ODIL_TRACE_SHARE=$mode timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/B -- python3 $R/bench.py --config 3b --no_cpu_baseline --steps 4 --warmup 2 > /dev/null 2>&1 )
  ( cd /tmp && ODIL_TRACE_SHARE=$mode timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/K -- python3 $R/bench.py --config 3b --no_cpu_baseline --steps 4 --warmup 2 > /dev/null 2>&1 )
  python3 profiles/summarize.py $OUT "r03 heat2d 256x512^2 f32, ODIL_TRACE_SHARE=$mode: bench.py --config 3b" | grep -v "^at::\|elementwise\|copyBuffer" | cut -c1-900 > $R/gpurun_out/r03_heat2d_share${mode}_pmc.txt
  grep -E "^k_fwd|^k_gat" $R/gpurun_out/r03_heat2d_share${mode}_pmc.txt | cut -c1-700
done
