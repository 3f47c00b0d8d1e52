#!/bin/bash
# Round-2 profile set: kernel stats of the headline epoch, of config 5 (slab class, one rank) and of heat2d.
export TMPDIR=/tmp
R=$PWD
run() { # tag, command...
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- "$@" > $R/gpurun_out/prof_$tag.log 2>&1
  python3 profiles/summarize.py gpurun_out/prof_$tag "r02 $tag: $*" | head -40 > $R/gpurun_out/r02_${tag}_kernel_stats.txt
}
run v3_poisson python3 bench.py --no_cpu_baseline --steps 20 --warmup 5
run v3_cfg5_slab python3 bench.py --config 5 --no_cpu_baseline --steps 5 --warmup 2
run v3_heat2d python3 tools/heat2d_epochs.py 256 512 6
run v3_cfg5b_tracer4d python3 bench.py --config 5b --no_cpu_baseline --steps 5 --warmup 1
tail -2 gpurun_out/prof_v3_poisson.log | cut -c1-300
