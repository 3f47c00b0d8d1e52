#!/bin/bash
# Late round-3 profiles (after the transfer / traversal work): kernel stats of config 5 (one slab rank) and 5b, HBM counters of config 5.
# (separate --pmc passes, nothing else enabled) for the headline epoch and for config 5's generated kernels.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
run() { # tag, command...
  tag=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- "$@" > $R/gpurun_out/prof_$tag.log 2>&1 )
  python3 profiles/summarize.py gpurun_out/prof_$tag "r03 (late) $tag: $*" | head -40 > $R/gpurun_out/r03_c_${tag}_kernel_stats.txt
  head -12 $R/gpurun_out/r03_c_${tag}_kernel_stats.txt
}
pmc() { # tag, command...
  tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 600 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$tag/$c -- "$@" > /dev/null 2>&1 )
  done
  python3 - <<PY > $R/gpurun_out/r03_c_${tag}_pmc.txt
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for p in glob.glob('gpurun_out/pmc_${tag}/*/*/*counter_collection.csv') + glob.glob('gpurun_out/pmc_${tag}/*/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(p)):
        n = r['Kernel_Name'].replace('void odil::', '').split('(')[0]
        acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
print("# r03 ${tag}: HBM traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes; counters in KB;")
print("# FETCH_SIZE doubled: gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md); per-dispatch maximum; command: $*")
for n in sorted(acc, key=lambda k: -max(acc[k].get('FETCH_SIZE', [0]))):
    f = max(acc[n].get('FETCH_SIZE', [0])) * 2 * 1024 / 1e9
    w = max(acc[n].get('WRITE_SIZE', [0])) * 1024 / 1e9
    if f + w > 0.05:
        print('%-44s fetch %7.3f GB  write %7.3f GB  total %7.3f GB  (n=%d)' % (n[:44], f, w, f + w, len(acc[n].get('FETCH_SIZE', []))))
PY
  cat $R/gpurun_out/r03_c_${tag}_pmc.txt | head -14
}
run cfg5_slab python3 $R/bench.py --config 5 --no_cpu_baseline --steps 5 --warmup 2
run cfg5b_tracer4d python3 $R/bench.py --config 5b --no_cpu_baseline --steps 5 --warmup 2
pmc cfg5_slab python3 $R/bench.py --config 5 --no_cpu_baseline --steps 3 --warmup 1
