#!/bin/bash
# HIP API + kernel + memory-copy trace of a few config-5 epochs: which API calls issue the small copy kernels (answer: the
# float() conversions of the loss terms after the timed steps, none inside an epoch).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && timeout 600 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/hiptrace -- python3 $R/bench.py --config 5 --no_cpu_baseline --steps 3 --warmup 2 > $R/gpurun_out/hiptrace.log 2>&1
cd $R
ls gpurun_out/hiptrace/*/ | head
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/hiptrace/*/*hip_api_trace.csv')
print(f)
rows = list(csv.DictReader(open(f[0])))
print(rows[0].keys())
c = collections.Counter(r['Function'] for r in rows)
for k, v in c.most_common(25): print(v, k)
m = glob.glob('gpurun_out/hiptrace/*/*memory_copy_trace.csv')
if m:
    mr = list(csv.DictReader(open(m[0])))
    print(len(mr), mr[0].keys())
    cc = collections.Counter((r.get('Direction'), ) for r in mr)
    print(cc)
PY
