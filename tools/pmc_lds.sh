#!/bin/bash
export TMPDIR=/tmp
R=$PWD
for c in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS; do
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_lds/$c -- python3 tools/adj_timing.py > /dev/null 2>&1
done
python3 - <<'PY'
import csv,glob
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(list))
for p in glob.glob('gpurun_out/pmc_lds/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(p)):
        n=r['Kernel_Name'].replace('void odil::','').split('(')[0]
        acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
for n in acc:
    if 'tile' in n or 'poisson_adjoint' in n:
        print(n[:40], {k:'%.4g'%max(v) for k,v in sorted(acc[n].items())})
PY
