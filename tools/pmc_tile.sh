#!/bin/bash
export TMPDIR=/tmp
R=$PWD
for c in FETCH_SIZE WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES; do
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_tile/$c -- python3 tools/adj_timing.py > /dev/null 2>&1
done
python3 - <<'PY'
import csv,glob
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(list))
for p in glob.glob('gpurun_out/pmc_tile/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(p)):
        n=r['Kernel_Name'].replace('void odil::','').split('(')[0]
        acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
for n in acc:
    if 'adj_tile' in n or 'poisson' in n:
        print(n[:40], {k:'%.4g'%max(v) for k,v in acc[n].items()})
PY
