#!/bin/bash
export TMPDIR=/tmp
R=$PWD
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_epoch/$c -- python3 tools/adj_timing.py > /dev/null 2>&1
done
python3 - <<'PY'
import csv,glob
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(list))
for p in glob.glob('gpurun_out/pmc_epoch/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(p)):
        n=r['Kernel_Name'].replace('void odil::','').split('(')[0]
        acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
for n in acc:
    if 'tile' in n or 'poisson' in n:
        f=max(acc[n].get('FETCH_SIZE',[0]))*2*1024/1e9; w=max(acc[n].get('WRITE_SIZE',[0]))*1024/1e9
        print('%-40s fetch %.3f GB (x2 corrected)  write %.3f GB  total %.3f GB'%(n[:40],f,w,f+w))
PY
