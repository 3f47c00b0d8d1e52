#!/bin/bash
# rocprofv3 kernel durations of the 512^3 epoch (tools/adj_timing.py) -> per-kernel average of the largest launches
export TMPDIR=/tmp
R=$PWD
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pe_$1 -- python3 tools/adj_timing.py > /dev/null 2>&1
python3 - "$1" <<'PY'
import csv,glob,sys
v=sys.argv[1]
p=glob.glob('gpurun_out/pe_%s/*/*kernel_trace.csv'%v)[0]
d={}
for r in csv.DictReader(open(p)):
    n=r['Kernel_Name'].replace('void odil::','').split('(')[0]
    d.setdefault(n,[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for n in sorted(d,key=lambda k:-sum(d[k])):
    xs=sorted(d[n],reverse=True)
    big=[x for x in xs if x>0.5*xs[0]]
    if sum(xs)>200: print(v,n[:44],'n=%d largest avg %.1f us'%(len(xs),sum(big)/len(big)), 'total/epoch %.1f'%(sum(xs)/25))
PY
