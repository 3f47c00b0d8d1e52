#!/bin/bash
# usage: tools/prof_variants.sh <variant letters...>: rocprofv3 kernel stats of the 512^3 epoch per library variant
export TMPDIR=/tmp
R=$PWD
for v in "$@"; do
  export ODIL_HIP_LIB=$R/odil_amd/libodil_hip_$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pv_$v -- python3 tools/adj_timing.py > /dev/null 2>&1
  python3 - "$v" <<'PY'
import csv,glob,sys
v=sys.argv[1]
p=glob.glob('gpurun_out/pv_%s/*/*kernel_trace.csv'%v)[0]
d={}
for r in csv.DictReader(open(p)):
    n=r['Kernel_Name'].replace('void odil::','').split('(')[0]
    d.setdefault(n,[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for n in d:
    if 'adj' in n or 'synth' in n:
        xs=sorted(d[n],reverse=True)
        big=[x for x in xs if x>0.5*xs[0]]
        print(v,n[:40],'n=%d max-level avg %.1f us'%(len(xs),sum(big)/len(big)), 'total/epoch %.1f'%(sum(xs)/25))
PY
done
