#!/bin/bash
# register / scratch use of every kernel of one source file: tools/kres.sh odil_amd/csrc/smooth2.hip
cd $(dirname $1) && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I../../include -c $(basename $1) -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re,subprocess
name=None; rec={}
for line in sys.stdin:
    if 'error' in line: print(line.rstrip())
    m=re.search(r'Function Name: (\S+)', line)
    if m:
        name=m.group(1); rec[name]={}
    for key in ('VGPRs','VGPRs Spill','ScratchSize [bytes/lane]','LDS Size [bytes/block]','SGPRs Spill','Occupancy [waves/SIMD]'):
        m=re.search(r'remark:\s+'+re.escape(key)+r': (\d+)', line)
        if m and name: rec[name][key]=m.group(1)
for n,r in rec.items(): print(n[-60:], ' '.join('%s=%s'%(k.split()[0]+('S' if 'Spill' in k else ''),v) for k,v in r.items()))
"
