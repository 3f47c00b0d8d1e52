#!/bin/bash
cd $GRAFT_REPO_ROOT
for x in 0 1 0 1; do
ODIL_DEBUG_FLAT_AXIS0=$x timeout 300 python bench.py --config 5 --no_cpu_baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg5 flat $x', d['ms_per_step'], d.get('kernel_ms'))"
done
