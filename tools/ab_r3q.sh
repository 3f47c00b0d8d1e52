#!/bin/bash
cd $GRAFT_REPO_ROOT
for mb in 5 2.5 1.1 0.6 5 2.5; do
ODIL_TRACE_CHUNK_MB=$mb timeout 300 python bench.py --config 5 --no_cpu_baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg5 chunk $mb', d['ms_per_step'], d['kernel_ms']['forward'], d['kernel_ms']['gather'])"
done
