#!/bin/bash
cd $GRAFT_REPO_ROOT
for mb in 1 0.5 2 1; do
ODIL_TRACE_CHUNK_MB=$mb timeout 300 python bench.py --config 5b --no_cpu_baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg5b chunk $mb', d['ms_per_step'], d.get('kernel_ms'))"
done
for mb in 1 8 0.5 8; do
ODIL_TRACE_CHUNK_MB=$mb timeout 300 python bench.py --config 5 --no_cpu_baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg5 chunk $mb', d['ms_per_step'], d.get('kernel_ms'))"
done
for mb in 0 4 1; do
ODIL_TRACE_CHUNK_MB=$mb timeout 300 python bench.py --config 3b --no_cpu_baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg3b chunk $mb', d['ms_per_step'], d.get('kernel_ms'))"
done
