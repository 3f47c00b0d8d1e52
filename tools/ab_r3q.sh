#!/bin/bash
cd $GRAFT_REPO_ROOT
for x in 0 1 0 1; do
ODIL_SLAB_STREAMS=$x timeout 300 python bench.py --config 5 --no_cpu_baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg5 streams $x', d['ms_per_step'], d.get('kernel_ms'))"
done
ODIL_SLAB_STREAMS=1 timeout 600 python -m pytest tests/test_slab_gpu.py -m gpu -q -k "traced or config5" 2>&1 | tail -3
