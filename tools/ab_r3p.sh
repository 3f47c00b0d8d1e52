#!/bin/bash
# A/B of the 'nccc' float transfers: mg tests, then config 5 (one rank) and 5b.
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_slab_gpu.py tests/test_properties_gpu.py -m gpu -q 2>&1 | tail -25
timeout 600 python -m pytest tests/test_workloads_gpu.py tests/test_fullsize_traced_gpu.py -m gpu -q -k "tracer or full" 2>&1 | tail -8
for mode in 0 1 1 0; do
ODIL_ADJ_ROWS=$mode timeout 300 python bench.py --config 5 --no_cpu_baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg5 rows $mode', d['ms_per_step'], d.get('kernel_ms'))"
done
for mode in 0 1 0 1; do
ODIL_ADJ_ROWS=$mode timeout 300 python bench.py --config 5b --no_cpu_baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg5b rows $mode', d['ms_per_step'], d.get('kernel_ms'))"
done
