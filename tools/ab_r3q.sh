#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_slab_gpu.py tests/test_properties_gpu.py -m gpu -q 2>&1 | tail -4
for x in 1 2 3; do
timeout 300 python bench.py --config 5 --no_cpu_baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg5', d['ms_per_step'], d.get('kernel_ms'))"
done
timeout 300 python bench.py --config 5b --no_cpu_baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg5b', d['ms_per_step'])"
timeout 300 python bench.py --config 5b --no_cpu_baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg5b', d['ms_per_step'])"
