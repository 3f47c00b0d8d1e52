#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_kernels.py tests/test_properties_gpu.py tests/test_fullsize_gpu.py -m gpu -q 2>&1 | tail -3
for k in 1 2 3; do
timeout 300 python bench.py --no_cpu_baseline --no_other_configs --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('poisson', d['ms_per_step'], d['kernel_ms'])"
done
timeout 300 python bench.py --config 3b --no_cpu_baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg3b', d['ms_per_step'])"
