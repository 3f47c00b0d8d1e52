#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_workloads_gpu.py tests/test_fullsize_traced_gpu.py tests/test_slab_gpu.py -m gpu -q 2>&1 | tail -4
ODIL_TRACE_XCD_GROUP=2 timeout 900 python -m pytest tests/test_workloads_gpu.py -m gpu -q -k "random or windows or equals_generic" 2>&1 | tail -3
for x in 1 2; do
timeout 300 python bench.py --config 5 --no_cpu_baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg5', d['ms_per_step'], d['kernel_ms'])"
done
timeout 300 python bench.py --config 5b --no_cpu_baseline --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg5b', d['ms_per_step'], d['steps'])"
mkdir -p gpurun_out/jit_cache && cp odil_amd/_jit_cache/*.so gpurun_out/jit_cache/ 2>/dev/null
