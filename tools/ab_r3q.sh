#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_workloads_gpu.py tests/test_fullsize_traced_gpu.py tests/test_slab_gpu.py tests/test_api_gpu.py tests/test_trajectories.py -m gpu -q 2>&1 | tail -4
for k in 1 2; do
timeout 300 python bench.py --config 5 --no_cpu_baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg5', d['ms_per_step'], d['kernel_ms'])"
done
for k in 0 1; do
ODIL_TRACE_NT=$k timeout 300 python bench.py --config 5b --no_cpu_baseline --steps 20 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg5b nt $k', d['ms_per_step'])"
ODIL_TRACE_NT=$k timeout 300 python bench.py --config 3b --no_cpu_baseline --steps 20 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg3b nt $k', d['ms_per_step'])"
done
mkdir -p gpurun_out/jit_cache && cp odil_amd/_jit_cache/*.so gpurun_out/jit_cache/ 2>/dev/null
