#!/bin/bash
# A/B of the chunked traversal of the generated kernels (ODIL_TRACE_CHUNK_MB): parity with tiny chunks, then config 5 / 5b.
cd $GRAFT_REPO_ROOT
ODIL_TRACE_CHUNK_MB=0.002 timeout 900 python -m pytest tests/test_workloads_gpu.py tests/test_slab_gpu.py -m gpu -q -x -k "tracer or traced or slab" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_fullsize_traced_gpu.py -m gpu -q 2>&1 | tail -3
for mb in 0 4 2 8 0 4; do
ODIL_TRACE_CHUNK_MB=$mb timeout 300 python bench.py --config 5 --no_cpu_baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg5 chunk $mb', d['ms_per_step'], d.get('kernel_ms'))"
done
for mb in 0 4 2 8 16 0 4; do
ODIL_TRACE_CHUNK_MB=$mb timeout 300 python bench.py --config 5b --no_cpu_baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg5b chunk $mb', d['ms_per_step'], d.get('kernel_ms'))"
done
mkdir -p gpurun_out/jit_cache && cp odil_amd/_jit_cache/*.so gpurun_out/jit_cache/ 2>/dev/null
