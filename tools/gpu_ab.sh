#!/bin/bash
# GPU-side A/B harness (run through gpurun): traced-operator parity tests, then the traced configs with a switch of the
# code generator on / off.  Usage: tools/gpu_ab.sh <ENV_VAR> [configs...]   (the variable is run with 1 and 0)
cd $GRAFT_REPO_ROOT
VAR=${1:-ODIL_TRACE_SHARE}; shift
CFGS=${@:-3b 5 5b}
touch /tmp/odil_run_start
if [ -z "$SKIP_TESTS" ]; then
  timeout 1500 python -m pytest ${TESTS:-tests/test_workloads_gpu.py tests/test_fullsize_traced_gpu.py tests/test_slab_gpu.py} -m gpu -x -q 2>&1 | tail -${TAIL:-5}
fi
for cfg in $CFGS; do
  for val in ${VALS:-1 0 1 0}; do
    env $VAR=$val timeout 400 python bench.py --config $cfg --no_cpu_baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg $cfg $VAR=$val', round(d['ms_per_step'],3), d.get('kernel_ms'))"
  done
done
rm -rf gpurun_out/jit_used; mkdir -p gpurun_out/jit_used
find odil_amd/_jit_cache -name '*.so' -newer /tmp/odil_run_start -exec cp {} gpurun_out/jit_used/ \;
ls gpurun_out/jit_used | wc -l
