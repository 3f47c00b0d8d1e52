#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 200 python tools/debug_heat_newton.py 2>&1 | grep -v amdgpu | tail -12
timeout 600 python -m pytest tests/test_workloads_gpu.py tests/test_slab_gpu.py -m gpu -x -q -k "tracer or slab or random or windows" 2>&1 | tail -4
run() { name=$1; shift
  timeout 300 env "$@" python bench.py --no_cpu_baseline --steps 8 --warmup 3 ${CFG} > gpurun_out/r3_ab_${name}.json 2>gpurun_out/r3_ab_${name}.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/r3_ab_${name}.json") if l.startswith("{")][-1])
    print("${name}", "ms_per_step", round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d.get("kernel_ms", {}).items()})
except Exception as e:
    print("${name}", "FAILED", e)
PY
}
CFG="--config 5"; run cfg5_tinner_ch4 X=1
CFG="--config 5"; run cfg5_flat ODIL_TRACE_TINNER=0
CFG="--config 5"; run cfg5_tinner_ch1 ODIL_TRACE_TCHUNK=1
CFG="--config 5"; run cfg5_tinner_ch16 ODIL_TRACE_TCHUNK=16
CFG="--config 5b"; run 5b_tinner X=1
CFG="--config 5b"; run 5b_flat ODIL_TRACE_TINNER=0
CFG="--config 3b"; run 3b_tinner X=1
CFG="--config 3b"; run 3b_flat ODIL_TRACE_TINNER=0
