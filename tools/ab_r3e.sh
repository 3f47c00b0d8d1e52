#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_workloads_gpu.py -m gpu -x -q -k "implicit" -s 2>&1 | grep -v amdgpu | tail -15
timeout 600 python -m pytest tests/test_workloads_gpu.py -m gpu -x -q -k "heat or tiled" 2>&1 | tail -4
run() { name=$1; shift
  timeout 300 env "$@" python bench.py --no_cpu_baseline --steps 10 --warmup 3 ${CFG} > gpurun_out/r3_ab_${name}.json 2>gpurun_out/r3_ab_${name}.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/r3_ab_${name}.json") if l.startswith("{")][-1])
    print("${name}", "ms_per_step", round(d["ms_per_step"], 3))
except Exception as e:
    print("${name}", "FAILED", e)
PY
}
CFG="--config 3b"; run 3b_share_7x32 ODIL_TRACE_SHARE=1
CFG="--config 3b"; run 3b_noshare ODIL_TRACE_SHARE=0
CFG="--config 3b"; run 3b_share_3x64 ODIL_TRACE_SHARE=1 ODIL_TRACE_TILE=3x64
CFG="--config 3"; run 3_share ODIL_TRACE_SHARE=1
CFG="--config 3"; run 3_noshare ODIL_TRACE_SHARE=0
