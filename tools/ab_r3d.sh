#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_workloads_gpu.py -m gpu -x -q -k "heat or tiled or implicit" > gpurun_out/r3_t6.log 2>&1; tail -4 gpurun_out/r3_t6.log
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
CFG="--config 3b"; run 3b_share_8x32 X=1
CFG="--config 3b"; run 3b_noshare ODIL_TRACE_SHARE=0
CFG="--config 3b"; run 3b_share_4x32 ODIL_TRACE_TILE=4x32
CFG="--config 3b"; run 3b_share_16x16 ODIL_TRACE_TILE=16x16
CFG="--config 3b"; run 3b_share_8x16 ODIL_TRACE_TILE=8x16
CFG="--config 3b"; run 3b_share_4x64 ODIL_TRACE_TILE=4x64
CFG="--config 3b"; run 3b_share_8x32_w3 ODIL_TRACE_WAVES_FWD=3
CFG="--config 3"; run 3_share X=1
CFG="--config 3"; run 3_noshare ODIL_TRACE_SHARE=0
