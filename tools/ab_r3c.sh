#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_slab_gpu.py tests/test_workloads_gpu.py tests/test_api_gpu.py -m gpu -x -q > gpurun_out/r3_t4.log 2>&1; tail -3 gpurun_out/r3_t4.log
run() { name=$1; shift
  env "$@" python bench.py --no_cpu_baseline --steps 5 --warmup 2 ${CFG} > gpurun_out/r3_ab_${name}.json 2>gpurun_out/r3_ab_${name}.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/r3_ab_${name}.json") if l.startswith("{")][-1])
    print("${name}", "ms_per_step", round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d.get("kernel_ms", {}).items()})
except Exception as e:
    print("${name}", "FAILED", e)
PY
}
CFG="--config 5"; run cfg5_merged X=1
CFG="--config 5"; run cfg5_separate ODIL_TRACE_MERGE=0
CFG="--config 5"; run cfg5_merged_w3 ODIL_TRACE_WAVES_GAT=3
CFG="--config 5"; run cfg5_merged_w4 ODIL_TRACE_WAVES_GAT=4
CFG="--config 5"; run cfg5_fwd_w4 ODIL_TRACE_WAVES_FWD=4
CFG="--config 5"; run cfg5_fwd_w2 ODIL_TRACE_WAVES_FWD=2
CFG="--config 5b"; run 5b_merged X=1
CFG="--config 5b"; run 5b_separate ODIL_TRACE_MERGE=0
CFG="--config 5 --steps 10"; run cfg5_merged_again X=1
