#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_slab_gpu.py tests/test_workloads_gpu.py -m gpu -x -q > gpurun_out/r3_t3.log 2>&1; tail -3 gpurun_out/r3_t3.log
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
CFG="--config 5"; run cfg5_default X=1
CFG="--config 3b"; run 3b_default X=1
CFG="--config 3b"; run 3b_gat_scalar ODIL_TRACE_VEC=0
CFG="--config 3b"; run 3b_legacy ODIL_TRACE_RECOMPUTE=0 ODIL_TRACE_NEWGATHER=0 ODIL_TRACE_VEC=0
CFG="--config 5b"; run 5b_default X=1
python tools/slab_traced_emulated.py 2 32 128 2>&1 | grep -v amdgpu.ids
python tools/slab_traced_emulated.py 2 128 32 2>&1 | grep -v amdgpu.ids | head -3
