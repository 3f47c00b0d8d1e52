#!/bin/bash
# round-3 A/B of the traced path: tests of the slab kernels, then config 5 (one rank) and 3b under variants
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_slab_gpu.py -m gpu -x -q > gpurun_out/r3_t2.log 2>&1; tail -3 gpurun_out/r3_t2.log
run() { # name, env..., -- args
  name=$1; shift
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
CFG="--config 5"; run cfg5_legacy ODIL_TRACE_RECOMPUTE=0 ODIL_TRACE_NEWGATHER=0 ODIL_TRACE_VEC=0
CFG="--config 5"; run cfg5_novec ODIL_TRACE_VEC=0
CFG="--config 3b"; run 3b_default X=1
CFG="--config 3b"; run 3b_novec ODIL_TRACE_VEC=0
CFG="--config 3b"; run 3b_legacy ODIL_TRACE_RECOMPUTE=0 ODIL_TRACE_NEWGATHER=0 ODIL_TRACE_VEC=0
CFG="--config 3b"; run 3b_vec_oldgather ODIL_TRACE_NEWGATHER=0
CFG="--config 5b"; run 5b_novec ODIL_TRACE_VEC=0
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_cfg5_r3a -- python3 $GRAFT_REPO_ROOT/bench.py --config 5 --steps 5 --warmup 2 --no_cpu_baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_3b_r3a -- python3 $GRAFT_REPO_ROOT/bench.py --config 3b --steps 5 --warmup 2 --no_cpu_baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python profiles/summarize_db.py gpurun_out/prof_cfg5_r3a cfg5 | head -30
python profiles/summarize_db.py gpurun_out/prof_3b_r3a 3b | head -16
