#!/bin/bash
# Final validation of the round: the whole GPU suite (durations reported), the compiled JIT cache copied back so that the
# driver's run does not spend its time in hipcc, the default bench line.
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q --durations=25 > gpurun_out/r3_final_tests.log 2>&1
tail -40 gpurun_out/r3_final_tests.log
mkdir -p gpurun_out/jit_cache && cp odil_amd/_jit_cache/*.so gpurun_out/jit_cache/ 2>/dev/null; du -sh gpurun_out/jit_cache | tail -1
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench_full.json 2> gpurun_out/r3_bench_full.err; tail -2 gpurun_out/r3_bench_full.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r3_bench_full.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["frac_model"], d["kernel_ms"])
print(json.dumps(d.get("other_configs"), indent=1)[:2500])
print(d["cpu_baseline"])
PY
