#!/bin/bash
# Final validation of the round: the whole GPU suite (durations reported), the compiled JIT cache copied back so that the
# driver's run does not spend its time in hipcc, the default bench line.
cd $GRAFT_REPO_ROOT
touch /tmp/odil_run_start
timeout 2400 python -m pytest tests -m gpu -q --durations=25 > gpurun_out/r3_final_tests.log 2>&1
tail -40 gpurun_out/r3_final_tests.log
# only the generated kernels this run loaded or compiled (every load touches its file): a cache pruned of older builds
rm -rf gpurun_out/jit_used; mkdir -p gpurun_out/jit_used
timeout 300 python bench.py --config 5b --no_cpu_baseline --steps 3 --warmup 1 > /dev/null 2>&1
timeout 300 python bench.py --config 5 --no_cpu_baseline --steps 3 --warmup 1 > /dev/null 2>&1
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > /dev/null 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench_full.json 2> gpurun_out/r3_bench_full.err; tail -2 gpurun_out/r3_bench_full.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r3_bench_full.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["frac_model"], d["kernel_ms"])
print(json.dumps(d.get("other_configs"), indent=1)[:2500])
print(d["cpu_baseline"])
PY
find odil_amd/_jit_cache -name '*.so' -newer /tmp/odil_run_start -exec cp {} gpurun_out/jit_used/ \;
ls gpurun_out/jit_used | wc -l; du -sh gpurun_out/jit_used | tail -1
