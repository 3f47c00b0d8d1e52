cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/pmc_detail.sh v3_poisson tools/adj_timing.py > gpurun_out/pmc_v3.log 2>&1
timeout 600 python3 bench.py --no_cpu_baseline 2>&1 | tail -1 > gpurun_out/bench4a.json
timeout 600 python3 bench.py --config 5b --no_cpu_baseline --steps 5 --warmup 2 2>&1 | tail -1 > gpurun_out/bench5b.json
timeout 600 python3 bench.py --config 5 --no_cpu_baseline --steps 5 --warmup 2 2>&1 | tail -1 > gpurun_out/bench5.json
