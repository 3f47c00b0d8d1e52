cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ODIL_HIP_LIB=$PWD/odil_amd/libodil_hip_base.so timeout 600 python3 bench.py --config 5b --no_cpu_baseline --steps 5 --warmup 2 2>&1 | tail -1 | cut -c1-300
timeout 600 python3 bench.py --config 5b --no_cpu_baseline --steps 5 --warmup 2 2>&1 | tail -1 | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_5b -- python3 bench.py --config 5b --no_cpu_baseline --steps 5 --warmup 2 > gpurun_out/prof_5b.log 2>&1
python3 profiles/summarize.py gpurun_out/prof_5b "5b new" | head -24
