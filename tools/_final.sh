cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python3 bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
tail -c 300 gpurun_out/bench_final.err
bash tools/prof_r02.sh > gpurun_out/prof_r02.log 2>&1
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
