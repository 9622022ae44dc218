set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python bench.py --steps 1 --warmup 1 --frames 64 2>&1 | tail -3
python bench.py --steps 2 --warmup 1 > gpurun_out/bench_r1_first.json 2> gpurun_out/bench_r1_first.err; tail -2 gpurun_out/bench_r1_first.err; cat gpurun_out/bench_r1_first.json
