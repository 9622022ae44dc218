cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== plain, MALLOC_CHECK_=3"
MALLOC_CHECK_=3 MALLOC_PERTURB_=165 timeout -k 10 300 python3 bench.py --steps 1 --warmup 0 --frames 256 --no-cpu-baseline 2>&1 | tail -3 | cut -c1-400
echo "== rocprof 64 frames"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1b -- python3 bench.py --steps 1 --warmup 0 --frames 64 --no-cpu-baseline > gpurun_out/prof_r1b.log 2>&1
echo rc=$?
grep -v "^W2026\|^E2026\|^I2026" gpurun_out/prof_r1b.log | tail -5 | cut -c1-300
