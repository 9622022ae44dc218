cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1c -- python3 bench.py --steps 1 --warmup 1 --frames 256 --no-cpu-baseline > gpurun_out/prof_r1c.log 2>&1
echo rc=$?
grep -v "^W2026\|^E2026\|^I2026" gpurun_out/prof_r1c.log | tail -3 | cut -c1-600
