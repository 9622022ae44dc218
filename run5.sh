cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1a -- python3 bench.py --steps 1 --warmup 0 --frames 256 --no-cpu-baseline > gpurun_out/prof_r1a.log 2>&1
ls -R gpurun_out/prof_r1a | head -20
f=$(find gpurun_out/prof_r1a -name "*kernel_stats.csv" | head -1); echo $f; head -40 $f
