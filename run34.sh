cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_r1f
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1f/trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_r1f/trace.log 2>&1
echo trace rc=$?
grep -v "^W2026\|^E2026\|^I2026" gpurun_out/prof_r1f/trace.log | tail -1 | cut -c1-1500
find gpurun_out/prof_r1f/trace -name "*kernel_stats.csv" | head -3
f=$(find gpurun_out/prof_r1f/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/prof_r1f/kernel_stats.csv && head -12 "$f" | cut -c1-200
# drop the bulky per-dispatch trace, keep the summary
find gpurun_out/prof_r1f/trace -name "*kernel_trace.csv" -delete
echo "---- PMC FETCH_SIZE"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_r1f/fetch -- python3 bench.py --steps 1 --warmup 1 --frames 64 --no-cpu-baseline > gpurun_out/prof_r1f/fetch.log 2>&1
echo fetch rc=$?
c=$(find gpurun_out/prof_r1f/fetch -name "*counter_collection.csv" | head -1); [ -n "$c" ] && (head -1 "$c"; grep -E "k_t1_encode|k_t1_decode|k_rate|k_j2k_rows" "$c" | head -12) | cut -c1-400 > gpurun_out/prof_r1f/fetch_rows.csv; cat gpurun_out/prof_r1f/fetch_rows.csv | head -8
find gpurun_out/prof_r1f/fetch -name "*.csv" -size +2M -delete
