#!/bin/bash
# GPU box: rocprofv3 kernel statistics of the default bench run + FETCH_SIZE / WRITE_SIZE passes (separate runs,
# counters never combined with API tracing).  Results under gpurun_out/prof/; copy the summaries to profiles/.
#   gpurun --timeout 1100 -- 'bash tools/gpu/profile.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof
rm -rf $O && mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/trace.log 2>&1
echo "trace rc=$?"
grep -v "^[WEI]2026" $O/trace.log | tail -1 > $O/bench_under_rocprof.json
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv
find $O/trace -name "*kernel_trace.csv" -delete
for C in FETCH_SIZE WRITE_SIZE; do
  EBCC_HIP_SLICES=1 timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$C -- python3 bench.py --steps 1 --warmup 1 --frames 64 --no-cpu-baseline --no-extras > $O/$C.log 2>&1
  echo "$C rc=$?"
  c=$(find $O/$C -name "*counter_collection.csv" | head -1)
  [ -n "$c" ] && python3 tools/pmc_summary.py "$c" $C 64 > $O/pmc_$C.json
  rm -rf $O/$C
done
python3 bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-2500
cat $O/pmc_FETCH_SIZE.json $O/pmc_WRITE_SIZE.json | head -40
head -14 $O/kernel_stats.csv | cut -c1-160
