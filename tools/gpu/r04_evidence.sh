#!/bin/bash
# GPU box: the round's evidence in one go -> gpurun_out/r04/ (copy to profiles/):
#   r04_kernel_stats_bench.csv   rocprofv3 --kernel-trace --stats of the default bench command (no extras)
#   r04_bench_under_rocprof.json the bench line of that run
#   r04_hbm_kernels.json, r04_pmc_tier1.json, r04_step_traffic.json   (tools/gpu/hbm_table.sh: every kernel alone, counters)
#   r04_bench_runs.txt           five consecutive default bench runs (ms/step, encode, decode, host figures)
#   r04_timeline_3slices.txt     kernel timeline summary of one traced step with the default three slices (idle time, gaps)
#   r04_bench_final.json         the full bench line with the extras
#   gpurun --timeout 1150 -- 'bash tools/gpu/r04_evidence.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04
rm -rf $O && mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/trace.log 2>&1
echo "trace rc=$?"
grep -v "^[WEI]2026" $O/trace.log | grep '^{' | tail -1 > $O/r04_bench_under_rocprof.json
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r04_kernel_stats_bench.csv
rm -rf $O/trace
bash tools/gpu/hbm_table.sh > $O/hbm_table.log 2>&1; echo "hbm rc=$?"
cp gpurun_out/hbm/r04_hbm_kernels.json gpurun_out/hbm/r04_pmc_tier1.json gpurun_out/hbm/r04_step_traffic.json $O/ 2>/dev/null
# the counter files name the kernel sources they were measured with: bench.py reads them from profiles/
mkdir -p profiles && cp $O/r04_hbm_kernels.json $O/r04_pmc_tier1.json $O/r04_step_traffic.json profiles/ 2>/dev/null
SLICES=3 bash tools/gpu/timeline.sh > $O/r04_timeline_3slices.txt 2>&1; echo "timeline rc=$?"
S='import json,sys; d=json.loads(sys.stdin.read()); h=d["host"]["ranks"][0]; print(d["ms_per_step"], d["value"], d["encode_GBps"], d["decode_GBps"], "zstd core-s", h["zstd_core_s_per_step"], "wait ms", h["zstd_wait_ms_per_step"], "throttled ms", (h.get("cgroup") or {}).get("throttled_ms_per_step"))'
for rep in 1 2 3 4 5; do
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python3 -c "$S"
done | tee $O/r04_bench_runs.txt
timeout -k 10 600 python3 bench.py --steps 10 --warmup 3 > $O/r04_bench_final.json 2> $O/bench_final.err
tail -c 400 $O/bench_final.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/r04_bench_final.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "encode_GBps", "decode_GBps")})
print("roofline", d["roofline"])
for k in ("config3", "residual_population", "shard4096", "h5_path", "host_api"):
    print(k, json.dumps(d.get(k))[:400])
print("parity", d.get("cpu_baseline", {}).get("stream_parity"))
PY
head -16 $O/r04_kernel_stats_bench.csv | cut -c1-170
