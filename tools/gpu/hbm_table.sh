#!/bin/bash
# GPU box: every bandwidth-bound kernel measured alone -> gpurun_out/hbm/r04_hbm_kernels.json (copy to profiles/), and the
# tier-1 encoder's counter traffic -> gpurun_out/hbm/r04_pmc_tier1.json.   gpurun --timeout 1100 -- 'bash tools/gpu/hbm_table.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/hbm
F=${1:-256}
rm -rf $O && mkdir -p $O
export EBCC_HIP_SLICES=1 EBCC_HIP_DECODE_SLICES=1
CMD="python3 bench.py --steps 1 --warmup 1 --frames $F --no-cpu-baseline --no-extras"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- $CMD > $O/trace.log 2>&1; echo "trace rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$C -- $CMD > $O/$C.log 2>&1; echo "$C rc=$?"
done
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
A=$(find $O/FETCH_SIZE -name "*counter_collection.csv" | head -1)
B=$(find $O/WRITE_SIZE -name "*counter_collection.csv" | head -1)
SHA=$(cat ebcc_amd/csrc/j2k_analysis.hip ebcc_amd/csrc/t1_core.hpp ebcc_amd/csrc/j2k_rate.hip ebcc_amd/csrc/residual_dwt.hip ebcc_amd/csrc/residual_spiht.hip | sha256sum | cut -c1-16)
head -1 $T > $O/trace_header.txt; head -1 $A > $O/counter_header.txt
python3 tools/hbm_table.py $T $A $B $F $SHA > $O/r04_hbm_kernels.json
python3 - $A $B $F $SHA > $O/r04_pmc_tier1.json <<'PY'
import csv, json, re, sys
def per_kernel(path, counter):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter: continue
        m = re.search(r"(k_t1_(scan|rowoffs|emit|mqrows))", r["Kernel_Name"])
        if not m: continue
        a = acc.setdefault(m.group(1), [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
    return {k: v[1] / v[0] for k, v in acc.items()}
f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
ks = sorted(f)
frames = int(sys.argv[3])
per_frame = sum(2 * f[k] + w.get(k, 0) for k in ks) * 1024 / frames
print(json.dumps({"kernels": ks, "fetch_kb": f, "write_kb": w, "frames_per_dispatch": frames, "kernel_sources_sha": sys.argv[4],
                  "bytes_per_frame_fetch_x2_plus_write": int(per_frame),
                  "unit": "KB as reported by rocprofv3; FETCH_SIZE x 2 on gfx950 (micro-architecture guide)"}, indent=1))
PY
python3 - $A $B $F $SHA > $O/r04_step_traffic.json <<'PY'
import csv, json, re, sys
def total(path, counter):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter: continue
        m = re.search(r"(k_[A-Za-z0-9_]+)", r["Kernel_Name"])
        if not m: continue
        acc[m.group(1)] = acc.get(m.group(1), 0.0) + float(r["Counter_Value"])
    return acc
f, w = total(sys.argv[1], "FETCH_SIZE"), total(sys.argv[2], "WRITE_SIZE")
frames = int(sys.argv[3])
round_trips = 2 * frames                                      # warm-up + 1 step of the command above
per = {k: (2 * f.get(k, 0) + w.get(k, 0)) * 1024 / round_trips for k in sorted(set(f) | set(w))}
print(json.dumps({"what": "HBM bytes per frame round trip (encode + decode) over all k_* kernels: FETCH_SIZE x 2 + WRITE_SIZE (KB units), warm-up + 1 step",
                  "frames_per_dispatch": frames, "kernel_sources_sha": sys.argv[4], "bytes_per_frame_round_trip": int(sum(per.values())),
                  "algorithmic_bytes_per_frame_round_trip": 2 * 4152960 + 2 * 94479,
                  "by_kernel": {k: int(v) for k, v in sorted(per.items(), key=lambda kv: -kv[1])}}, indent=1))
PY
rm -rf $O/trace $O/FETCH_SIZE $O/WRITE_SIZE
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/hbm/r04_hbm_kernels.json"))
print("unmatched", d["unmatched_dispatches"])
for r in d["kernels"][:40]:
    print(f'{r["kernel"][:34]:34s} {r["grid_threads"]:9d} {r["duration_us"]:9.1f} us  {r["achieved_GBps"]:8.1f} GB/s  {100*(r["frac_of_6290"] or 0):5.1f}% of 6.29  x{r["counter_over_algorithmic"]:5.2f} counter  ({r["dispatches_all_frames_active"]}/{r["dispatches"]}) {"cache-served" if r.get("cache_served") else ""}')
print(open("gpurun_out/hbm/r04_pmc_tier1.json").read()[:700])
print(open("gpurun_out/hbm/r04_step_traffic.json").read()[:1500])
PY
