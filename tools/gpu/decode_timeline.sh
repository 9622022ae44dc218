#!/bin/bash
# GPU box: what runs when in the decode of one batch (one slice): start offset and duration of every kernel from the SPIHT
# decoder's launch to the end of the batch.   gpurun --timeout 600 -- 'bash tools/gpu/decode_timeline.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/dectl
rm -rf $O && mkdir -p $O
export EBCC_HIP_SLICES=1 EBCC_HIP_DECODE_SLICES=1
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/trace.log 2>&1
echo "trace rc=$?"
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, re, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"^void ", "", n).split("(")[0].replace("ebcc::", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "?")))
rows.sort()
last = max(i for i, r in enumerate(rows) if r[2].startswith("k_spiht_decode"))
t0 = rows[last][0]
sel = [r for r in rows if r[0] >= t0 - 3_000_000]
print(f"{'start ms':>9s} {'dur ms':>8s}  queue kernel")
for s, e, n, q in sel:
    if e - s > 20_000: print(f"{(s - t0) / 1e6:9.3f} {(e - s) / 1e6:8.3f}  {q:>5s} {n}")
print("end of batch at", (max(r[1] for r in sel) - t0) / 1e6, "ms")
PY
rm -rf $O/trace
