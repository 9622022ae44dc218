#!/bin/bash
# GPU box: FETCH_SIZE / WRITE_SIZE calibration on known byte counts (tools/gpu/fetch_calib.hip) -> gpurun_out/calib/fetch_calib.json
#   gpurun --timeout 600 -- 'bash tools/gpu/fetch_calib.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/calib
rm -rf $O && mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $O/fetch_calib tools/gpu/fetch_calib.hip || exit 1
timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- $O/fetch_calib > $O/trace.log 2>&1; echo "trace rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$C -- $O/fetch_calib > $O/$C.log 2>&1; echo "$C rc=$?"
done
python3 - $O <<'PY'
import csv, glob, json, sys
O = sys.argv[1]
GIB = 1 << 30
def counter(name):
    acc = {}
    for p in glob.glob(f"{O}/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] == name:
                k = r["Kernel_Name"].split("(")[0]
                acc.setdefault(k, []).append(float(r["Counter_Value"]))
    return {k: v[-1] for k, v in acc.items()}                      # (second repetition)
dur = {}
for p in glob.glob(f"{O}/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        dur[r["Kernel_Name"].split("(")[0]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
f, w = counter("FETCH_SIZE"), counter("WRITE_SIZE")
rows = {}
for k in sorted(set(f) | set(w)):
    rows[k] = {"fetch_size_kb": f.get(k), "write_size_kb": w.get(k), "fetch_over_bytes": None if k not in f else round(f[k] * 1024 / GIB, 4),
               "write_over_bytes": None if k not in w else round(w[k] * 1024 / GIB, 4), "duration_us": dur.get(k),
               "GBps": None if k not in dur else round(GIB / dur[k] / 1e3, 1)}
json.dump({"bytes_moved_per_kernel": GIB, "kernels": rows}, open(f"{O}/fetch_calib.json", "w"), indent=1)
for k, r in rows.items(): print(k, r)
PY
rm -rf $O/trace $O/FETCH_SIZE $O/WRITE_SIZE $O/fetch_calib
