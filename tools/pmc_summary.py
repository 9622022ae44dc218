#!/usr/bin/env python3
"""Per-kernel totals of one rocprofv3 --pmc counter (FETCH_SIZE / WRITE_SIZE, reported in KiB-like units of 1 KB)
from a counter_collection.csv:  {kernel: {"dispatches": n, "sum": total, "per_dispatch": mean}}.
The run profiled is `bench.py --frames 64` with EBCC_HIP_SLICES=1: every encode kernel is dispatched once per step
for all 64 frames (argv[3] = frames per dispatch, recorded for bench.py)."""
import csv
import json
import re
import sys

path, counter = sys.argv[1], sys.argv[2]
acc = {}
for r in csv.DictReader(open(path)):
    if r["Counter_Name"] != counter:
        continue
    m = re.search(r"(k_[A-Za-z0-9_]+)", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"][:40]
    a = acc.setdefault(name, [0, 0.0])
    a[0] += 1
    a[1] += float(r["Counter_Value"])
out = {k: {"dispatches": v[0], "sum": round(v[1], 1), "per_dispatch": round(v[1] / v[0], 2)} for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])}
frames = int(sys.argv[3]) if len(sys.argv) > 3 else None
print(json.dumps({"counter": counter, "frames_per_dispatch": frames, "unit": "KB as reported by rocprofv3 (FETCH_SIZE on gfx950: x2 for wide coalesced reads)", "kernels": out}, indent=1))
