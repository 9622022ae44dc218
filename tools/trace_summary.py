#!/usr/bin/env python3
"""Timeline summary of a rocprofv3 --kernel-trace CSV: GPU busy / idle time, the largest gaps and what ran around
them, per-kernel totals - for the LAST bench step (everything after the largest gap-free k_scale_shift start).
    python3 tools/trace_summary.py <kernel_trace.csv> [n_gaps]"""
import csv
import re
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name)
        name = name.split("(")[0].replace("ebcc::", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?")))
rows.sort()
# last step = from the first k_in_init (input statistics open an encode slice) of the LAST cluster of them: the slices of a
# step start within a few milliseconds of each other, the steps are a whole step apart
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_in_init")]
first = starts[-1]
for i in reversed(starts[:-1]):
    if rows[first][0] - rows[i][0] > 30_000_000:
        break
    first = i
step = rows[first:]
t0, t1 = step[0][0], max(r[1] for r in step)
# union of busy intervals
busy, cur_s, cur_e = 0, None, None
gaps = []
for s, e, n, _q in step:
    if cur_e is None:
        cur_s, cur_e, last = s, e, n
    elif s <= cur_e:
        if e > cur_e:
            cur_e, last = e, n
    else:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e - t0, last, n))
        cur_s, cur_e, last = s, e, n
busy += cur_e - cur_s
print(f"step: {(t1 - t0) / 1e6:.2f} ms wall, GPU busy (any kernel running) {busy / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms, "
      f"sum of kernel durations {sum(r[1] - r[0] for r in step) / 1e6:.2f} ms, {len(step)} kernels")
per = {}
for s, e, n, _q in step:
    a = per.setdefault(n, [0, 0])
    a[0] += e - s
    a[1] += 1
print("kernel totals (ms, launches):")
for n, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {t / 1e6:8.2f} {c:5d}  {n}")
ng = int(sys.argv[2]) if len(sys.argv) > 2 else 15
print("largest idle gaps (ms at offset ms: after -> before):")
for g, off, a, b in sorted(gaps, reverse=True)[:ng]:
    print(f"  {g / 1e6:7.3f} at {off / 1e6:8.2f}: {a} -> {b}")
tot = sum(g for g, *_ in gaps)
print(f"gaps: {len(gaps)}, total {tot / 1e6:.2f} ms; gaps > 100 us: {sum(1 for g, *_ in gaps if g > 100_000)} totalling {sum(g for g, *_ in gaps if g > 100_000) / 1e6:.2f} ms")
# rounds: time between consecutive k_rate launches
kr = [r[0] for r in step if r[2] == "k_rate"]
if len(kr) > 1:
    d = [(b - a) / 1e6 for a, b in zip(kr, kr[1:])]
    print("k_rate to k_rate (ms):", " ".join(f"{x:.2f}" for x in d))

# per hardware queue: span, kernel time, time with nothing of this queue running (host turnaround + queueing)
qs = {}
for s_, e_, n_, q_ in step:
    qs.setdefault(q_, []).append((s_, e_, n_))
print("per queue: first..last (ms), kernels, kernel time (ms), idle inside the span (ms)")
for q_, ks in sorted(qs.items(), key=lambda kv: kv[1][0][0]):
    ks.sort()
    span0, span1 = ks[0][0], max(e for _, e, _ in ks)
    kt = sum(e - s for s, e, _ in ks)
    print(f"  queue {q_}: {(span0 - t0) / 1e6:7.1f} .. {(span1 - t0) / 1e6:7.1f}  {len(ks):5d}  {kt / 1e6:8.1f}  {(span1 - span0 - kt) / 1e6:8.1f}   first {ks[0][2]}  last {ks[-1][2]}")
