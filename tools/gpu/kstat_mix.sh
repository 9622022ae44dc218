#!/bin/bash
# GPU box: average duration of every kernel when one slice of 85 frames runs alone against the default three slices of a
# 256-frame step (what do the slices cost each other?).   gpurun -- 'bash tools/gpu/kstat_mix.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {   # name, frames, slices
  O=gpurun_out/kmix_$1; rm -rf $O; mkdir -p $O
  env EBCC_HIP_SLICES=$3 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --frames $2 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/log.txt 2>&1
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  cp "$f" gpurun_out/kmix_$1.csv; rm -rf $O
}
run alone 85 1
run mix 256 3
python3 - <<'PY'
import csv, re
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        n = re.sub(r"\(anonymous namespace\)::|ebcc::|void ", "", r["Name"]).split("(")[0]
        d[n] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / 1e6)
    return d
a, m = load("gpurun_out/kmix_alone.csv"), load("gpurun_out/kmix_mix.csv")
print("%-36s %7s %9s %9s   %7s %9s %9s  ratio" % ("kernel", "calls", "avg us", "tot ms", "calls", "avg us", "tot ms"))
ta = tm = 0
for n in sorted(m, key=lambda k: -m[k][2]):
    if n not in a or n.startswith("at::") : continue
    ca, aa, sa = a[n]; cm, am, sm = m[n]
    ta += sa; tm += sm
    print("%-36s %7d %9.1f %9.1f   %7d %9.1f %9.1f  %.2f" % (n[:36], ca, aa, sa, cm, am, sm, am / aa if aa else 0))
print("total kernel time (4 steps incl. warm-up): alone %.1f ms, mix %.1f ms" % (ta, tm))
PY
