#!/bin/bash
# GPU box: average duration of selected kernels of the default bench (one slice) under environment variants.
#   gpurun -- 'bash tools/gpu/kstat.sh "k_j2k_level5_fin|k_finest_inv_use" "A=1" "EBCC_T1_TWO_PHASE=0" ...'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PAT=$1; shift
for V in "$@"; do
  O=gpurun_out/kstat_$$; rm -rf $O; mkdir -p $O
  env EBCC_HIP_SLICES=1 $V timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/log.txt 2>&1
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  echo "[$V]"
  python3 - "$f" "$PAT" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(anonymous namespace\)::|ebcc::|void ", "", r["Name"]).split("(")[0]
    if re.search(sys.argv[2], n):
        print(f"   {n[:40]:40s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:9.1f} us total {int(r['TotalDurationNs'])/1e6:8.1f} ms")
PY
  rm -rf $O
done
