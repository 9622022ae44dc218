#!/bin/bash
# GPU box: kernel timelines of one default bench step under several environment variants (three slices).
#   gpurun --timeout 900 -- 'bash tools/gpu/timeline_ab.sh "A=1" "EBCC_HIP_NO_EARLY_GROUP=1" ...'   ('+' joins variables)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for V in "$@"; do
  i=$((i+1))
  O=gpurun_out/tl_$i
  rm -rf $O && mkdir -p $O
  env ${V//+/ } timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/trace.log 2>&1
  echo "[$V] trace rc=$?"
  f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
  python3 tools/trace_summary.py "$f" 12 > $O/summary.txt
  rm -rf $O/trace
  grep '^{' $O/trace.log | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["encode_GBps"], d["decode_GBps"])'
  head -40 $O/summary.txt
done
