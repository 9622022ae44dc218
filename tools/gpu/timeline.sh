#!/bin/bash
# GPU box: kernel timeline of one bench step on a single slice (where does a search round spend its time?).
#   gpurun --timeout 900 -- 'bash tools/gpu/timeline.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/timeline
rm -rf $O && mkdir -p $O
export EBCC_HIP_SLICES=${SLICES:-1}
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/trace.log 2>&1
echo "trace rc=$?"
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 tools/trace_summary.py "$f" 25 > $O/summary.txt
rm -rf $O/trace
cat $O/summary.txt
