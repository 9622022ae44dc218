#!/bin/bash
# GPU box: parity suite (both tier-1 encoder variants with BOTH=1).  Fails on a red test AND on a GPU memory fault in the log
# (the runtime can report one and still exit 0).   gpurun -- 'bash tools/gpu/check.sh [log-name]'
L=gpurun_out/${1:-check}.log
mkdir -p gpurun_out
( timeout -k 10 900 python -X faulthandler -m pytest tests -m gpu -x -q 2>&1 ) > $L 2>&1
rc=$?
if [ $rc -eq 0 ] && [ -n "$BOTH" ]; then ( EBCC_T1_TWO_PHASE=0 timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 ) >> $L 2>&1; rc=$?; fi
tail -6 $L
if grep -q "Memory access fault\|Fatal Python error" $L; then echo "GPU FAULT OR ABORT IN THE TEST RUN"; grep -n -B8 "Memory access fault\|Fatal Python error" $L | head -40; exit 3; fi
exit $rc
