#!/bin/bash
# GPU box: phase report of ONE 721x1440 frame through ebcc_encode / ebcc_decode (the HDF5 filter callback's calls).
cd "$GRAFT_REPO_ROOT"
EBCC_HIP_PHASE_TIMING=1 timeout -k 10 200 python tools/gpu/latency.py > gpurun_out/lat_phases.out 2> gpurun_out/lat_phases.err
cat gpurun_out/lat_phases.out
grep -n "phase\|zstd:" gpurun_out/lat_phases.err | sed -n '/analysis/,$p' | tail -${1:-40}
