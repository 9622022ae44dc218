#!/bin/bash
# GPU box: isolate a failing env-switch case (which toggle changes the outcome)
cd "$GRAFT_REPO_ROOT"
T='tests/test_codec_gpu.py::test_env_switches_match_oracle'
for E in "" "EBCC_HIP_NO_SPECULATION=1" "EBCC_HIP_HOST_SEARCH=1" "EBCC_HIP_RESIDUAL_SPLIT=1"; do
  echo "== [$E]"
  env $E timeout -k 10 200 python -m pytest "$T" -q -m gpu 2>&1 | tail -4
done
