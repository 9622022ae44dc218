#!/bin/bash
# GPU box: vector and scalar instructions issued per kernel over ONE step of the default workload (one slice, so that every
# dispatch belongs to one stage): which kernels fill the issue slots the slices compete for?
#   gpurun --timeout 900 -- 'bash tools/gpu/valu_census.sh [frames]'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
F=${1:-256}
O=gpurun_out/census
rm -rf $O && mkdir -p $O
for C in SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES; do
  EBCC_HIP_SLICES=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$C -- python3 bench.py --steps 1 --warmup 0 --frames $F --no-cpu-baseline --no-extras > $O/$C.log 2>&1
  echo "$C rc=$?"
  c=$(find $O/$C -name "*counter_collection.csv" | head -1)
  [ -n "$c" ] && python3 tools/pmc_summary.py "$c" $C $F > $O/$C.json
  rm -rf $O/$C
done
python3 - $F <<'PY'
import json, sys
O = "gpurun_out/census"
v = json.load(open(O + "/SQ_INSTS_VALU.json"))["kernels"]; s = json.load(open(O + "/SQ_INSTS_SALU.json"))["kernels"]; w = json.load(open(O + "/SQ_WAVES.json"))["kernels"]
tot = sum(x["sum"] for x in v.values())
print("one step of %s frames, one slice: %.2f G vector instructions (wave64 instructions; 1024 SIMDs issue one per 4 cycles: %.1f ms of all SIMDs at 2.1 GHz)" % (sys.argv[1], tot / 1e9, tot / 1024 * 4 / 2.1e6))
print("%-28s %6s %10s %6s %10s %10s" % ("kernel", "disp", "VALU M", "%", "SALU M", "waves k"))
for k, x in sorted(v.items(), key=lambda kv: -kv[1]["sum"])[:28]:
    print("%-28s %6d %10.1f %6.1f %10.1f %10.1f" % (k[:28], x["dispatches"], x["sum"] / 1e6, 100 * x["sum"] / tot, s.get(k, {}).get("sum", 0) / 1e6, w.get(k, {}).get("sum", 0) / 1e3))
PY
