#!/bin/bash
# GPU box: instruction counts per kernel (which kernels use the issue slots?) - one counter per pass, no API tracing.
#   gpurun --timeout 900 -- 'bash tools/gpu/inst_mix.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/inst
rm -rf $O && mkdir -p $O
for C in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES; do
  EBCC_HIP_SLICES=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$C -- python3 bench.py --steps 1 --warmup 1 --frames 64 --no-cpu-baseline --no-extras > $O/$C.log 2>&1
  echo "$C rc=$?"
  c=$(find $O/$C -name "*counter_collection.csv" | head -1)
  [ -n "$c" ] && python3 tools/pmc_summary.py "$c" $C 64 > $O/$C.json
  rm -rf $O/$C
done
python3 - <<'PY'
import json, os
O = "gpurun_out/inst"
d = {c: json.load(open(f"{O}/{c}.json"))["kernels"] for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES") if os.path.exists(f"{O}/{c}.json")}
ks = sorted(d["SQ_INSTS_VALU"], key=lambda k: -d["SQ_INSTS_VALU"][k]["sum"])
tot = sum(v["sum"] for v in d["SQ_INSTS_VALU"].values())
print(f"{'kernel':24s} {'dispatches':>10s} {'VALU inst (M)':>14s} {'share':>6s} {'SALU (M)':>10s} {'LDS (M)':>9s} {'wave cycles (M)':>16s}")
for k in ks[:22]:
    g = lambda c: d.get(c, {}).get(k, {}).get("sum", 0) / 1e6
    print(f"{k:24s} {d['SQ_INSTS_VALU'][k]['dispatches']:10d} {g('SQ_INSTS_VALU'):14.1f} {100 * d['SQ_INSTS_VALU'][k]['sum'] / tot:5.1f}% {g('SQ_INSTS_SALU'):10.1f} {g('SQ_INSTS_LDS'):9.1f} {g('SQ_WAVE_CYCLES'):16.1f}")
PY
