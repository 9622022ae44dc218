#!/bin/bash
# GPU box: SQ counters of chosen kernels (default: the tier-1 kernels), one counter per pass, no API tracing.
#   gpurun --timeout 900 -- 'bash tools/gpu/t1_pmc.sh [kernel-regex] [frames]'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
K=${1:-k_t1_}
F=${2:-64}
O=gpurun_out/t1pmc
rm -rf $O && mkdir -p $O
for C in SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT; do
  EBCC_HIP_SLICES=1 EBCC_HIP_T1_STATS=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$C -- python3 bench.py --steps 1 --warmup 1 --frames $F --no-cpu-baseline --no-extras > $O/$C.log 2>&1
  echo "$C rc=$?"
  c=$(find $O/$C -name "*counter_collection.csv" | head -1)
  [ -n "$c" ] && python3 tools/pmc_summary.py "$c" $C $F > $O/$C.json
  rm -rf $O/$C
done
grep "tier-1:" $O/SQ_WAVE_CYCLES.log | head -2
python3 - "$K" <<'PY'
import json, os, re, sys, glob
O = "gpurun_out/t1pmc"
pat = re.compile(sys.argv[1])
d = {os.path.basename(f)[:-5]: json.load(open(f))["kernels"] for f in glob.glob(O + "/*.json")}
ks = [k for k in d.get("SQ_WAVE_CYCLES", {}) if pat.search(k)]
cols = sorted(d)
print("kernel".ljust(22), "disp".rjust(5), *(c.replace("SQ_", "").rjust(15) for c in cols))
for k in ks:
    print(k.ljust(22), str(d["SQ_WAVE_CYCLES"][k]["dispatches"]).rjust(5), *(f"{d[c].get(k, {}).get('per_dispatch', 0) / 1e6:15.3f}" for c in cols))
print("(millions per dispatch; SQ_WAVE_CYCLES / WAIT_* / ACTIVE_* count quad-cycles)")
PY
