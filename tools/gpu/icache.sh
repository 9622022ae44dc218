#!/bin/bash
# GPU box: instruction-cache and wait counters of the tier-1 kernels (is a kernel waiting for instructions?) - one counter per pass.
#   gpurun --timeout 900 -- 'bash tools/gpu/icache.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/icache
rm -rf $O && mkdir -p $O
rocprofv3 --list-avail 2>/dev/null | grep -o "SQC\?_[A-Z_0-9]*" | sort -u > $O/avail.txt
for C in ${COUNTERS:-SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_IFETCH SQ_WAIT_ANY SQ_ACTIVE_INST_ANY}; do
  EBCC_HIP_SLICES=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$C -- python3 bench.py --steps 1 --warmup 1 --frames ${FRAMES:-64} --no-cpu-baseline --no-extras > $O/$C.log 2>&1
  echo "$C rc=$?"
  c=$(find $O/$C -name "*counter_collection.csv" | head -1)
  [ -n "$c" ] && python3 tools/pmc_summary.py "$c" $C ${FRAMES:-64} > $O/$C.json
  rm -rf $O/$C
done
python3 - <<'PY'
import json, os, glob
O = "gpurun_out/icache"
d = {}
for f in glob.glob(f"{O}/*.json"):
    d[os.path.basename(f)[:-5]] = json.load(open(f))["kernels"]
cs = sorted(d)
ks = ["k_t1_decode_lds", "k_t1_decode", "k_t1_resume", "k_t1_mqrows", "k_t1_emit", "k_t1_scan", "k_rate", "k_spiht_decode", "k_spiht_encode"]
print("kernel".ljust(20), " ".join(c[-18:].rjust(18) for c in cs))
for k in ks:
    print(k.ljust(20), " ".join(f"{d[c].get(k, {}).get('sum', 0) / 1e6:18.2f}" for c in cs))
PY
