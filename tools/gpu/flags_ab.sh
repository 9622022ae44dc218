#!/bin/bash
# GPU box: one source file of the library built with other compiler flags (the production library is put back afterwards).
#   gpurun -- 'bash tools/gpu/flags_ab.sh j2k_rate.hip "" "-O2" "-mllvm -amdgpu-sched-strategy=max-ilp"'
cd "$GRAFT_REPO_ROOT/ebcc_amd/csrc"
cp ../libh5z_ebcc.so /tmp/libh5z_ebcc.so.keep
F=$1; shift
BASE="--offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fvisibility=hidden -I../../include"
S='import json,sys; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(d["ms_per_step"], d["encode_GBps"], d["decode_GBps"], {n: v["ms_avg"] for n, v in k.items() if n in ("t1_mq","t1_symbols","t1_decode","t1_probe_decode","rate_alloc","spiht_decode","spiht_encode")})'
OBJS=""
for o in engine residual_dwt residual_spiht j2k j2k_analysis j2k_rate search host_pool batch_codec host_codec h5z_filter; do
  if [ "$o.hip" == "$F" ]; then OBJS="$OBJS /tmp/flags_ab.o"; else OBJS="$OBJS $o.o"; fi
done
for V in "$@"; do
  case "$V" in *-O*) OPT="";; *) OPT="-O3";; esac
  /opt/rocm/bin/hipcc $BASE $OPT $V -c $F -o /tmp/flags_ab.o 2>/dev/null || { echo "[$V] does not build"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libh5z_ebcc.so $OBJS -Wl,-rpath,/opt/rocm/lib -ldl -lpthread || exit 1
  echo -n "[$F $V] "
  (cd "$GRAFT_REPO_ROOT" && timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "$S")
done
cp /tmp/libh5z_ebcc.so.keep ../libh5z_ebcc.so
