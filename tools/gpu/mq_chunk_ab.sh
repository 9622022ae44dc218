#!/bin/bash
# GPU box: the MQ pass with 1 / 2 / 4 rows between the barriers of its two waves (builds a copy of j2k_analysis.hip with
# another kRowChunk on the box; the production library is put back afterwards).   gpurun -- 'bash tools/gpu/mq_chunk_ab.sh "2 4 1 2 4"'
cd "$GRAFT_REPO_ROOT/ebcc_amd/csrc"
cp ../libh5z_ebcc.so /tmp/libh5z_ebcc.so.keep
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fvisibility=hidden -I../../include"
S='import json,sys; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(d["ms_per_step"], d["encode_GBps"], "t1_mq", k["t1_mq"]["ms_avg"], "t1_encode", k["t1_encode"]["ms_avg"])'
for C in $1; do
  sed "s/^constexpr int kRowChunk = 2;/constexpr int kRowChunk = $C;/" j2k_analysis.hip > j2k_analysis_chunk_ab.hip
  grep -q "kRowChunk = $C;" j2k_analysis_chunk_ab.hip || { echo "kRowChunk not found"; rm -f j2k_analysis_chunk_ab.hip; exit 1; }
  /opt/rocm/bin/hipcc $FLAGS -c j2k_analysis_chunk_ab.hip -o /tmp/j2k_analysis_c$C.o || exit 1
  rm -f j2k_analysis_chunk_ab.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libh5z_ebcc.so engine.o residual_dwt.o residual_spiht.o j2k.o /tmp/j2k_analysis_c$C.o j2k_rate.o search.o host_pool.o batch_codec.o host_codec.o h5z_filter.o -Wl,-rpath,/opt/rocm/lib -ldl -lpthread || exit 1
  echo -n "[rows per barrier $C] "
  (cd "$GRAFT_REPO_ROOT" && timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "$S")
done
cp /tmp/libh5z_ebcc.so.keep ../libh5z_ebcc.so
