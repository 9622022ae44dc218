#!/bin/bash
# GPU box: where a k_rate call spends its time (a build of j2k_rate.hip with -DEBCC_RATE_PROFILE prints frame 0's split; the
# production library is put back afterwards).   gpurun -- 'bash tools/gpu/rate_profile.sh'
cd "$GRAFT_REPO_ROOT/ebcc_amd/csrc"
cp ../libh5z_ebcc.so /tmp/libh5z_ebcc.so.keep
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fvisibility=hidden -I../../include"
/opt/rocm/bin/hipcc $FLAGS -DEBCC_RATE_PROFILE -c j2k_rate.hip -o /tmp/j2k_rate_prof.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libh5z_ebcc.so engine.o residual_dwt.o residual_spiht.o j2k.o j2k_analysis.o /tmp/j2k_rate_prof.o search.o host_pool.o batch_codec.o host_codec.o h5z_filter.o -Wl,-rpath,/opt/rocm/lib -ldl -lpthread || exit 1
cd "$GRAFT_REPO_ROOT"
EBCC_HIP_SLICES=${SLICES:-1} timeout -k 10 300 python bench.py --frames ${1:-85} --steps 1 --warmup 0 --no-cpu-baseline --no-extras 2>&1 | grep "k_rate frame0:" | head -${2:-60}
cp /tmp/libh5z_ebcc.so.keep ebcc_amd/libh5z_ebcc.so
