#!/bin/bash
# VGPR / SGPR / scratch of the tier-1 kernels (development aid): tools/vgpr.sh
cd "$(dirname "$0")/../ebcc_amd/csrc"
for f in j2k_analysis j2k_rate; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -I../../include -S --cuda-device-only -o /tmp/$f.s $f.hip 2>/dev/null
  awk '/\.name:/{n=$2} /\.vgpr_count:/{v=$2} /\.sgpr_count:/{s=$2} /\.private_segment_fixed_size:/{p=$2} /\.vgpr_spill_count:/{ if (n ~ /t1_/) printf "%-28s vgpr %s sgpr %s scratch %s spill %s\n", substr(n, index(n,"k_t1"), 20), v, s, p, $2}' /tmp/$f.s
done
