#!/bin/bash
# device-only assembly of libskyhip for gfx950 + resource summary of one kernel (default: sky_lz4s_compress)
K=${1:-sky_lz4s_compress}
cd /root/repo/skyplane_amd/csrc || exit 1
mkdir -p /tmp/t
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -DSKY_WITH_CDC $EXTRA -S --cuda-device-only -o /tmp/t/skyhip.s skyhip.hip 2>&1 | grep -v "hip-link" | grep -E "warning|error" | head
awk "/^$K:/,/\.end_amdhsa_kernel/" /tmp/t/skyhip.s > /tmp/t/$K.s
echo "lines: $(wc -l < /tmp/t/$K.s)"
grep -A40 "name:.*$K\$" /tmp/t/skyhip.s | grep -E "vgpr|sgpr|private_segment|group_segment_fixed|spill"
grep -E "^\s+(ds_|flat_|global_|scratch_|s_set_gpr|buffer_)" /tmp/t/$K.s | awk '{print $1}' | sort | uniq -c | sort -rn | head -${2:-12}
