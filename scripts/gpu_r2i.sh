#!/bin/bash
# round 2, ninth GPU call: compressor at 96 VGPRs + MD5 at 128 so that MD5 / frame-gather waves fit beside compressor workgroups
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export HSA_ENABLE_IPC_MODE_LEGACY=0
L=$PWD/scripts/dev/libskyhip_coexist.so
echo "== shipping"; CHUNKS=2048 timeout 300 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids
echo "== coexist build"; SKYHIP_LIB_PATH=$L CHUNKS=2048 timeout 300 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids
echo "== coexist build, MD5 wg 64, SKYHIP_DEBUG"; SKYHIP_DEBUG=1 SKYHIP_LIB_PATH=$L SKYHIP_MD5_WG=64 timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep "^{\|skyhip\]" | cut -c1-900
echo "== coexist build, MD5 auto"; SKYHIP_LIB_PATH=$L timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep "^{" | cut -c1-900
echo "== coexist, mixed 16384"; SKYHIP_LIB_PATH=$L SKYHIP_MD5_WG=64 timeout 600 python bench.py --stream mixed --chunks 16384 --steps 3 --no-cpu-baseline 2>&1 | grep "^{" | cut -c1-400
