#!/bin/bash
# round 2, GPU call 13: phase tables of the compressor with one change at a time (table layout, aligned parse loads, OR emit, visit cap, extension cap)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
make -s -C tests/model; make -s -C tests/emu
echo "== pytest gpu parity, variant base"; SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_p_base.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
for v in ${VARIANTS:-p_base p_soa p_ald p_ore p_v8 p_ext64}; do
  echo "== $v"; SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_$v.so CHUNKS=1024 ONLY=lz4 timeout 200 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids
done
