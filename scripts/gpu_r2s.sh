#!/bin/bash
# round 2, GPU call 20: queue ticket drawn beside the flush
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
make -s -C tests/model 2>/dev/null; make -s -C tests/emu 2>/dev/null
echo "== pytest gpu parity"; timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
for st in silesia mixed; do
  echo "== $st: before (masked loads, no priority, __syncthreads)";  SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_noprio.so STREAM=$st CHUNKS=2048 ONLY=lz4 timeout 300 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids
  echo "== $st: now";  STREAM=$st CHUNKS=2048 timeout 300 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids
done
echo "== by wave"; BY_WAVE=1 SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_prof.so CHUNKS=512 ONLY=lz4 timeout 200 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2s_lz4s_by_wave.txt
echo "== phases"; SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_prof.so CHUNKS=1024 ONLY=lz4 timeout 200 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2s_lz4s_phases.txt
