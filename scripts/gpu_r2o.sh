#!/bin/bash
# round 2, GPU call 15: candidate loads predicated on the tag test (A/B against unconditional loads)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
make -s -C tests/model 2>/dev/null; make -s -C tests/emu 2>/dev/null
echo "== pytest gpu parity"; timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
for st in silesia mixed; do
  echo "== $st: unconditional loads";  SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_nomask.so STREAM=$st CHUNKS=2048 ONLY=lz4 timeout 300 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids
  echo "== $st: predicated loads";  STREAM=$st CHUNKS=2048 ONLY=lz4 timeout 300 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids
done
for v in ${VARIANTS:-p_nomask prof}; do
  echo "== $v"; SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_$v.so CHUNKS=1024 ONLY=lz4 timeout 200 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids
done
