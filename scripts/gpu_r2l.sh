#!/bin/bash
# round 2, GPU call 12: aligned-only LDS traffic (dword reads + v_alignbit in the parse, OR-accumulated emit), SoA table
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
make -s -C tests/model; make -s -C tests/emu
echo "== pytest gpu parity"; timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
PROF=$PWD/scripts/dev/libskyhip_prof.so
for st in silesia mixed; do
  echo "== $st: new";  STREAM=$st CHUNKS=2048 ONLY=lz4 timeout 300 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids
done
echo "== silesia: new, phase table"; SKYHIP_LIB_PATH=$PROF CHUNKS=1024 ONLY=lz4 timeout 200 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2l_lz4s_phases.txt
