#!/bin/bash
# round 2, GPU call 21: compressor waves at base priority 1 (MD5 waves, at 0, take the idle issue slots only): lz4 + md5 side by side
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export HSA_ENABLE_IPC_MODE_LEGACY=0
for v in shipping prio1; do
  if [ $v = shipping ]; then unset SKYHIP_LIB_PATH; else export SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_$v.so; fi
  echo "== $v"; CHUNKS=2048 timeout 300 python scripts/dev/lz4s_exp.py 2>&1 | grep -v "amdgpu.ids\|^md5"
  echo "== $v: bench"; timeout 600 python bench.py --no-cpu-baseline --verify sample 2>&1 | grep "^{" | cut -c1-300
done
