#!/bin/bash
# round 2, second GPU call: phase profile of the slice-parallel compressor, block queue + MD5 workgroup-size experiments
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
make -s -C tests/model; make -s -C tests/emu
echo "== parity (quick)"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
echo "== phase profile (prof build)"; SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_prof.so CHUNKS=512 timeout 300 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids
echo "== phase profile, mixed stream"; STREAM=mixed SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_prof.so CHUNKS=512 timeout 300 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids | head -20
for wg in 64 256 512 1024; do echo "== shipping build, MD5 workgroup $wg"; SKYHIP_MD5_WG=$wg CHUNKS=2048 timeout 300 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids; done
echo "== bench default"
timeout 900 python bench.py --steps 5 --warmup 1 2>&1 | grep "^{" | tee gpurun_out/bench_r2b.json | cut -c1-1800
