#!/bin/bash
# round 2, fourth GPU call: new parse loop + honest phase profile; then the reference's REAL daemons on the GPU box
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
make -s -C tests/model; make -s -C tests/emu
echo "== parity"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
echo "== shipping build"; CHUNKS=2048 timeout 300 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids
echo "== shipping build, mixed"; STREAM=mixed CHUNKS=2048 ONLY=lz4 timeout 300 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids
echo "== phase profile (prof build)"; SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_prof.so ONLY=lz4 CHUNKS=1024 timeout 300 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids | head -16
echo "== phase profile, mixed"; STREAM=mixed SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_prof.so ONLY=lz4 CHUNKS=1024 timeout 300 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids | head -16
echo "== bench default"
timeout 900 python bench.py --steps 5 --warmup 1 2>&1 | grep "^{" | tee gpurun_out/bench_r2d.json | cut -c1-1500
echo "== host resources"; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; df -h /dev/shm | tail -1; free -g | head -2
echo "== configs[0]: reference daemons, CPU path, 200 x 8 MiB PRNG chunks, 32 connections"
timeout 900 python oracle/ref_daemon.py --chunks 200 --chunk-kib 8192 --connections 32 --stream random --timeout 700 --out gpurun_out/r2_config0_200.json 2>&1 | tail -3 | cut -c1-800
echo "== configs[4]: reference daemons + gpu_compress on the MI355X, 200 x 8 MiB, 32 connections"
timeout 900 python oracle/ref_daemon.py --chunks 200 --chunk-kib 8192 --connections 32 --gpu-op --context hip --max-batch 64 --timeout 700 --out gpurun_out/r2_config4_200.json 2>&1 | tail -3 | cut -c1-800
