#!/bin/bash
# round 2, sixth GPU call (short): A/B of the pre-pass pre-check, MD5 prefetch depth x workgroup size
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== shipping"; CHUNKS=2048 ONLY=lz4 timeout 200 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids
echo "== precheck"; SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_precheck.so CHUNKS=2048 ONLY=lz4 timeout 200 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids
echo "== precheck, mixed"; STREAM=mixed SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_precheck.so CHUNKS=2048 ONLY=lz4 timeout 200 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids
for lib in libskyhip_md5pf2 libskyhip_md5pf1; do for wg in 64 128 256; do
  echo "== $lib MD5 workgroup $wg"; SKYHIP_LIB_PATH=$PWD/scripts/dev/$lib.so SKYHIP_MD5_WG=$wg CHUNKS=2048 ONLY=md5 timeout 200 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids
done; done
echo "== shipping MD5 wg 128"; SKYHIP_MD5_WG=128 CHUNKS=2048 ONLY=md5 timeout 200 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids
echo "== phases (shipping source, prof build)"; SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_prof.so CHUNKS=1024 ONLY=lz4 timeout 200 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids | head -15
echo "== configs[4] at 6 connections (same as the configs[0] run): reference daemons + gpu_compress on the MI355X, 1000 x 8 MiB"
timeout 330 python oracle/ref_daemon.py --chunks 1000 --chunk-kib 8192 --connections 6 --gpu-op --context hip --max-batch 64 --workers 1 --timeout 240 --stream random --out gpurun_out/r2_config4_6.json > gpurun_out/r2_config4_6.log 2>&1; tail -1 gpurun_out/r2_config4_6.log | cut -c1-700; grep -v "^\[\|DEBUG\|INFO" gpurun_out/r2_config4_6.log | tail -25 | cut -c1-300
