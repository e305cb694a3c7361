#!/bin/bash
# round 2, GPU call 14: shipping compressor after the clean-up (b128 rows, four block loads in flight); hardware queues (GPU_MAX_HW_QUEUES) for the
# multi-context host path, the bench and the loopback; loopback with 4 workers per side
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
make -s -C tests/model 2>/dev/null; make -s -C tests/emu 2>/dev/null
echo "== pytest gpu parity"; timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
for st in silesia mixed; do
  echo "== $st: shipping";  STREAM=$st CHUNKS=2048 timeout 300 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids
done
echo "== silesia: phase table"; SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_prof.so CHUNKS=1024 ONLY=lz4 timeout 200 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2n_lz4s_phases.txt
for q in default 16; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  echo "== host path, GPU_MAX_HW_QUEUES=$q"; timeout 300 python scripts/host_path_bench.py --chunks 256 --max-batch 64 --skip-pageable 2> gpurun_out/r2n_hostpath_$q.err | tee gpurun_out/r2n_hostpath_$q.json
  echo "== bench default, GPU_MAX_HW_QUEUES=$q"; timeout 600 python bench.py --no-cpu-baseline --verify sample 2>&1 | grep "^{" | tee gpurun_out/r2n_bench_$q.json | cut -c1-420
  echo "== e2e steady hip, 4 workers, GPU_MAX_HW_QUEUES=$q"; E2E_TRACE=1 timeout 300 python scripts/e2e_steady.py --chunks 1024 --connections 32 --workers 4 --max-batch 64 2> gpurun_out/r2n_e2e_$q.err | tail -1 | tee gpurun_out/r2n_e2e_steady_w4_$q.json | cut -c1-500; grep trace gpurun_out/r2n_e2e_$q.err
done
unset GPU_MAX_HW_QUEUES
echo "== e2e steady null, 4 workers"; E2E_TRACE=1 timeout 300 python scripts/e2e_steady.py --context null --chunks 1024 --connections 32 --workers 4 --max-batch 64 2> gpurun_out/r2n_e2e_null.err | tail -1 | cut -c1-500; grep trace gpurun_out/r2n_e2e_null.err
