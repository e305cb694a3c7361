#!/bin/bash
# round 2, tenth GPU call: compressor with every-second-position table insertion + region-specialised parse (A/B against the committed one),
# the pipelined decode entry point, host path in lanes, steady-state loopback with sendfile / mmap receive
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
make -s -C tests/model; make -s -C tests/emu
echo "== pytest -m gpu (everything)"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r2j_pytest_gpu.log
OLD=$PWD/scripts/dev/libskyhip_old.so; PROF=$PWD/scripts/dev/libskyhip_prof.so
for st in silesia mixed; do
  echo "== $st: old";  STREAM=$st SKYHIP_LIB_PATH=$OLD CHUNKS=2048 ONLY=lz4 timeout 200 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids
  echo "== $st: new";  STREAM=$st CHUNKS=2048 timeout 300 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids
done
echo "== silesia: new, phase table"; SKYHIP_LIB_PATH=$PROF CHUNKS=1024 ONLY=lz4 timeout 200 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2j_lz4s_phases.txt
echo "== bench default"; timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep "^{" | tee gpurun_out/r2j_bench_default.json | cut -c1-1200
echo "== host path"; timeout 300 python scripts/host_path_bench.py --chunks 256 --max-batch 64 --skip-pageable 2> gpurun_out/r2j_hostpath.err | tee gpurun_out/r2j_hostpath.json; tail -3 gpurun_out/r2j_hostpath.err
echo "== e2e steady, hip"; E2E_TRACE=1 timeout 300 python scripts/e2e_steady.py --chunks 1024 --connections 32 --workers 2 --max-batch 64 2> gpurun_out/r2j_e2e.err | tail -1 | tee gpurun_out/r2j_e2e_steady.json | cut -c1-700; grep trace gpurun_out/r2j_e2e.err
echo "== e2e steady, null context (plumbing only)"; E2E_TRACE=1 timeout 300 python scripts/e2e_steady.py --context null --chunks 1024 --connections 32 --workers 2 --max-batch 64 2> gpurun_out/r2j_e2e_null.err | tail -1 | tee gpurun_out/r2j_e2e_null.json | cut -c1-700; grep trace gpurun_out/r2j_e2e_null.err
