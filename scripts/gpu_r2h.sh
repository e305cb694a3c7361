#!/bin/bash
# round 2, eighth GPU call: frame assembly overlapped with the next sub-batch; adaptive pre-check; final numbers
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
make -s -C tests/model; make -s -C tests/emu
echo "== pytest -m gpu (everything)"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r2_pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for st in silesia mixed; do
  echo "== $st: shipping"; STREAM=$st CHUNKS=2048 ONLY=lz4 timeout 300 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids
  echo "== $st: 10 visits"; STREAM=$st SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_visits10.so CHUNKS=2048 ONLY=lz4 timeout 300 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids
done
echo "== bench default"; timeout 900 python bench.py 2>&1 | grep "^{" | tee gpurun_out/r2_bench_default.json | cut -c1-1500
echo "== bench default, MD5 one wave per CU"; SKYHIP_MD5_WG=64 timeout 900 python bench.py --no-cpu-baseline 2>&1 | grep "^{" | cut -c1-200
echo "== bench default, visits 10"; SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_visits10.so timeout 900 python bench.py --no-cpu-baseline --verify sample 2>&1 | grep "^{" | cut -c1-700
echo "== bench default under rocprofv3 --kernel-trace --stats"
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_default -o def -- python $OLDPWD/bench.py --no-cpu-baseline > $OLDPWD/gpurun_out/bench_default_prof.log 2>&1 )
grep "^{" gpurun_out/bench_default_prof.log | tee gpurun_out/r2_bench_default_under_rocprof.json | cut -c1-300; head -6 gpurun_out/prof_default/def_kernel_stats.csv
find gpurun_out/prof_default -name "*kernel_trace.csv" -delete
echo "== bench --stream mixed --chunks 16384"; timeout 900 python bench.py --stream mixed --chunks 16384 --steps 3 --no-cpu-baseline 2>&1 | grep "^{" | tee gpurun_out/r2_bench_mixed.json | cut -c1-900
echo "== bench --cdc"; timeout 900 python bench.py --cdc --steps 3 --no-cpu-baseline 2>&1 | grep "^{" | tee gpurun_out/r2_bench_cdc.json | cut -c1-1000
