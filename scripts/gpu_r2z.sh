#!/bin/bash
# round 2, final GPU call: whole GPU suite, smoke, PMC traffic of the shipping compressor, default bench + rocprofv3 kernel stats, mixed / cdc lines,
# host path, steady-state loopback, phase tables.  Everything under `timeout`, outputs under gpurun_out/ (copied to profiles/ afterwards).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
make -s -C tests/model 2>/dev/null; make -s -C tests/emu 2>/dev/null
echo "== pytest -m gpu (everything)"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r2_pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== PMC traffic (separate passes)"
for st in silesia mixed; do for ctr in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && export TMPDIR=/tmp && STREAM=$st ONLY=lz4 CHUNKS=2048 timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OLDPWD/gpurun_out/pmc_r2z -o ${st}_${ctr} -- python $OLDPWD/scripts/dev/lz4s_exp.py > $OLDPWD/gpurun_out/pmc_r2z_${st}_${ctr}.log 2>&1 )
done; python scripts/pmc_traffic.py gpurun_out/pmc_r2z $st 2048 sky_lz4s_compress; done
find gpurun_out/pmc_r2z -name "*kernel_trace.csv" -delete
cp profiles/traffic.json gpurun_out/r2_traffic.json
echo "== bench default"; timeout 900 python bench.py 2>&1 | grep "^{" | tee gpurun_out/r2_bench_default.json | cut -c1-2200
echo "== bench default under rocprofv3 --kernel-trace --stats"
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_default -o def -- python $OLDPWD/bench.py --no-cpu-baseline --verify sample > $OLDPWD/gpurun_out/bench_default_prof.log 2>&1 )
grep "^{" gpurun_out/bench_default_prof.log | tee gpurun_out/r2_bench_default_under_rocprof.json | cut -c1-600; head -8 gpurun_out/prof_default/def_kernel_stats.csv
find gpurun_out/prof_default -name "*kernel_trace.csv" -delete
echo "== bench --stream mixed --chunks 16384"; timeout 900 python bench.py --stream mixed --chunks 16384 --steps 3 --no-cpu-baseline 2>&1 | grep "^{" | tee gpurun_out/r2_bench_mixed.json | cut -c1-900
echo "== bench --cdc"; timeout 900 python bench.py --cdc --steps 3 --no-cpu-baseline 2>&1 | grep "^{" | tee gpurun_out/r2_bench_cdc.json | cut -c1-1000
echo "== host path (512 chunks = 4 GiB)"; timeout 400 python scripts/host_path_bench.py --chunks 512 --max-batch 64 --skip-pageable 2> gpurun_out/r2_hostpath.err | tee gpurun_out/r2_host_path.json; tail -2 gpurun_out/r2_hostpath.err
echo "== e2e steady, hip, 2 workers"; E2E_TRACE=1 timeout 300 python scripts/e2e_steady.py --chunks 1024 --connections 32 --workers 2 --max-batch 64 2> gpurun_out/r2_e2e.err | tail -1 | tee gpurun_out/r2_e2e_steady.json | cut -c1-500; grep trace gpurun_out/r2_e2e.err
echo "== e2e steady, null context, 2 workers"; timeout 300 python scripts/e2e_steady.py --context null --chunks 1024 --connections 32 --workers 2 --max-batch 64 2>/dev/null | tail -1 | tee gpurun_out/r2_e2e_steady_null.json | cut -c1-500
echo "== phases"; SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_prof.so CHUNKS=1024 ONLY=lz4 timeout 200 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_lz4s_phases.txt
echo "== by wave"; BY_WAVE=1 SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_prof.so CHUNKS=512 ONLY=lz4 timeout 200 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_lz4s_by_wave.txt
echo "== linked decode"; timeout 200 python scripts/dev/linked_decode.py 2>&1 | grep -v amdgpu.ids | tail -8 | tee gpurun_out/r2_linked_decode.txt
