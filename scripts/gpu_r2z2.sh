#!/bin/bash
# round 2, final GPU call (lean: 8 GPU-minutes were left): parity, default bench, rocprofv3 kernel stats, PMC traffic, mixed / cdc lines, phase table,
# host path, steady-state loopback -- most important first, every step under its own short timeout
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
make -s -C tests/model 2>/dev/null; make -s -C tests/emu 2>/dev/null
echo "== pytest gpu parity"; timeout 120 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
echo "== bench default"; timeout 240 python bench.py 2>&1 | grep "^{" | tee gpurun_out/r2_bench_default.json | cut -c1-2400
echo "== bench default under rocprofv3 --kernel-trace --stats"
( cd /tmp && export TMPDIR=/tmp && timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_default -o def -- python $OLDPWD/bench.py --no-cpu-baseline --verify sample > $OLDPWD/gpurun_out/bench_default_prof.log 2>&1 )
grep "^{" gpurun_out/bench_default_prof.log | tee gpurun_out/r2_bench_default_under_rocprof.json | cut -c1-500; head -8 gpurun_out/prof_default/def_kernel_stats.csv
find gpurun_out/prof_default -name "*kernel_trace.csv" -delete
echo "== PMC traffic, silesia (separate passes)"
for ctr in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && export TMPDIR=/tmp && STREAM=silesia ONLY=lz4 CHUNKS=2048 timeout 90 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OLDPWD/gpurun_out/pmc_r2z -o silesia_${ctr} -- python $OLDPWD/scripts/dev/lz4s_exp.py > $OLDPWD/gpurun_out/pmc_r2z_silesia_${ctr}.log 2>&1 )
done; python scripts/pmc_traffic.py gpurun_out/pmc_r2z silesia 2048 sky_lz4s_compress
find gpurun_out/pmc_r2z -name "*kernel_trace.csv" -delete
cp profiles/traffic.json gpurun_out/r2_traffic.json
echo "== bench --stream mixed --chunks 16384"; timeout 200 python bench.py --stream mixed --chunks 16384 --steps 3 --no-cpu-baseline --verify sample 2>&1 | grep "^{" | tee gpurun_out/r2_bench_mixed.json | cut -c1-700
echo "== bench --cdc"; timeout 200 python bench.py --cdc --steps 3 --no-cpu-baseline --verify sample 2>&1 | grep "^{" | tee gpurun_out/r2_bench_cdc.json | cut -c1-900
echo "== phases"; SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_prof.so CHUNKS=1024 ONLY=lz4 timeout 90 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_lz4s_phases.txt
echo "== host path (512 chunks = 4 GiB)"; timeout 150 python scripts/host_path_bench.py --chunks 512 --max-batch 64 --skip-pageable 2> gpurun_out/r2_hostpath.err | tee gpurun_out/r2_host_path.json; tail -2 gpurun_out/r2_hostpath.err
echo "== e2e steady, hip, 2 workers"; timeout 120 python scripts/e2e_steady.py --chunks 1024 --connections 32 --workers 2 --max-batch 64 2> gpurun_out/r2_e2e.err | tail -1 | tee gpurun_out/r2_e2e_steady.json | cut -c1-500
