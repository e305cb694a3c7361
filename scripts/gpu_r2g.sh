#!/bin/bash
# round 2, seventh GPU call: final library -- parity, A/B on both streams, default bench + its rocprofv3 kernel stats, PMC traffic
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
make -s -C tests/model; make -s -C tests/emu
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for st in silesia mixed; do
  echo "== $st: shipping (precheck)"; STREAM=$st CHUNKS=2048 timeout 300 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids
  echo "== $st: no precheck";        STREAM=$st SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_noprecheck.so CHUNKS=2048 ONLY=lz4 timeout 300 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids
done
echo "== phases"; SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_prof.so CHUNKS=1024 ONLY=lz4 timeout 200 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids | head -15 | tee gpurun_out/r2_lz4s_phases.txt
echo "== bench default"; timeout 900 python bench.py 2>&1 | grep "^{" | tee gpurun_out/r2_bench_default.json | cut -c1-1700
echo "== bench default under rocprofv3 --kernel-trace --stats"
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_default -o def -- python $OLDPWD/bench.py --no-cpu-baseline > $OLDPWD/gpurun_out/bench_default_prof.log 2>&1 )
grep "^{" gpurun_out/bench_default_prof.log | tee gpurun_out/r2_bench_default_under_rocprof.json | cut -c1-600; head -8 gpurun_out/prof_default/def_kernel_stats.csv
find gpurun_out/prof_default -name "*kernel_trace.csv" -delete
echo "== PMC traffic (separate passes)"
for st in silesia mixed; do for ctr in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && export TMPDIR=/tmp && STREAM=$st ONLY=lz4 CHUNKS=2048 timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OLDPWD/gpurun_out/pmc_r2g -o ${st}_${ctr} -- python $OLDPWD/scripts/dev/lz4s_exp.py > $OLDPWD/gpurun_out/pmc_r2g_${st}_${ctr}.log 2>&1 )
done; python scripts/pmc_traffic.py gpurun_out/pmc_r2g $st 2048 sky_lz4s_compress; done
find gpurun_out/pmc_r2g -name "*kernel_trace.csv" -delete
cp profiles/traffic.json gpurun_out/r2_traffic.json
