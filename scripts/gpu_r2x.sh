#!/bin/bash
# round 2, closing GPU call: scan helpers no longer pin lane constants across the block (24 -> 13 spilled VGPRs): canary, parity, timing, memory-side
# counters on both streams, default bench + rocprofv3 kernel stats
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
make -s -C tests/model 2>/dev/null; make -s -C tests/emu 2>/dev/null
echo "== smoke (canary)"; timeout 90 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | grep -q "smoke ok" || { echo "canary failed: bad box or bad build, stopping"; exit 1; }
echo "smoke ok"
echo "== pytest -m gpu"; timeout 200 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
echo "== timing"; CHUNKS=2048 timeout 120 python scripts/dev/lz4s_exp.py 2>&1 | grep -v "amdgpu.ids\|^md5"
for st in silesia mixed; do for ctr in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && export TMPDIR=/tmp && STREAM=$st ONLY=lz4 CHUNKS=2048 timeout 80 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OLDPWD/gpurun_out/pmc_r2x -o ${st}_${ctr} -- python $OLDPWD/scripts/dev/lz4s_exp.py > $OLDPWD/gpurun_out/pmc_r2x_${st}_${ctr}.log 2>&1 )
done; python scripts/pmc_traffic.py gpurun_out/pmc_r2x $st 2048 sky_lz4s_compress | cut -c1-330; done
find gpurun_out/pmc_r2x -name "*kernel_trace.csv" -delete
cp profiles/traffic.json gpurun_out/r2_traffic.json
echo "== bench default"; timeout 240 python bench.py 2>&1 | grep "^{" | tee gpurun_out/r2_bench_default.json | cut -c1-1500
echo "== bench default under rocprofv3 --kernel-trace --stats"
( cd /tmp && export TMPDIR=/tmp && timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_default -o def -- python $OLDPWD/bench.py --no-cpu-baseline --verify sample > $OLDPWD/gpurun_out/bench_default_prof.log 2>&1 )
grep "^{" gpurun_out/bench_default_prof.log | tee gpurun_out/r2_bench_default_under_rocprof.json | cut -c1-300; head -5 gpurun_out/prof_default/def_kernel_stats.csv
find gpurun_out/prof_default -name "*kernel_trace.csv" -delete
