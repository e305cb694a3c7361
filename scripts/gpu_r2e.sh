#!/bin/bash
# round 2, fifth GPU call: issue-rate probe, linked-frame decode, configs[2]/[3] bench lines with their rocprof stats, steady-state e2e,
# the reference's real daemons (configs[0], configs[4])
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
make -s -C tests/model; make -s -C tests/emu
echo "== parity + error paths"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
echo "== issue rate probe"; timeout 120 scripts/dev/issue_rate 2>&1 | tee gpurun_out/r2_issue_rate.txt
echo "== linked-frame decode"; timeout 600 python scripts/dev/linked_decode.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_linked_decode.txt | head -8
echo "== bench --cdc (configs[2]) under rocprofv3"
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_cdc -o cdc -- python $OLDPWD/bench.py --cdc --steps 3 --warmup 1 --no-cpu-baseline > $OLDPWD/gpurun_out/bench_cdc_prof.log 2>&1 )
grep "^{" gpurun_out/bench_cdc_prof.log | tee gpurun_out/r2_bench_cdc_under_rocprof.json | cut -c1-900; head -12 gpurun_out/prof_cdc/cdc_kernel_stats.csv
find gpurun_out/prof_cdc -name "*kernel_trace.csv" -delete
echo "== bench --cdc"; timeout 900 python bench.py --cdc --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep "^{" | tee gpurun_out/r2_bench_cdc.json | cut -c1-1200
echo "== bench --stream mixed (configs[3] stream, 1 GPU)"; timeout 900 python bench.py --stream mixed --chunks 16384 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep "^{" | tee gpurun_out/r2_bench_mixed.json | cut -c1-1200
echo "== e2e steady (this repo's mirrors), 1024 x 8 MiB, 32 connections, 2 workers"
timeout 600 python scripts/e2e_steady.py --chunks 1024 --connections 32 --workers 2 --max-batch 64 2>&1 | tail -1 | tee gpurun_out/r2_e2e_steady.json | cut -c1-700
echo "== configs[0]: reference daemons, CPU path, 1000 x 8 MiB PRNG chunks, 6 connections"
timeout 520 python oracle/ref_daemon.py --chunks 1000 --chunk-kib 8192 --connections 6 --stream random --timeout 400 --out gpurun_out/r2_config0.json 2>&1 | tail -2 | cut -c1-900
sleep 3
echo "== configs[4]: reference daemons + gpu_compress on the MI355X, 1000 x 8 MiB, 128 connections"
timeout 520 python oracle/ref_daemon.py --chunks 1000 --chunk-kib 8192 --connections 128 --gpu-op --context hip --max-batch 64 --workers 1 --timeout 400 --out gpurun_out/r2_config4_128.json 2>&1 | tail -2 | cut -c1-900
sleep 3
echo "== configs[4] at 8 connections"
timeout 400 python oracle/ref_daemon.py --chunks 400 --chunk-kib 8192 --connections 8 --gpu-op --context hip --max-batch 64 --workers 1 --timeout 300 --out gpurun_out/r2_config4_8.json 2>&1 | tail -2 | cut -c1-900
