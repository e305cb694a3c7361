#!/bin/bash
# round 2, first GPU call: parity of the slice-parallel compressor, A/B bench against the wave-per-block kernel, kernel trace
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== rocminfo"; /opt/rocm/bin/rocminfo 2>/dev/null | grep -E 'Marketing Name|gfx|Compute Unit' | head -4; nproc
make -s -C tests/model; make -s -C tests/emu
echo "== pytest -m gpu (parity files first)"
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu_parity.log
echo "== bench lz4s (2048 chunks, 3 steps)"
SKYHIP_DEBUG=1 timeout 600 python bench.py --steps 3 --warmup 1 --chunks 2048 --no-cpu-baseline 2>&1 | tail -4 | tee gpurun_out/bench_lz4s_2048.log
echo "== bench wave kernel (2048 chunks, 3 steps)"
SKYHIP_LZ4_KERNEL=wave timeout 600 python bench.py --steps 3 --warmup 1 --chunks 2048 --no-cpu-baseline 2>&1 | tail -2 | tee gpurun_out/bench_wave_2048.log
echo "== bench lz4s default size"
timeout 900 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -2 | tee gpurun_out/bench_lz4s_full.log
echo "== rocprofv3 kernel trace (lz4s, 2048 chunks)"
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_r2a -o r2a -- python $OLDPWD/bench.py --steps 3 --warmup 1 --chunks 2048 --no-cpu-baseline > $OLDPWD/gpurun_out/prof_r2a_bench.log 2>&1 )
tail -1 gpurun_out/prof_r2a_bench.log | cut -c1-300
for f in $(find gpurun_out/prof_r2a -name "*kernel_stats.csv" | head -1); do head -10 $f; done
find gpurun_out/prof_r2a -name "*kernel_trace.csv" -delete
echo "== remaining gpu tests"
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py 2>&1 | tail -8 | tee gpurun_out/pytest_gpu_rest.log
