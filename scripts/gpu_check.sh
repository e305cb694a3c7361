#!/bin/bash
# One gpurun call: GPU parity tests, smoke, a short bench and a rocprofv3 kernel trace.  Everything lands in gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
CHUNKS=${CHUNKS:-1024}
echo "== rocminfo" ; /opt/rocm/bin/rocminfo 2>/dev/null | grep -E 'Marketing Name|gfx|Compute Unit' | head -6 ; nproc
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench"
timeout 900 python bench.py --steps ${STEPS:-3} --warmup 1 --chunks $CHUNKS 2>&1 | tail -5 | tee gpurun_out/bench.log
echo "== rocprofv3 kernel trace"
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof -o r1 -- python /root/repo/bench.py --steps 2 --warmup 1 --chunks 512 --no-cpu-baseline > /root/repo/gpurun_out/prof_bench.log 2>&1 )
tail -2 gpurun_out/prof_bench.log
find gpurun_out/prof -name "*stats*" | head; for f in $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); do head -12 $f; done
