#!/bin/bash
# round 2: dedup on the wire with the real library -- the GPU test, then the loopback with recipes on the wire
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
make -s -C tests/model 2>/dev/null; make -s -C tests/emu 2>/dev/null
echo "== gpu test"; timeout 120 python -m pytest tests/test_gpu_parity.py -x -q -k "dedup_wire or emit_corner or ragged_lengths" 2>&1 | tail -3
echo "== e2e steady, hip, dedup on the wire"; timeout 150 python scripts/e2e_steady.py --dedup-wire --chunks 384 --connections 16 --workers 2 --max-batch 32 2> gpurun_out/r2_e2e_dedup.err | tail -1 | tee gpurun_out/r2_e2e_steady_dedup_wire.json | cut -c1-500; tail -3 gpurun_out/r2_e2e_dedup.err
