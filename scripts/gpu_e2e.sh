#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 420 python -m pytest tests -m gpu -x -q > gpurun_out/e2e_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/e2e_pytest.log
df -h /dev/shm | tail -1
timeout 240 python scripts/e2e_loopback.py --chunks 256 --connections 8 --max-batch 64 > gpurun_out/e2e.json 2> gpurun_out/e2e.err; echo "e2e rc=$?"; cat gpurun_out/e2e.json; tail -5 gpurun_out/e2e.err
