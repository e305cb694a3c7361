#!/bin/bash
# decoder check: GPU decompress tests, then throughput on device-resident frames (ablate.py prints it last)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_gpu_decompress.py tests/test_gpu_operator.py -x -q > gpurun_out/dec_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/dec_pytest.log
ABLS=0 CHUNKS=512 timeout 200 python scripts/ablate.py > gpurun_out/dec_ablate.log 2>&1; echo "ablate rc=$?"; tail -5 gpurun_out/dec_ablate.log
