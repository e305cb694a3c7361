#!/bin/bash
# skim-mode check: GPU parity suite with the shipping library, then LZ4 kernel time with skim on (0) and off (64)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_decompress.py -x -q > gpurun_out/skim_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/skim_pytest.log
SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_abl.so DEC_ONLY=1 ABLS=0,64,0,64 CHUNKS=512 timeout 200 python scripts/ablate.py > gpurun_out/skim_ablate.log 2>&1; echo "ablate rc=$?"; grep -E "ablate=|decompress" gpurun_out/skim_ablate.log
