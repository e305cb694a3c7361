#!/bin/bash
# One GPU call: the -m gpu suite, then the mixed-class stream at the default bench size (the run that used to fault).
# Everything under `timeout`, everything logged to gpurun_out/.
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 420 python -m pytest tests -m gpu -x -q > gpurun_out/verify_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/verify_pytest.log
tail -3 gpurun_out/verify_pytest.log
timeout 200 python bench.py --stream mixed --no-cpu-baseline > gpurun_out/verify_mixed.json 2> gpurun_out/verify_mixed.err; echo "mixed rc=$?"
tail -c 1500 gpurun_out/verify_mixed.json; tail -5 gpurun_out/verify_mixed.err
