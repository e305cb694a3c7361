#!/bin/bash
# GPU suite + PCIe-inclusive host-path rate.  Everything under `timeout`, logs in gpurun_out/.
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 420 python -m pytest tests -m gpu -x -q > gpurun_out/hostpath_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/hostpath_pytest.log
tail -4 gpurun_out/hostpath_pytest.log
timeout 240 python scripts/host_path_bench.py --chunks 256 --max-batch 64 > gpurun_out/hostpath.json 2> gpurun_out/hostpath.err; echo "hostpath rc=$?"
cat gpurun_out/hostpath.json; tail -3 gpurun_out/hostpath.err
