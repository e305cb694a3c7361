#!/bin/bash
# round-end refresh: smoke(), then the default bench line with the final library
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/final_smoke.log
timeout 300 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench rc=$?"; tail -c 1800 gpurun_out/final_bench.json; tail -3 gpurun_out/final_bench.err
