#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_cdc
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_cdc -o r1 -- python /root/repo/bench.py --cdc --steps 2 --warmup 1 --chunks 2048 --no-cpu-baseline > /root/repo/gpurun_out/prof_cdc_bench.log 2>&1
tail -1 /root/repo/gpurun_out/prof_cdc_bench.log | cut -c1-300
head -14 /root/repo/gpurun_out/prof_cdc/r1_kernel_stats.csv
rm -f /root/repo/gpurun_out/prof_cdc/r1_kernel_trace.csv
