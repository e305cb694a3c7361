#!/bin/bash
# rocprofv3 kernel-trace of the EXACT default bench command; summary goes to gpurun_out/prof_full (copy into profiles/).
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_full
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_full -o r1 -- python /root/repo/bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/prof_full_bench.log 2>&1
tail -1 /root/repo/gpurun_out/prof_full_bench.log | cut -c1-400
head -8 /root/repo/gpurun_out/prof_full/r1_kernel_stats.csv
rm -f /root/repo/gpurun_out/prof_full/r1_kernel_trace.csv   # large; the stats file is the summary
