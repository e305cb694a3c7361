#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_cdc1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_cdc1 -o r1 -- env ABLS=0 CHUNKS=512 python /root/repo/scripts/ablate.py > /root/repo/gpurun_out/prof_cdc1.log 2>&1
grep -E "cdc alone|md5 alone|ablate=" /root/repo/gpurun_out/prof_cdc1.log
grep -E "sky_" /root/repo/gpurun_out/prof_cdc1/r1_kernel_stats.csv
rm -f /root/repo/gpurun_out/prof_cdc1/r1_kernel_trace.csv
