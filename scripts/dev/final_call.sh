#!/bin/bash
# the round's closing GPU call: full -m gpu suite, the default bench line (+ secondary), its kernel stats under rocprofv3
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$1; mkdir -p $OUT; cd $R
timeout 400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; tail -2 $OUT/bench_default.err
(cd /tmp && export TMPDIR=/tmp && timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/prof_default -o bd -- python $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-secondary --verify none > $OUT/bench_default_under_rocprof.json 2>/dev/null); echo "rocprof rc=$?"
(cd /tmp && export TMPDIR=/tmp && timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/prof_cdc -o bc -- python $R/bench.py --cdc --steps 6 --warmup 1 --no-cpu-baseline --no-secondary --verify none > $OUT/bench_cdc_under_rocprof.json 2>/dev/null); echo "rocprof cdc rc=$?"
