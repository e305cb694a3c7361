#!/bin/bash
# the round's closing GPU call: full -m gpu suite, PMC traffic of the compressor on the configs[2] stream, the default bench line (+ secondary), its kernel stats
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$1; mkdir -p $OUT; cd $R
timeout 400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
B="--steps 1 --warmup 0 --depth 1 --no-cpu-baseline --verify none --no-secondary"
PMC_T=120 $R/scripts/pmc.sh $OUT cdc "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum" -- python $R/bench.py $B --cdc
python $R/scripts/pmc_traffic.py $OUT cdc 8192 sky_lz4s_frames; cp $R/profiles/traffic.json $OUT/traffic.json
timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; tail -2 $OUT/bench_default.err
(cd /tmp && export TMPDIR=/tmp && timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/prof_default -o bd -- python $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-secondary --verify none > $OUT/bench_default_under_rocprof.json 2>/dev/null); echo "rocprof rc=$?"
