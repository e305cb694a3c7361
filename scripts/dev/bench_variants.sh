#!/bin/bash
# scripts/dev/bench_variants.sh NAME...  -- bench.py (5 steps, no CPU legs) once per scripts/dev/libskyhip_NAME.so ("ship" = the shipping library)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in "$@"; do
  if [ "$v" = ship ]; then unset SKYHIP_LIB_PATH; else export SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_$v.so; fi
  timeout 100 python bench.py --steps ${STEPS:-5} --warmup 1 --no-cpu-baseline --verify none ${BENCH_ARGS:-} 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('$v', j['value'], 'GiB/s', j['ms_per_step'], 'ms/step', j['kernels_ms_per_step'], 'lz4 launch', j['roofline']['avg_launch_ms'])"
done
