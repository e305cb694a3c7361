# usage: bl.sh TAG args...  -> one line summary
tag=$1; shift
python bench.py --warmup 1 --no-cpu-baseline --no-secondary --verify none "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', j['value'], 'GiB/s', j['ms_per_step'], 'ms/step', j['kernels_ms_per_step'], 'lz4 launch', j['roofline']['avg_launch_ms'])"
