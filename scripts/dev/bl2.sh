# like bl.sh but with the full verification (what the default line's secondary runs do)
tag=$1; shift
python bench.py --warmup 1 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', j['value'], 'GiB/s', j['ms_per_step'], 'ms/step', j['kernels_ms_per_step'], 'verified', j['verified']['frames_vs_liblz4'])"
