#!/bin/bash
# PMC passes over the LZ4 kernel (separate rocprofv3 runs per counter group; --kernel-trace only).
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc; mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -oE '\b(SQ|TA|TCP|TCC|GRBM|TD)_[A-Z0-9_]+' | sort -u > $OUT/counters.txt; wc -l $OUT/counters.txt
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT -o p$i -- env ABLS=0 CHUNKS=${CHUNKS:-256} python /root/repo/scripts/ablate.py > $OUT/p$i.log 2>&1
  f=$OUT/p${i}_counter_collection.csv
  if [ -f $f ]; then python3 - $f <<'PY'
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r.get('Kernel_Name','?')[:24]; agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in agg.items():
    if 'sky_lz4' in k or 'md5' in k: print(k, {a:int(b) for a,b in v.items()})
PY
  else echo "no csv for group $i"; tail -3 $OUT/p$i.log; fi
done
