#!/bin/bash
# PMC passes, one rocprofv3 run per counter group (--kernel-trace only: gpurun refuses --pmc combined with other trace domains).
#   scripts/pmc.sh OUTDIR PREFIX "COUNTERS A" "COUNTERS B" ... -- command...
# writes OUTDIR/PREFIX_<first counter>_counter_collection.csv per group and prints per-kernel sums / launch counts.
OUT=$(realpath -m "$1"); PFX=$2; shift 2
groups=(); while [ "$1" != "--" ]; do groups+=("$1"); shift; done; shift
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
for grp in "${groups[@]}"; do
  tag=${PFX}_${grp%% *}
  timeout ${PMC_T:-300} rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT" -o "$tag" -- "$@" > "$OUT/$tag.log" 2>&1
  f="$OUT/${tag}_counter_collection.csv"
  if [ -f "$f" ]; then python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.Counter())
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get('Kernel_Name', '?').split('(')[0][:28]; agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
for k, v in agg.items():
    if k.startswith('sky_'): print(k, {a: (round(b / n[k][a]), n[k][a]) for a, b in v.items()}, "(mean per launch, launches)")
PY
  else echo "no csv for $grp"; tail -3 "$OUT/$tag.log"; fi
done
