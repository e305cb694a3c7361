#!/bin/bash
# scripts/dev/time_linked_variants.sh NAME...  -- linked-frame decode (scripts/dev/linked_decode.py) once per scripts/dev/libskyhip_NAME.so ("ship" = the shipping library),
# with per-kernel times from rocprofv3 --kernel-trace when PROF=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in "$@"; do
  if [ "$v" = ship ]; then unset SKYHIP_LIB_PATH; else export SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_$v.so; fi
  echo "== $v"
  if [ -n "${PROF:-}" ]; then
    (cd /tmp && TMPDIR=/tmp ONLY_LINKED=1 SIZES=${SIZES:-32} FRAMES=${FRAMES:-32} timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/${TAG:-lv}/prof_$v" -o ld -- python "$OLDPWD/scripts/dev/linked_decode.py" 2>/dev/null | grep frames:)
    python - "$PWD/gpurun_out/${TAG:-lv}/prof_$v" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
if f:
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        if k.startswith("sky_"): print(f"   {k:24s} n={len(v):3d} mean {sum(v) / len(v):9.1f} us  min {min(v):9.1f}")
PY
  else
    ONLY_LINKED=1 SIZES=${SIZES:-1024,256,128,64,32} timeout 150 python scripts/dev/linked_decode.py 2>/dev/null | grep frames:
  fi
done
