#!/bin/bash
# round 2, last GPU call: canary, whole GPU suite on the committed tree, memory-side counters of the compressor: mixed stream (refresh), and on the
# Silesia-like stream without the prefetch touches / without register spills (128-VGPR build) to attribute the fetch above the input
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
make -s -C tests/model 2>/dev/null; make -s -C tests/emu 2>/dev/null
echo "== smoke (canary)"; timeout 90 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | grep -q "smoke ok" || { echo "canary failed: bad box or bad build, stopping"; exit 1; }
echo "smoke ok"
echo "== pytest -m gpu"; timeout 200 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
pass() {  # name stream counter libpath
  ( cd /tmp && export TMPDIR=/tmp && SKYHIP_LIB_PATH=$4 STREAM=$2 ONLY=lz4 CHUNKS=2048 timeout 80 rocprofv3 --pmc $3 --kernel-trace --output-format csv -d $OLDPWD/gpurun_out/pmc_r2y -o $1_$3 -- python $OLDPWD/scripts/dev/lz4s_exp.py > $OLDPWD/gpurun_out/pmc_r2y_$1_$3.log 2>&1 )
  grep "^lz4" gpurun_out/pmc_r2y_$1_$3.log | cut -c1-70
  python3 - gpurun_out/pmc_r2y/$1_$3_counter_collection.csv $1 $3 <<'PY'
import csv, sys
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("sky_lz4s") and r["Counter_Name"] == sys.argv[3]]
print(sys.argv[2], sys.argv[3], "per input byte (raw): %.4f over %d launches" % (sum(v) / len(v) * 1024 / (2048 * 8 * 2**20), len(v)))
PY
}
SHIP=$PWD/skyplane_amd/csrc/libskyhip.so
pass mixed mixed FETCH_SIZE $SHIP; pass mixed mixed WRITE_SIZE $SHIP
python scripts/pmc_traffic.py gpurun_out/pmc_r2y mixed 2048 sky_lz4s_compress
pass nopf silesia FETCH_SIZE $PWD/scripts/dev/libskyhip_nopf.so
pass v128 silesia FETCH_SIZE $PWD/scripts/dev/libskyhip_v128.so; pass v128 silesia WRITE_SIZE $PWD/scripts/dev/libskyhip_v128.so
find gpurun_out/pmc_r2y -name "*kernel_trace.csv" -delete
cp profiles/traffic.json gpurun_out/r2_traffic.json
