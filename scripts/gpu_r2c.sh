#!/bin/bash
# round 2, third GPU call: records in registers + L2 prefetch; phase profile; EXT variants; LDS counters; bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
make -s -C tests/model; make -s -C tests/emu
echo "== parity"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
echo "== shipping build"; CHUNKS=2048 timeout 300 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids
echo "== EXT=64 build (frames differ from the model by design: timing only)"; SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_ext64.so CHUNKS=2048 timeout 300 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids | head -3
echo "== phase profile (prof build)"; SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_prof.so CHUNKS=512 timeout 300 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids | head -16
echo "== phase profile EXT=64"; SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_ext64prof.so CHUNKS=512 timeout 300 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids | head -16
echo "== phase profile, mixed"; STREAM=mixed SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_prof.so CHUNKS=512 timeout 300 python scripts/dev/lz4s_exp.py 2>&1 | grep -v amdgpu.ids | head -16
echo "== PMC (LZ4 only, 512 chunks)"
cd /tmp && export TMPDIR=/tmp
OUT=$OLDPWD/gpurun_out/pmc_r2c; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  ONLY=lz4 CHUNKS=512 timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT -o p$i -- python $OLDPWD/scripts/dev/lz4s_exp.py > $OUT/p$i.log 2>&1
  f=$OUT/p${i}_counter_collection.csv
  if [ -f $f ]; then python3 - $f <<'PY'
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r.get('Kernel_Name','?')[:24]; agg[k][r['Counter_Name']]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])]+=1
for k,v in agg.items():
    if 'sky_lz4' in k: print(k, {a:(int(b), n[(k,a)]) for a,b in v.items()})
PY
  else echo "no csv for group $i"; tail -3 $OUT/p$i.log; fi
done
cd $OLDPWD
find gpurun_out/pmc_r2c -name "*kernel_trace.csv" -delete
echo "== bench default"
timeout 900 python bench.py --steps 5 --warmup 1 2>&1 | grep "^{" | tee gpurun_out/bench_r2c.json | cut -c1-2000
