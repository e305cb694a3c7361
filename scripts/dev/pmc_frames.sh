#!/bin/bash
# HBM-side counters of sky_lz4s_frames ON THE BENCH'S OWN STREAMS (separate rocprofv3 --pmc passes, --kernel-trace only): scripts/dev/pmc_frames.sh OUTDIR [sq]
#   one launch = one step of `python bench.py` (8192 chunks, Silesia-like) / `--stream mixed --chunks 16384`; writes profiles/traffic.json through
#   scripts/pmc_traffic.py and copies it to OUTDIR (gpurun only brings gpurun_out/ back).  With "sq": the SQ counters over scripts/dev/lz4s_exp.py as well.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$1; mkdir -p $OUT
B="--steps 1 --warmup 0 --depth 1 --no-cpu-baseline --verify none"
$R/scripts/pmc.sh $OUT silesia "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum" -- python $R/bench.py $B
python $R/scripts/pmc_traffic.py $OUT silesia 8192 sky_lz4s_frames
$R/scripts/pmc.sh $OUT mixed "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum" -- python $R/bench.py $B --stream mixed --chunks 16384
python $R/scripts/pmc_traffic.py $OUT mixed 16384 sky_lz4s_frames
$R/scripts/pmc.sh $OUT cdc "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum" -- python $R/bench.py $B --cdc
python $R/scripts/pmc_traffic.py $OUT cdc 8192 sky_lz4s_frames
# the calibration: the same kernel built without the prefetch touches must show one request per line and FETCH_SIZE = half of the input
if [ -f $R/scripts/dev/libskyhip_nopf.so ]; then
  SKYHIP_LIB_PATH=$R/scripts/dev/libskyhip_nopf.so $R/scripts/pmc.sh $OUT nopf "FETCH_SIZE" "TCC_EA0_RDREQ_sum" -- env CHUNKS=1024 ONLY=lz4 python $R/scripts/dev/lz4s_exp.py
fi
if [ "$2" = sq ]; then
$R/scripts/pmc.sh $OUT sq "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" -- env CHUNKS=1024 ONLY=lz4 python $R/scripts/dev/lz4s_exp.py
$R/scripts/pmc.sh $OUT md5 "FETCH_SIZE" -- env CHUNKS=1024 ONLY=md5 python $R/scripts/dev/lz4s_exp.py
fi
cp $R/profiles/traffic.json $OUT/traffic.json
