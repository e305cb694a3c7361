#!/bin/bash
# HBM-side and SQ counters of sky_lz4s_frames (separate rocprofv3 --pmc passes, --kernel-trace only): scripts/dev/pmc_frames.sh OUTDIR [sq]
# writes profiles/traffic.json entries for the silesia and mixed streams (scripts/pmc_traffic.py); with "sq" also the SQ counters per launch
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$1; mkdir -p $OUT
for st in silesia mixed; do
  $R/scripts/pmc.sh $OUT $st "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum" -- env CHUNKS=1024 ONLY=lz4 STREAM=$st python $R/scripts/dev/lz4s_exp.py
  python $R/scripts/pmc_traffic.py $OUT $st 1024 sky_lz4s_frames
done
# the calibration: the same kernel built without the prefetch touches must show one request per line and FETCH_SIZE = half of the input
if [ -f $R/scripts/dev/libskyhip_nopf.so ]; then
  SKYHIP_LIB_PATH=$R/scripts/dev/libskyhip_nopf.so $R/scripts/pmc.sh $OUT nopf "FETCH_SIZE" "TCC_EA0_RDREQ_sum" -- env CHUNKS=1024 ONLY=lz4 python $R/scripts/dev/lz4s_exp.py
fi
if [ "$2" = sq ]; then
$R/scripts/pmc.sh $OUT sq "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" -- env CHUNKS=1024 ONLY=lz4 python $R/scripts/dev/lz4s_exp.py
$R/scripts/pmc.sh $OUT md5 "FETCH_SIZE" -- env CHUNKS=1024 ONLY=md5 python $R/scripts/dev/lz4s_exp.py
fi
cp $R/profiles/traffic.json $OUT/traffic.json
