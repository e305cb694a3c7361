#!/bin/bash
# scripts/dev/pc_sample.sh OUTDIR [stochastic|host_trap] -- rocprofv3 PC sampling (beta) of the compressor alone (CHUNKS=512 ONLY=lz4 scripts/dev/lz4s_exp.py):
# where do the wavefronts of sky_lz4s_frames spend their cycles, instruction by instruction (stochastic: and what do they wait for)?
# Bounded: the profiled command gets 150 s.  Writes OUTDIR/pcs_*.csv.
OUT=$(realpath -m "$1"); M=${2:-stochastic}; mkdir -p "$OUT"
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
if [ "$M" = stochastic ]; then U="--pc-sampling-unit cycles --pc-sampling-interval ${PCS_INTERVAL:-1048576}"; else U="--pc-sampling-unit time --pc-sampling-interval ${PCS_INTERVAL:-100}"; fi
CHUNKS=${CHUNKS:-512} ONLY=lz4 timeout 150 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $M $U --kernel-trace --output-format csv -d "$OUT" -o pcs_$M -- python $R/scripts/dev/lz4s_exp.py > "$OUT/pcs_$M.log" 2>&1
echo "rc=$?"; tail -5 "$OUT/pcs_$M.log"; ls -la "$OUT" | head -20
f=$(ls "$OUT"/pcs_${M}_pc_sampling*.csv 2>/dev/null | head -1)
[ -n "$f" ] && { head -3 "$f" | cut -c1-400; wc -l "$f"; }
