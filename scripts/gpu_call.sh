#!/bin/bash
# One gpurun call = a canary + a list of bounded steps.      scripts/gpu_call.sh TAG 'command 1' 'command 2' ...
#   * smoke() runs first under its own 150 s limit; when it fails the call ends at once (a broken box once ate 19 GPU-minutes);
#   * every step runs under `timeout ${STEP_T:-300}` from the repo root, stdout+stderr to gpurun_out/TAG/stepK.log, exit code and tail echoed;
#   * a step of the form  T=600:command  gets its own limit.
# Everything a step wants kept must be written under gpurun_out/ (merged back by gpurun).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
if [ -z "${NO_CANARY:-}" ]; then
  timeout 150 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/canary.log" 2>&1; rc=$?
  echo "== canary rc=$rc ($(( $(date +%s) - t0 )) s)"; tail -2 "$OUT/canary.log"
  [ $rc -ne 0 ] && { echo "canary failed: giving the box back"; exit 9; }
fi
k=0
for step in "$@"; do
  k=$((k+1)); lim=${STEP_T:-300}
  case "$step" in T=*:*) lim=${step%%:*}; lim=${lim#T=}; step=${step#*:};; esac
  s0=$(date +%s)
  timeout "$lim" bash -c "$step" > "$OUT/step$k.log" 2>&1; rc=$?
  echo "== step $k rc=$rc ($(( $(date +%s) - s0 )) s, limit $lim): $step"; tail -${TAIL:-12} "$OUT/step$k.log"
done
echo "== total $(( $(date +%s) - t0 )) s"
