#!/bin/bash
# scripts/dev/time_decode_variants.sh NAME...  -- decoder rate on this library's frames (scripts/dev/linked_decode.py) once per scripts/dev/libskyhip_NAME.so ("ship" = the shipping library)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in "$@"; do
  if [ "$v" = ship ]; then unset SKYHIP_LIB_PATH; else export SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_$v.so; fi
  echo "== $v"; ONLY_OURS=${ONLY_OURS-1} SIZES=${SIZES:-1024,32} timeout 120 python scripts/dev/linked_decode.py 2>/dev/null | grep frames:
done
