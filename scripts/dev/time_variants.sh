#!/bin/bash
# scripts/dev/time_variants.sh NAME...  -- compressor alone (1024 x 8 MiB, scripts/dev/lz4s_exp.py) once per scripts/dev/libskyhip_NAME.so ("ship" = the shipping library)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in "$@"; do
  if [ "$v" = ship ]; then unset SKYHIP_LIB_PATH; else export SKYHIP_LIB_PATH=$PWD/scripts/dev/libskyhip_$v.so; fi
  echo -n "$v: "; ONLY=lz4 timeout 60 python scripts/dev/lz4s_exp.py 2>/dev/null | grep "^lz4"
done
