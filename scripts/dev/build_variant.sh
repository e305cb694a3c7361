#!/bin/bash
# scripts/dev/build_variant.sh NAME [hipcc flags...]  ->  scripts/dev/libskyhip_NAME.so (git-ignored; select with SKYHIP_LIB_PATH)
set -e
name=$1; shift
cd /root/repo/skyplane_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -DSKY_WITH_CDC "$@" -shared -o ../../scripts/dev/libskyhip_$name.so skyhip.hip 2>&1 | grep -E "error" -A3 | head -20 || true
ls -la ../../scripts/dev/libskyhip_$name.so
