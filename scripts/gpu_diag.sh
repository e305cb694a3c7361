#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
make -s -C tests/model 2>/dev/null; make -s -C tests/emu 2>/dev/null
echo "== smoke"; timeout 90 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== pytest -m gpu -v"; timeout 240 python -u -m pytest tests -m gpu -x -v -p no:cacheprovider > gpurun_out/diag_pytest.log 2>&1; echo "rc=$?"; grep -E "PASSED|FAILED|ERROR|::" gpurun_out/diag_pytest.log | tail -8
