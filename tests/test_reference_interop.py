"""Wire-level drop-in check with the reference's own code on the other end (build container only: needs
/root/reference; skipped on the GPU box).  Scenarios and rationale: tests/_reference_interop.py."""
import subprocess
import sys
from pathlib import Path

import pytest

from oracle import refshim

ROOT = Path(__file__).resolve().parents[1]

pytestmark = pytest.mark.skipif(not refshim.available(), reason="reference tree not present (GPU box)")


def test_operator_is_a_drop_in_for_the_reference_classes():
    """INTEGRATION.md sections 3 and 6, executed: the operator module on the reference's own ChunkStore / GatewayQueue /
    ChunkRequest, then the reference's GatewaySender with the documented edits applied in memory, streaming the
    operator's frames to the reference's GatewayReceiver; then the destination side: the reference's GatewayReceiver with
    section 6b's deferred-decode branch feeding GatewayHipDecompress on reference objects (tests/_reference_dropin.py)."""
    from tests.emu import emulib
    emulib.lib()
    p = subprocess.run([sys.executable, str(ROOT / "tests" / "_reference_dropin.py")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, f"{p.stdout[-3000:]}\n{p.stderr[-3000:]}"
    assert "OK dropin" in p.stdout


@pytest.mark.parametrize("scenario", ["to_reference", "from_reference"])
def test_interop_with_reference_gateway(scenario):
    from tests.emu import emulib
    emulib.lib()
    p = subprocess.run([sys.executable, str(ROOT / "tests" / "_reference_interop.py"), scenario], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, f"{p.stdout[-3000:]}\n{p.stderr[-3000:]}"
    assert f"OK {scenario}" in p.stdout


def _run_daemon_harness(*args, timeout=240):
    """oracle/ref_daemon.py forks two reference daemons (which fork busy-spinning workers): own session, and the
    whole group is killed if it overstays."""
    import json
    import os
    import signal

    p = subprocess.Popen([sys.executable, str(ROOT / "oracle" / "ref_daemon.py"), *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        out, err = p.communicate()
        pytest.fail(f"daemon harness timed out\n{err[-3000:]}")
    assert p.returncode == 0, f"{out[-2000:]}\n{err[-4000:]}"
    return json.loads(out.strip().splitlines()[-1])


def test_reference_daemons_plain_and_with_gpu_compress_operator():
    """Two REAL GatewayDaemons on localhost (Flask API, forked operators, receiver servers: all from /root/reference).
    First the reference's own DAG (BASELINE configs[0] shape, small), then the same with `gpu_compress` registered through
    INTEGRATION.md section 5's branch and section 6's sender edits -- the CPU compressor is booby-trapped in that run."""
    import socket

    for port in (8080, 8081, 8083):          # the reference hard-codes 8080 (sender -> API) and 8081 (API); 8083 is the harness's
        with socket.socket() as probe:
            probe.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                probe.bind(("127.0.0.1", port))
            except OSError:
                pytest.skip(f"port {port} is taken on this machine")
    from tests.emu import emulib
    emulib.lib()
    r = _run_daemon_harness("--chunks", "6", "--chunk-kib", "512", "--connections", "2")
    assert r["verified"] and r["chunks"] == 6 and not r["gpu_op"]
    r = _run_daemon_harness("--chunks", "6", "--chunk-kib", "128", "--connections", "2", "--gpu-op")
    assert r["verified"] and r["gpu_op"] and r["gpu_compress_chunks"] == 6
