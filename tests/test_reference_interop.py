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
