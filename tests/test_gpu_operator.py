"""gpu_compress operator end to end on a real GPU: forked worker creates its HIP context after fork, drains the
queue in batches, writes sidecars; the cooperating sender's wire bytes pass the receiver-side checks of
skyplane/gateway/operators/gateway_receiver.py:150-218 (header parse, lz4.frame.decompress, size == raw_data_len).
The scenario runs in a fresh interpreter (tests/_operator_e2e.py explains why)."""
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu


def test_operator_end_to_end():
    script = Path(__file__).resolve().parent / "_operator_e2e.py"
    p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=400)
    assert p.returncode == 0 and "operator e2e ok" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


def test_loopback_sender_receiver_end_to_end():
    """operator -> sender -> loopback TCP -> receiver (GPU decompress) -> files; every destination file verified."""
    script = Path(__file__).resolve().parents[1] / "scripts" / "e2e_loopback.py"
    p = subprocess.run([sys.executable, str(script), "--chunks", "16", "--connections", "2", "--max-batch", "8"], capture_output=True, text=True, timeout=400)
    assert p.returncode == 0 and '"verified": true' in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
