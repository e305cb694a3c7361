"""Out-of-bounds hunt: the shipping kernels run under the CPU emulator on buffers fenced by PROT_NONE pages.

The GPU only faults when an over-read leaves an allocation, i.e. at the very end of a multi-GiB buffer (that is how a
16-byte over-read in the LZ4 match finder first showed, 110 s into a 64 GiB run).  Here every chunk is the last one.
The cases live in tests/_guard_cases.py and run in a child interpreter so that SIGSEGV is a failed assertion."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.parametrize("suite", ["lz4", "cdc", "lz4d"])
def test_kernels_stay_inside_their_buffers(suite):
    from tests.emu import emulib
    emulib.lib()   # build once in the parent
    p = subprocess.run([sys.executable, str(ROOT / "tests" / "_guard_cases.py"), suite], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, f"rc={p.returncode} (-11 = SIGSEGV: a kernel touched memory outside its buffer)\n{p.stdout[-2000:]}\n{p.stderr[-4000:]}"
    assert f"OK {suite}" in p.stdout
