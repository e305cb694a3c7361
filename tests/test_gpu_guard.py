"""GPU guard-band harness: the shipping kernels on device buffers that END on the last mapped byte before an unmapped
address range -- the device-side twin of tests/test_emu_guard.py, which cannot see what only the device build does (DPP / inline-asm paths, the prefetch
touches of the compressor, the decoder's 256-byte window loads).  Buffers come from hipMemAddressReserve + hipMemCreate + hipMemMap
(skyhip_debug_guard_alloc; SKYHIP_GUARD_ALLOC=1 does the same to everything the library allocates for itself).  Each case runs in a child interpreter
under AMD_SERIALIZE_KERNEL=3: an access outside a buffer kills the child with a memory access fault, which is a failing test here."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
pytestmark = pytest.mark.gpu


def _run(case, timeout=600, **env):
    e = dict(os.environ, SKYHIP_GUARD_ALLOC="1", AMD_SERIALIZE_KERNEL="3", HSA_ENABLE_IPC_MODE_LEGACY="0", **{k: str(v) for k, v in env.items()})
    p = subprocess.run([sys.executable, str(ROOT / "tests" / "_gpu_guard_run.py"), case], capture_output=True, text=True, timeout=timeout, env=e, cwd=str(ROOT))
    if os.environ.get("GUARD_LOG_DIR"):      # (a development aid: the whole transcript of every child, e.g. under gpurun_out/)
        tag = case + "".join(f"_{k}{v}" for k, v in env.items())
        Path(os.environ["GUARD_LOG_DIR"]).mkdir(parents=True, exist_ok=True)
        (Path(os.environ["GUARD_LOG_DIR"]) / f"guard_{tag}.log").write_text(f"rc={p.returncode}\n--- stdout\n{p.stdout}\n--- stderr\n{p.stderr}")
    return p


def test_the_fence_works_one_byte_past_and_one_byte_before():
    """The harness proves itself first: the last byte of a guarded buffer reads fine, one byte further kills the process -- so a passing suite below
    means no kernel went even one byte outside."""
    p = _run("probe_in")
    assert p.returncode == 0 and "OK probe_in" in p.stdout, p.stdout[-1000:] + p.stderr[-2000:]
    for case in ("probe_over", "probe_under"):
        p = _run(case)
        assert p.returncode != 0 and "SURVIVED" not in p.stdout, f"{case}: a deliberate 1-byte access outside the buffer went unnoticed\n{p.stdout[-1000:]}\n{p.stderr[-2000:]}"
        assert "about to read" in p.stdout


@pytest.mark.parametrize("case,env", [("lz4", {}), ("lz4", {"SKYHIP_FRAMES_MIN": 1}), ("batch", {}), ("lz4d", {}), ("cdc", {}), ("dedup", {}), ("smoke", {})],
                         ids=["lz4-block-queue", "lz4-frames-in-place", "host-batch", "lz4d", "cdc", "dedup-on-the-wire", "smoke"])
def test_kernels_stay_inside_their_buffers_on_the_gpu(case, env):
    p = _run(case, **env)
    assert p.returncode == 0 and f"OK {case}" in p.stdout, \
        f"rc={p.returncode} (a memory access fault = a kernel touched memory outside its buffer; the last line names the case)\n{p.stdout[-600:]}\n{p.stderr[-3000:]}"


def test_frames_in_place_at_production_shape_inside_the_fences():
    """One sky_lz4s_frames launch taken by the library's OWN rule (>= 2 chunks of 8 MiB per CU), on guarded buffers, every frame
    decoded by liblz4 (the reference's decoder, gateway_receiver.py:195-201) and every digest compared with hashlib."""
    p = _run("frames512", timeout=900)
    assert p.returncode == 0 and "OK frames512" in p.stdout, f"rc={p.returncode}\n{p.stdout[-2000:]}\n{p.stderr[-4000:]}"
