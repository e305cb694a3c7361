import json
import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a box without a gfx950 device skips the gpu-marked tests instead of erroring in their fixtures.
    (`-m gpu` on the GPU box is unaffected: there the device exists, and a missing libskyhip.so still fails loudly.)"""
    try:
        import torch

        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU on this box (gpu-marked tests run with -m gpu on an MI355X)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    return json.loads((ROOT / "tests" / "golden" / "golden.json").read_text())


@pytest.fixture(scope="session")
def small_cases():
    from skyplane_amd import synth

    return synth.small_cases()
