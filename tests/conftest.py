import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full-size parity run (tens of seconds of CPU oracle time); part of -m gpu")


def pytest_collection_modifyitems(config, items):
    """GPU-marked tests are skipped automatically when no device is visible, so a bare
    ``pytest tests`` also works in the CPU-only build container."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def seeded_states():
    from voicefixer_amd import weights
    return weights.seeded_vocoder_state(1234), weights.seeded_restorer_state(4321)
