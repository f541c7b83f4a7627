import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def vq_sd():
    """Regenerated listener+speaker VQ-VAE weights (same seed as tests/golden/make_golden.py)."""
    import dimx  # noqa: F401
    from dimx import weights
    return weights.synth_state_dict(weights.vq_spec(prefix="listener_vq.") +
                                    weights.vq_spec(prefix="speaker_vq."), 20260928)


@pytest.fixture(scope="session")
def full_sd():
    """Regenerated full SLMFT state dict."""
    import dimx  # noqa: F401
    from dimx import weights
    return weights.synth_state_dict(weights.slmft_spec(), 20260928)
