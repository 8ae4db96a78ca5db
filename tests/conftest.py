import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _build_once():
    so = os.path.join(ROOT, "goal-conditioned-rl-framework_amd", "libgcrl_hip.so")
    if not os.path.exists(so):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def gcrl():
    _build_once()
    import gcrl_amd
    return gcrl_amd


@pytest.fixture(scope="session")
def lib(gcrl):
    return gcrl._ffi.lib


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def hparams_from_golden(g):
    """SimpleNamespace with the reference config's field names, typed back from the fixture."""
    from types import SimpleNamespace
    out = {}
    for k, v in zip(g["hparams_keys"], g["hparams_vals"]):
        v = str(v)
        try:
            out[str(k)] = int(v)
        except ValueError:
            try:
                out[str(k)] = float(v)
            except ValueError:
                out[str(k)] = None if v == "None" else v
    return SimpleNamespace(**out)
