"""pytest configuration: `gpu` marker + shared golden-fixture helpers."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    return dict(np.load(GOLDEN / f"{name}.npz", allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]

    return get


CASES = ["c1", "c2s", "c3s", "c5s"]
# mid-size cases (oracle/make_golden.py `mid`): K * NsK/2 rows exceed one batch per workgroup of the
# wave-split entropy kernel, so its batch loop runs rg >= 2 times
MID_CASES = ["c2f", "c3m", "c5m"]
