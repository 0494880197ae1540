import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


class Golden:
    """Reader for tests/golden (reference outputs produced by tests/golden/gen_golden.js)."""

    def __init__(self):
        with open(os.path.join(GOLDEN, "manifest.json")) as f:
            self.manifest = json.load(f)

    def cases(self, kind):
        return [c for c in self.manifest["cases"] if c["kind"] == kind]

    def load(self, name, dtype):
        return np.fromfile(os.path.join(GOLDEN, name), dtype=dtype)

    def signal(self, name):
        return self.load(self.manifest["signals"][name]["file"], np.float32)

    def sample_rate(self, name):
        return self.manifest["signals"][name]["sample_rate"]

    def colormap(self, name):
        return self.load(self.manifest["colormaps"][name], np.float64).reshape(256, 4)


_GOLDEN = None


def golden_store():
    global _GOLDEN
    if _GOLDEN is None:
        _GOLDEN = Golden()
    return _GOLDEN


@pytest.fixture(scope="session")
def golden():
    return golden_store()
