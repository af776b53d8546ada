import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the C-ABI library and the oracle when they are missing or were built from other sources (a content hash of the
    sources sits next to each library, so a stale library is never tested and an up-to-date one costs nothing; hipcc
    cross-compiles in the CPU container, the GPU box uses the snapshot's prebuilt files)."""
    from blackman_harris_win_amd import _build
    _build.build_library()
    if _build.oracle_stale():
        _build.build_oracle()


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
        return json.load(f)["entries"]


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
