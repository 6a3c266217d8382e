import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:   # (tests/merge_fixtures.py: groups shared by a GPU test and a CPU test)
    sys.path.insert(1, HERE)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_collection_modifyitems(config, items):
    # Plain `pytest tests` on a machine without an AMD GPU driver node: gpu-marked tests are skipped, not failed.  Where
    # /dev/kfd exists (any GPU box) nothing is ever skipped: a library that does not load or finds no device there must
    # show up as failures, never as a silent skip.
    if os.path.exists("/dev/kfd") or os.environ.get("LC_REQUIRE_GPU") == "1":
        return
    skip = pytest.mark.skip(reason="no /dev/kfd on this machine (gpu-marked tests run with -m gpu on an MI355X)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
