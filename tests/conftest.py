import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deepsphere-cosmo-tf2_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` through gpurun)")
    # skipped tests are named with their reason in the short summary (the two-GPU test on a one-GPU box), also under -q
    if "s" not in (config.option.reportchars or "") and "a" not in (config.option.reportchars or "").lower():
        config.option.reportchars = (config.option.reportchars or "") + "s"


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
