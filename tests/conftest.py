import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "path-tracer_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

DATA = os.path.join(ROOT, "data")
GOLDEN = os.path.join(ROOT, "tests", "golden")

# Seed used for every recorded reference run.
PIN_SEED = (0x9E3779B97F4A7C15, 0xD1B54A32D192ED03)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def scene_path(name):
    return os.path.join(DATA, "scenes", name + ".scene")


def weights_path(name):
    return os.path.join(DATA, "weights", name + ".txt")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O
