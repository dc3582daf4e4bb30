import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


# the gloo / CPU tests of the multi-process and training LOGIC step small models on CPU tensors: test scaffolding the product refuses
# by default (tests/test_config_surface.py::test_cpu_tensors_raise_without_the_scaffolding_switch checks the refusal)
os.environ.setdefault("ONSSEN_CPU_AUTOGRAD", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
