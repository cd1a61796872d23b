import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)

os.environ.setdefault("DEBUG", "False")  # the reference's splatter reads os.environ["DEBUG"] unguarded


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the mirrors keep the reference's old-style torch.nn.utils.weight_norm (checkpoint key compatibility)
    config.addinivalue_line("filterwarnings", "ignore:.*weight_norm.*is deprecated:FutureWarning")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
