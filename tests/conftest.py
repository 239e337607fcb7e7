import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# the test YAMLs keep the reference's default `encoder_weights: imagenet`; no pretrained file exists offline, and the model
# constructor refuses a silent random start unless told so (models.SegModel.compile)
os.environ.setdefault("STP_ALLOW_RANDOM_ENCODER", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
