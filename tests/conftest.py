import os
import sys

import pytest

# tests force kernel variants / tile plans through the library's development switches (SR_GEMM_TILE, SR_DENSE_VARIANT ...),
# which libsr_hip.so only honours when SR_DEV_SWITCHES=1 (csrc/common.h: sr_dev_getenv)
os.environ["SR_DEV_SWITCHES"] = "1"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "fp32_regime: GPU test of the encoder's no-autocast regime (the reference's dense-query "
                                       "regime, eval_dense.py:94-106)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
