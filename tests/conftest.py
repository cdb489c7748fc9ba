import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# parity tests compare SHARES bit for bit with the oracle: contexts created by the tests use the documented replayable
# share-randomness stream instead of the ChaCha20 production stream (include/zksaas.h "share randomness")
os.environ.setdefault("ZK_RNG_REPLAY", "1")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
