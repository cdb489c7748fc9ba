import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# parity tests compare SHARES bit for bit with the oracle: contexts created by the tests use the documented replayable
# share-randomness stream instead of the ChaCha20 production stream (include/zksaas.h "share randomness").  The option is
# set through the Python mirror's DEFAULT_OPTIONS; worker processes the tests spawn set it themselves (dist_worker.py)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import zksaas_amd.api as _zk_api  # noqa: E402  (pure Python: the library is loaded by the first context)

_zk_api.DEFAULT_OPTIONS["rng_replay"] = 1


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # A GPU test that stops making progress (a library call that never returns cannot be interrupted from Python) ends the
    # run with thread dumps after ten minutes instead of sitting in the driver's limit: the slowest test takes ~60 s.
    try:
        import pytest_timeout  # noqa: F401
    except ImportError:
        return
    for it in items:
        if it.get_closest_marker("gpu") and not it.get_closest_marker("timeout"):
            it.add_marker(pytest.mark.timeout(600, method="thread"))
