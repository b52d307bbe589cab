import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(HERE, "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full-ring CPU oracle runs (tens of seconds)")


@pytest.fixture(scope="session")
def small_params():
    """Reduced ring for fast CPU tests: N = 2^11 (1024 slots), same limb structure as the real context
    (12 Q limbs = 60 + 11x45 bits, 4 P limbs, dnum 3), vector dimension 64."""
    import oracle_lib as O
    return O.Params(log_n=11, depth=11, dim=64)


@pytest.fixture(scope="session")
def small_keys(small_params):
    import oracle_lib as O
    return O.Keys(small_params, 7)


@pytest.fixture(scope="session")
def full_params():
    import oracle_lib as O
    return O.Params()
