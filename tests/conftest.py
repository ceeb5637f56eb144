import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


def pytest_sessionstart(session):
    """A fresh checkout has no binaries (they are git-ignored): build the HIP extension and the oracle once.
    hipcc cross-compiles gfx950 without a GPU; this is still the HIP path, not a fallback."""
    from deepcomp_amd import build as hip_build
    if not os.environ.get('DCOMP_LIB') and not hip_build.up_to_date():       # DCOMP_LIB: an explicitly named (A/B) library is under test
        hip_build.build()
    from oracle import oracle as orc
    orc.build()
