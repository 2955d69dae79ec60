import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full-size CPU oracle cases (tens of seconds)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The HIP library is built in-tree (hipcc cross-compiles gfx950 without a GPU); tests never run on a fallback."""
    from gtav_amd import lib as L
    L.build()          # mtime-aware (lib.build): a stale in-tree .so — they are git-ignored but travel to the GPU box — is rebuilt, a current one is kept
    L.load()


@pytest.fixture(scope="session")
def lib():
    from gtav_amd import lib as L
    L.load()
    return L
