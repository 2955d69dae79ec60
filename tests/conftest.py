import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# tests/exp/ holds the checks of code that exists only in the EXPERIMENTS build (csrc/build.sh exp -> libgtav_amd_exp.so: the LayerNorm fold, the persistent
# 256-token-tile kernel — measured slower, kept for tools/).  One process never mixes the two libraries, so they run in a session of their own:
#   GTAV_TEST_EXP=1 python -m pytest tests/exp -q        (GPU box)
# Every other session ignores the directory and loads the product library.
EXP_SESSION = os.environ.get("GTAV_TEST_EXP") == "1"
collect_ignore = [] if EXP_SESSION else ["exp"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full-size CPU oracle cases (tens of seconds)")
    config.addinivalue_line("markers", "exp: experiments-build checks (GTAV_TEST_EXP=1 python -m pytest tests/exp)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The HIP library is built in-tree (hipcc cross-compiles gfx950 without a GPU); tests never run on a fallback."""
    from gtav_amd import lib as L
    if EXP_SESSION:
        L.load_experiments()
        return
    L.build()          # mtime-aware (lib.build): a stale in-tree .so — they are git-ignored but travel to the GPU box — is rebuilt, a current one is kept
    L.load()


@pytest.fixture(scope="session")
def lib():
    from gtav_amd import lib as L
    L.load()
    return L
