import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure; built on demand)."""
    from oracle import oracle as O

    O.lib()
    return O


@pytest.fixture(scope="session")
def hiplib():
    """The product library; building is __graft_entry__.build()'s job, but make sure it is there."""
    from conjugategradient_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import subprocess

        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "conjugategradient_amd", "csrc")])
    return _lib.lib()


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


class _MgcgEnv:
    """monkeypatch for the library's MGCG_* variables: the library reads its environment once (csrc/runtime.hip: tuning()),
    so every change is followed by MgcgReloadEnvironment()."""

    def __init__(self, monkeypatch):
        self._mp = monkeypatch

    @staticmethod
    def _reload():
        from conjugategradient_amd import _lib

        _lib.lib().MgcgReloadEnvironment()

    def setenv(self, name, value):
        self._mp.setenv(name, value)
        self._reload()

    def delenv(self, name, raising=False):
        self._mp.delenv(name, raising=raising)
        self._reload()


@pytest.fixture
def mgcg_env(monkeypatch):
    env = _MgcgEnv(monkeypatch)
    yield env
    monkeypatch.undo()
    env._reload()
