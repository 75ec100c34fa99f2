import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def built_library():
    """Build (or reuse) libchatts_amd.so once per session; hipcc cross-compiles without a GPU."""
    from chatts_amd import build
    return build.build()


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


@pytest.fixture(autouse=True)
def _collect_gpu_garbage(request):
    """GPU tests: destroy what the test left behind (models, captured graphs, exchange buffers in reference cycles) at the test's
    end, with the device idle - not at whatever allocation of a LATER test happens to trigger the collector."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    if request.node.module.__name__.endswith(("test_gpu_kernels", "test_gpu_sampler")):
        return          # kernel tests hold plain tensors only: nothing a collection at a later moment could hurt (and 600 x gc = a minute)
    import gc
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.synchronize()


@pytest.fixture(autouse=True)
def _chatts_options_follow_the_environment(monkeypatch):
    """The library never reads the environment (tuning options are set through chatts_set_option, chatts_amd/_lib.py); the tests keep
    saying monkeypatch.setenv("CHATTS_<OPTION>", ...): every such change is mirrored into the option table, and every test starts from
    the table the (clean) environment describes."""
    from chatts_amd import _lib
    _lib.sync_env()                      # (no-op until the library has been loaded)
    real_set, real_del = monkeypatch.setenv, monkeypatch.delenv

    def setenv(name, value, *a, **k):
        real_set(name, value, *a, **k)
        if name.startswith("CHATTS_"):
            _lib.sync_env()

    def delenv(name, *a, **k):
        real_del(name, *a, **k)
        if name.startswith("CHATTS_"):
            _lib.sync_env()

    monkeypatch.setattr(monkeypatch, "setenv", setenv, raising=False)
    monkeypatch.setattr(monkeypatch, "delenv", delenv, raising=False)
    yield
