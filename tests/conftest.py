import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "opm-simulators-legacy_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.lib()
    return orc


@pytest.fixture(autouse=True)
def _oracle_single_thread():
    """The oracle's OpenMP thread count is process-global and changes its reduction order (f32 Krylov paths differ): tests that raise
    it for speed (full-size decks) must not leak it into the next test."""
    yield
    mod = sys.modules.get("oracle.oracle")
    if mod is not None and getattr(mod, "_lib", None) is not None:
        mod.set_threads(1)


@pytest.fixture(scope="session")
def gpu_lib():
    """libopmgpu.so through the C ABI; no fallback -- a missing library is an error."""
    from opmgpu import capi
    lib = capi.load()
    if lib.opmgpu_device_count() < 1:
        pytest.fail("no HIP device visible: GPU tests must run on the GPU box")
    return lib
