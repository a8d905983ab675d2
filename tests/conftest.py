import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs the upstream reference at /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    import ref_shim
    if ref_shim.available():
        return
    skip = pytest.mark.skip(reason="upstream reference not present on this machine")
    for it in items:
        if "reference" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _library_rereads_its_switches(request):
    """The library reads its developer switches (WG_LAYER_G, WG_G192_SPLITK, ...) from the environment once per process and again
    when asked (include/wgflow.h wg_reload_env).  A test that set one with monkeypatch has had it restored by now: start every GPU test
    from the environment as it stands."""
    if "gpu" in request.keywords:
        import constant_memory_waveglow_amd as pkg
        pkg._lib.lib().wg_reload_env()
    yield
