import os
import sys

import pytest

try:                     # PyTorch-ROCm ships its own HIP / HSA runtime libraries; a process that uses both torch and
    import torch         # libflexlight_hip.so (the tests that hand torch device pointers to the library) must load torch's
except ImportError:      # first — the other way round torch finds "No HIP GPUs" (Build).  Test infrastructure only.
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    config.addinivalue_line("markers", "experiments: walk schedulers that are not in the shipped library; run against `make EXPERIMENTS=1`'s "
                            "libflexlight_hip_experiments.so (FLX_LIB=...), skipped otherwise")


def pytest_collection_modifyitems(config, items):
    try:
        from flexlight_hip import capi
        have = capi.has_experiments()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="the loaded library has no experimental walk schedulers (make -C web-ray-tracer_amd/csrc EXPERIMENTS=1; FLX_LIB=.../libflexlight_hip_experiments.so)")
    for item in items:
        if "experiments" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): built on demand with gcc."""
    import flx_oracle
    flx_oracle.build()
    flx_oracle.lib()
    return flx_oracle


@pytest.fixture(scope="session")
def hip():
    """One flx_context on GPU 0 through the C ABI.  No fallback: a missing library or GPU is an error."""
    from flexlight_hip import capi
    ctx = capi.Context(0)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def scenes():
    from flexlight_hip.scene_io import Scene
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Scene.golden(name)
        return cache[name]
    return get
