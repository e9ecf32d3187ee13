import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _have_gpu():
    try:
        import raytrace_amd as rt
        return rt.lib.rt_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # A `-m gpu` run on a box without a GPU must fail loudly, not skip silently.
    pass


@pytest.fixture(scope="session")
def rt():
    import raytrace_amd
    return raytrace_amd


@pytest.fixture(scope="session")
def oracle():
    import oracle as o
    return o


@pytest.fixture(scope="session")
def gpu(rt):
    n = rt.lib.rt_device_count()
    assert n > 0, "no HIP device visible: " + rt.lib.rt_last_error().decode()
    return 0


MODELS = os.path.join(ROOT, "assets", "models")


def framed_primary(rt, bvh, W, H):
    origin = rt.frame_camera(bvh.root_box, W, H)
    return rt.primary_rays(W, H, origin)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
