"""The re-flattened records built by the GPU kernels (RT_SCENE_FLATTEN_DEVICE) against the host flattener's, byte for
byte, on the six meshes the benchmark configurations name; and the device-built scene traces the same hits."""
import os

import numpy as np
import pytest

from conftest import MODELS, bits, framed_primary

pytestmark = pytest.mark.gpu

CASES = [("cube.obj", "tolerant", "fixed"), ("cube_mirror_test.obj", "reference", "fixed"), ("teapot.obj", "tolerant", "fixed"),
         ("bunny.obj", "tolerant", "fixed"), ("bunny_big.obj", "tolerant", "fixed"), ("dragon.obj", "tolerant", "fixed"),
         ("dragon.obj", "tolerant", "sah"), ("bunny.obj", "reference", "reference")]


@pytest.mark.parametrize("model,loader,build", CASES)
def test_device_built_records_equal_host_built(rt, gpu, model, loader, build):
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, model), loader)
    bvh = rt.BVH(mesh, build)
    host = rt.Scene(mesh, bvh, device=-1)
    dev = rt.Scene(mesh, bvh, device=gpu, width=4, flatten_on_device=True)
    try:
        assert dev.info() == {**host.info()}
        for which, name in enumerate(("two-wide float", "two-wide quantized", "four-wide float", "four-wide quantized", "triangles")):
            a, b = host.records(which), dev.records(which)
            assert len(a) == len(b) and len(a) > 0, name
            assert a.tobytes() == b.tobytes(), f"{model}: {name} records differ"
        # and the scene built that way traces what the host-built one traces
        if model != "bunny.obj" or build != "reference":
            rays = framed_primary(rt, bvh, 160, 120)
            ref = rt.Scene(mesh, bvh, device=gpu, width=4)
            assert np.array_equal(bits(dev.intersect(rays)), bits(ref.intersect(rays)))
            ref.close()
    finally:
        host.close()
        dev.close()
