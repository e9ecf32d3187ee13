"""rt_intersect_static (the allocation-free entry point) against rt_intersect_async, through the C ABI:
two calls in a row, the second with a smaller input, records compared byte for byte."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_static_entry_point_equals_the_allocating_one():
    import torch
    import raytrace_amd as rt

    mesh = rt.Mesh.load_obj(os.path.join(rt.MODELS_DIR, "teapot.obj"), "tolerant")
    bvh = rt.BVH(mesh, "fixed")
    scene = rt.Scene(mesh, bvh, device=0)
    scene.set_param("feedback_period", 2)   # costs recorded and re-sorted by every second call
    try:
        for (W, H), hint in (((640, 360), 640), ((320, 200), 320), ((320, 200), 0)):
            rays = rt.primary_rays(W, H, rt.frame_camera(bvh.root_box, W, H))
            n = len(rays)
            scene.set_param("image_width", hint)
            rays_d = torch.from_numpy(rays).cuda()
            want = torch.zeros((n, 16), dtype=torch.float32, device="cuda")
            want16 = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
            scene.intersect_async(rays_d.data_ptr(), n, want.data_ptr(), want16.data_ptr())
            torch.cuda.synchronize()
            assert int((want[:, 14].view(torch.int32) != -1).sum()) > 0
            for rep in range(5):   # rep 0 runs without an order, the others with the order built behind rep 0, then rep 2
                got = torch.full((n, 16), -7.0, dtype=torch.float32, device="cuda")
                got16 = torch.full((n, 4), -7.0, dtype=torch.float32, device="cuda")
                scene.intersect_static(rays_d.data_ptr(), n, got.data_ptr(), got16.data_ptr())
                torch.cuda.synchronize()
                assert np.array_equal(got.cpu().numpy().view(np.uint32), want.cpu().numpy().view(np.uint32)), (W, H, hint, rep)
                assert np.array_equal(got16.cpu().numpy().view(np.uint32), want16.cpu().numpy().view(np.uint32)), (W, H, hint, rep)
        # the size limit is an error, not a silent truncation
        with pytest.raises(rt.RtError, match="at most"):
            scene.intersect_static(rays_d.data_ptr(), (1 << 24) + 1, want.data_ptr(), None)
    finally:
        scene.close()
