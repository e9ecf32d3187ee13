"""A caller on the legacy NULL stream with a cost-feedback sort behind EVERY launch ("feedback_period" = 1): the order
tables are built on the library's side stream, which is non-blocking and of default priority, so the null stream's implicit
barrier never includes it.  The hits must equal those of the same launches with the sort on the caller's own stream.
Collected late on purpose: it exercises stream ordering rather than arithmetic."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_null_stream_caller_with_a_sort_behind_every_launch():
    import torch
    import raytrace_amd as rt

    mesh = rt.Mesh.load_obj(os.path.join(rt.MODELS_DIR, "teapot.obj"), "tolerant")
    bvh = rt.BVH(mesh, "fixed")
    W, H = 640, 360
    rays = rt.primary_rays(W, H, rt.frame_camera(bvh.root_box, W, H))
    rays_d = torch.from_numpy(rays).cuda()
    out = []
    for side in (1, 0):
        scene = rt.Scene(mesh, bvh, device=0, width=4)
        scene.set_param("image_width", W)
        scene.set_param("feedback_period", 1)
        scene.set_param("side_stream", side)
        hits = torch.zeros((W * H, 16), dtype=torch.float32, device="cuda")
        for _ in range(50):
            scene.intersect_async(rays_d.data_ptr(), W * H, hits.data_ptr(), None, 0)   # stream 0 = the legacy null stream
        torch.cuda.synchronize()
        out.append(hits.cpu().numpy().view(np.uint32))
        scene.close()
    assert np.array_equal(out[0], out[1])
