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
        # ... and so is the work-unit limit, which an image hint reaches first: 2^24 records declared 4 wide round up to twice
        # as many 8x8 tiles as the tables hold (rejected before any launch, so the buffers are never touched)
        scene.set_param("image_width", 4)
        with pytest.raises(rt.RtError, match="work units exceed"):
            scene.intersect_static(rays_d.data_ptr(), 1 << 24, want.data_ptr(), None)
    finally:
        scene.close()


def test_static_entry_point_allocates_nothing_and_tells_scenes_apart():
    """What the entry point exists for: between rt_scene_create and rt_scene_destroy its calls leave the device's free memory
    where it was.  Two scenes of equal shape (same ray count and image width, different meshes) alternate on the one table set
    of the GPU, on two streams: the tables are keyed by scene as well as shape, so neither follows the other's cost order,
    and every record still equals rt_intersect_async's."""
    import torch
    import raytrace_amd as rt

    W, H = 512, 320
    scenes, rays_d, want = [], [], []
    for model in ("teapot.obj", "bunny.obj"):
        mesh = rt.Mesh.load_obj(os.path.join(rt.MODELS_DIR, model), "tolerant")
        bvh = rt.BVH(mesh, "fixed")
        sc = rt.Scene(mesh, bvh, device=0, width=4)
        sc.set_param("width", 4)
        sc.set_param("image_width", W)
        sc.set_param("feedback_period", 2)
        rays = rt.primary_rays(W, H, rt.frame_camera(bvh.root_box, W, H))
        r = torch.from_numpy(rays).cuda()
        w = torch.zeros((W * H, 16), dtype=torch.float32, device="cuda")
        sc.intersect_async(r.data_ptr(), W * H, w.data_ptr(), None)
        scenes.append(sc); rays_d.append(r); want.append(w)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    got = [torch.full((W * H, 16), -7.0, dtype=torch.float32, device="cuda") for _ in range(2)]
    # first use of a stream: the HIP runtime sets up the queue behind it, including its arena for the kernels' private memory
    # (the traversal stack's spill part); that happens once per stream, on the first launch there, whoever launches
    for k in (0, 1):
        scenes[k].intersect_static(rays_d[k].data_ptr(), W * H, got[k].data_ptr(), None, streams[k].cuda_stream)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    try:
        for rep in range(6):
            for k in (0, 1):
                scenes[k].intersect_static(rays_d[k].data_ptr(), W * H, got[k].data_ptr(), None, streams[k].cuda_stream)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before, "rt_intersect_static changed the device's free memory"
        for k in (0, 1):
            assert torch.equal(got[k].view(torch.int32), want[k].view(torch.int32)), k
        # same, every call checked (the order tables restart whenever the scene changes)
        for rep in range(4):
            for k in (0, 1):
                got[k].fill_(-7.0)
                torch.cuda.synchronize()
                free_now = torch.cuda.mem_get_info()[0]     # (the comparisons below allocate through torch: measured per call)
                scenes[k].intersect_static(rays_d[k].data_ptr(), W * H, got[k].data_ptr(), None, streams[rep & 1].cuda_stream)
                torch.cuda.synchronize()
                assert torch.cuda.mem_get_info()[0] == free_now, (rep, k)
                assert torch.equal(got[k].view(torch.int32), want[k].view(torch.int32)), (rep, k)
    finally:
        for sc in scenes:
            sc.close()
