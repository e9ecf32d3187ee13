"""The sharded frame as one object per rank (rt_tiled_*): with one rank, and with a one-rank RCCL communicator, the gathered
framebuffer must be the one-GPU frame bit for bit -- before and after a re-deal by measured cost, under a camera that moves,
for squares that divide the image and squares that do not."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _one_gpu_frame(rt, torch, mesh, bvh, rays, cam):
    scene = rt.Scene(mesh, bvh, device=0, width=4)
    d = torch.from_numpy(rays).cuda()
    out = torch.zeros(len(rays), dtype=torch.int32, device="cuda")
    scene.trace_pass_bgra8(d.data_ptr(), None, out.data_ptr(), len(rays), cam, 0, 0)
    torch.cuda.synchronize()
    scene.close()
    return out.cpu().numpy()


@pytest.mark.parametrize("tile,with_comm", [(64, False), (120, True), (100, False)])
def test_tiled_frame_equals_the_one_gpu_frame(rt, gpu, tile, with_comm):
    import torch
    W, H = 480, 360
    mesh = rt.Mesh.load_obj(os.path.join(rt.MODELS_DIR, "bunny.obj"), "tolerant")
    mats = mesh.materials.copy()        # the model ships none: one diffuse grey, so that a pixel depends on where its ray hits
    mats[:, 0:3] = 0.8
    mats[:, 9] = 1.0
    mesh = rt.Mesh.from_arrays(mesh.verts, mesh.faces, mesh.normals, mats)
    bvh = rt.BVH(mesh, "fixed")
    rays = rt.primary_rays(W, H, rt.frame_camera(bvh.root_box, W, H))
    comm = rt.Comm(rt.Comm.unique_id(), 1, 0, 0) if with_comm else None
    scene = rt.Scene(mesh, bvh, device=0, width=4)
    frame = rt.TiledFrame(scene, comm, 0, 1, W, H, tile, rays, 0)
    info = frame.info()
    assert info["rays"] == W * H and info["squares"] == info["squares_of_frame"] == -(-W // tile) * -(-H // tile)
    assert info["image_width"] == (tile if W % tile == 0 else 0)
    fb = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    cams = [np.eye(4, dtype=np.float32), rt.camera_matrix(0.05, -0.1, (0.0, 0.0, 0.0))]
    for k, cam in enumerate(cams):
        want = _one_gpu_frame(rt, torch, mesh, bvh, rays, cam)
        assert np.count_nonzero(want & 0xFFFFFF) > 1000 and len(np.unique(want)) > 50   # the model is in the picture, shaded
        fb.zero_()
        frame.render(cam, fb.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(fb.cpu().numpy(), want), (tile, k, "before the re-deal")
        moved = frame.redeal(cam)
        assert moved == 0                                           # one rank: every square stays where it is
        assert frame.info()["rays"] == W * H
        fb.zero_()
        frame.render(cam, fb.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(fb.cpu().numpy(), want), (tile, k, "after the re-deal")
    frame.close()
    scene.close()
    if comm is not None:
        comm.close()


def test_tiled_frame_rejects_bad_arguments(rt, gpu):
    W, H = 64, 64
    mesh = rt.Mesh.load_obj(os.path.join(rt.MODELS_DIR, "cube.obj"), "tolerant")
    bvh = rt.BVH(mesh, "fixed")
    rays = rt.primary_rays(W, H, rt.frame_camera(bvh.root_box, W, H))
    scene = rt.Scene(mesh, bvh, device=0, width=4)
    with pytest.raises(rt.RtError):
        rt.TiledFrame(scene, None, 0, 2, W, H, 32, rays, 0)          # two ranks need a communicator
    with pytest.raises(rt.RtError):
        rt.TiledFrame(scene, None, 1, 1, W, H, 32, rays, 0)          # rank out of range
    frame = rt.TiledFrame(scene, None, 0, 1, W, H, 32, rays, 0)
    with pytest.raises(rt.RtError):
        frame.render(np.eye(4, dtype=np.float32), None, 0)           # the gathering rank needs a framebuffer
    frame.close()
    scene.close()


def test_tiled_frame_of_a_rank_that_does_not_gather(rt, gpu):
    """Rank 1 of 2 holds its own squares' rays and no framebuffer tables; its re-deal goes through the communicator's sum
    (here a one-rank communicator, so only its own squares carry a cost) and leaves a consistent object.  Nothing is
    rendered: the exchange needs the other rank."""
    W, H, tile = 328, 200, 64
    mesh = rt.Mesh.load_obj(os.path.join(rt.MODELS_DIR, "bunny.obj"), "tolerant")
    bvh = rt.BVH(mesh, "fixed")
    rays = rt.primary_rays(W, H, rt.frame_camera(bvh.root_box, W, H))
    comm = rt.Comm(rt.Comm.unique_id(), 1, 0, 0)
    scene = rt.Scene(mesh, bvh, device=0, width=4)
    frame = rt.TiledFrame(scene, comm, 1, 2, W, H, tile, rays, 0, 0.05)
    plan = rt.ShardPlan(W, H, tile, 2)
    info = frame.info()
    assert info["rays"] == plan.records(1) and info["squares"] == len(plan.tiles[1]) and info["squares_of_frame"] == len(plan.tile_order)
    assert info["image_width"] == 0                        # 328 is not a multiple of 64: edge squares are narrower
    moved = frame.redeal(np.eye(4, dtype=np.float32))
    after = frame.info()
    assert 0 <= moved <= info["squares_of_frame"] and after["squares_of_frame"] == info["squares_of_frame"]
    assert 0 <= after["rays"] <= W * H
    frame.close()
    scene.close()
    comm.close()
