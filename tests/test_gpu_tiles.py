"""Image-tile sharding through the HIP path on one GPU: G logical ranks each trace their cost-assigned squares
(their rays stacked square after square, declared to the library as an image one square wide), the compact
records land in one gather buffer, rt_scatter_tiles rebuilds the row-major framebuffer on the device, and the
result must equal the single-launch framebuffer byte for byte."""
import os

import numpy as np
import pytest

from conftest import MODELS

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(640, 360, 64), (600, 360, 64), (328, 200, 32)])
def test_sharded_render_equals_single_launch(rt, gpu, shape):
    import torch
    from raytrace_amd.tiles import assign_tiles, balance, rays_of_tiles, tile_costs_from_units, tile_grid
    W, H, tile = shape
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, "bunny.obj"), "tolerant")
    bvh = rt.BVH(mesh, "fixed")
    origin = rt.frame_camera(bvh.root_box, W, H)
    rays = rt.primary_rays(W, H, origin)
    scene = rt.Scene(mesh, bvh, device=gpu, width=4)
    try:
        rays_d = torch.from_numpy(rays).cuda()
        want = torch.zeros((W * H, 4), dtype=torch.float32, device="cuda")
        scene.set_param("image_width", W)
        scene.intersect_async(rays_d.data_ptr(), W * H, None, want.data_ptr())
        torch.cuda.synchronize()
        want_h = want.cpu().numpy()
        assert int((want_h[:, 0].view(np.uint32) != rt.MISS_ID).sum()) > 1000

        units = scene.unit_costs_primary(W, H, origin)
        assert units.shape == ((H + 7) // 8, (W + 7) // 8) and units.max() > units.min()
        # the table of the array form (image hint W) is the same table
        assert np.array_equal(scene.unit_costs(rays_d.data_ptr(), W * H).reshape(units.shape), units)
        costs = tile_costs_from_units(units, W, H, tile)
        tx, ty = tile_grid(W, H, tile)
        for G in (2, 3, 8):
            lists = assign_tiles(tx * ty, G, costs)
            # greedy by measured cost: within 5 % of perfect balance, and never worse than dealing the squares round-robin
            # by more than the greedy rule's own granularity (on a small image round-robin can land within 0.01 % by luck)
            assert balance(lists, costs) <= 1.05
            assert balance(lists, costs) <= balance(assign_tiles(tx * ty, G, None), costs) + 0.01
            gather = torch.full((W * H, 4), -3.0, dtype=torch.float32, device="cuda")
            ids_all, first_all, base = [], [], 0
            # a rank's rays are whole squares, each `tile` wide when the width divides evenly: an image one square wide
            scene.set_param("image_width", tile if W % tile == 0 else 0)
            for r in range(G):
                idx, first = rays_of_tiles(W, H, tile, lists[r])
                mine = torch.from_numpy(rays[idx]).cuda()
                scene.intersect_async(mine.data_ptr(), len(idx), None, gather.data_ptr() + base * 16)
                torch.cuda.synchronize()
                ids_all.append(lists[r].astype(np.uint32))
                first_all.append(first + np.uint32(base))
                base += len(idx)
            assert base == W * H
            ids_d = torch.from_numpy(np.concatenate(ids_all).astype(np.int32)).cuda()
            first_d = torch.from_numpy(np.concatenate(first_all).astype(np.int32)).cuda()
            fb = torch.full((W * H, 4), -5.0, dtype=torch.float32, device="cuda")
            rt.scatter_tiles(gpu, gather.data_ptr(), W * H, ids_d.data_ptr(), first_d.data_ptr(), len(ids_d), tile, W, H, 16, fb.data_ptr())
            torch.cuda.synchronize()
            assert fb.cpu().numpy().tobytes() == want_h.tobytes(), (shape, G)
        # 64-byte records scatter the same way
        full = torch.zeros((W * H, 16), dtype=torch.float32, device="cuda")
        scene.set_param("image_width", W)
        scene.intersect_async(rays_d.data_ptr(), W * H, full.data_ptr(), None)
        lists = assign_tiles(tx * ty, 1, costs)
        idx, first = rays_of_tiles(W, H, tile, lists[0])
        stacked = full[torch.from_numpy(idx).cuda()].contiguous()
        fb64 = torch.zeros_like(full)
        ids_d = torch.from_numpy(lists[0].astype(np.int32)).cuda()       # named: they must outlive the launch
        first_d = torch.from_numpy(first.astype(np.int32)).cuda()
        rt.scatter_tiles(gpu, stacked.data_ptr(), W * H, ids_d.data_ptr(), first_d.data_ptr(), len(lists[0]), tile, W, H, 64, fb64.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(fb64.view(torch.int32), full.view(torch.int32))
        # list entries that name no square, or whose records would run off the source, are skipped, not followed
        bad_ids = torch.tensor([tx * ty, 0x7FFFFFFF, 0], dtype=torch.int32, device="cuda")
        bad_first = torch.tensor([0, 0, W * H - 1], dtype=torch.int32, device="cuda")
        before = fb64.clone()
        rt.scatter_tiles(gpu, stacked.data_ptr(), W * H, bad_ids.data_ptr(), bad_first.data_ptr(), 3, tile, W, H, 64, fb64.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(fb64.view(torch.int32), before.view(torch.int32))
    finally:
        scene.close()


def test_pixels_pack_to_bgra_and_scatter_as_4_byte_records(rt, gpu):
    """The full trace step's RGBA32F pixels packed to 8-bit BGRA (rt_pack_bgra8), stacked square by square and scattered
    back with 4-byte records: the framebuffer a multi-GPU render gathers when it ships pixels instead of hits."""
    import torch
    from raytrace_amd.tiles import assign_tiles, rays_of_tiles, tile_grid
    W, H, tile = 328, 200, 32
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, "cube_mirror_test.obj"), "reference")     # the one model that ships materials
    bvh = rt.BVH(mesh, "fixed")
    rays = rt.primary_rays(W, H, rt.frame_camera(bvh.root_box, W, H))
    scene = rt.Scene(mesh, bvh, device=gpu, width=4)
    try:
        scene.set_param("image_width", W)
        rays_d = torch.from_numpy(rays).cuda()
        rgba = torch.zeros((W * H, 4), dtype=torch.float32, device="cuda")
        scene.trace_pass(rays_d.data_ptr(), None, rgba.data_ptr(), W * H, np.eye(4, dtype=np.float32), 0)
        pixels = torch.zeros((W * H, 1), dtype=torch.int32, device="cuda")
        rt.pack_bgra8(gpu, rgba.data_ptr(), W * H, pixels.data_ptr())
        torch.cuda.synchronize()
        c = rgba.cpu().numpy()
        q = np.minimum(np.float32(255), np.maximum(np.float32(0), c * np.float32(255) + np.float32(0.5))).astype(np.uint32)
        want_px = (q[:, 2] | (q[:, 1] << 8) | (q[:, 0] << 16) | np.uint32(0xFF000000)).astype(np.uint32)
        assert np.array_equal(pixels.cpu().numpy().view(np.uint32).reshape(-1), want_px)
        assert len(np.unique(want_px)) > 2       # background, and faces shaded by their angle to the ray
        tx, ty = tile_grid(W, H, tile)
        for G in (1, 3):
            lists = assign_tiles(tx * ty, G, None)
            order = np.concatenate(lists)
            idx, first = rays_of_tiles(W, H, tile, order)
            stacked = pixels[torch.from_numpy(idx).cuda()].contiguous()
            ids_d = torch.from_numpy(order.astype(np.int32)).cuda()
            first_d = torch.from_numpy(first.astype(np.int32)).cuda()
            fb = torch.zeros_like(pixels)
            rt.scatter_tiles(gpu, stacked.data_ptr(), W * H, ids_d.data_ptr(), first_d.data_ptr(), len(order), tile, W, H, 4, fb.data_ptr())
            torch.cuda.synchronize()
            assert torch.equal(fb, pixels), G
        with pytest.raises(rt.RtError):
            rt.scatter_tiles(gpu, stacked.data_ptr(), W * H, ids_d.data_ptr(), first_d.data_ptr(), len(order), tile, W, H, 6, fb.data_ptr())
    finally:
        scene.close()


def test_gather_scatter_entry_point_with_a_one_rank_communicator_and_logical_ranks(rt, gpu):
    """rt_gather_scatter, the exchange step a C++ host calls.  (1) With a REAL RCCL communicator of one rank (made through
    rt_comm_unique_id / rt_comm_create): the call moves nothing between GPUs, copies the rank's records into the gather
    buffer and scatters.  (2) G logical ranks on one GPU: every rank's records are traced straight into its slice of the
    gather buffer (what the root does for its own), then one call with nranks = 1 over the plan's tables scatters them.
    Both framebuffers must equal the single-launch framebuffer byte for byte, for 16-byte hits and for 8-byte {id, t}
    records packed by rt_pack_hits8."""
    import torch
    W, H, tile = 600, 360, 64
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, "bunny.obj"), "tolerant")
    bvh = rt.BVH(mesh, "fixed")
    origin = rt.frame_camera(bvh.root_box, W, H)
    rays = rt.primary_rays(W, H, origin)
    scene = rt.Scene(mesh, bvh, device=gpu, width=4)
    scene.set_param("width", 4)
    comm = rt.Comm(rt.Comm.unique_id(), 1, 0, gpu)
    try:
        comm.check()     # grouped ncclSend/ncclRecv (with itself) and the host comparison: the run-time RCCL binding works
        rays_d = torch.from_numpy(rays).cuda()
        want = torch.zeros((W * H, 4), dtype=torch.float32, device="cuda")
        scene.set_param("image_width", W)
        scene.intersect_async(rays_d.data_ptr(), W * H, None, want.data_ptr())
        torch.cuda.synchronize()
        want8 = want[:, :2].contiguous()
        units = scene.unit_costs_primary(W, H, origin)
        scene.set_param("image_width", 0)
        for G in (1, 3):
            plan = rt.ShardPlan(W, H, tile, G, units)
            ids, first = plan.scatter_tables()
            ids_d, first_d = torch.from_numpy(ids.astype(np.int32)).cuda(), torch.from_numpy(first.astype(np.int32)).cuda()
            gather = torch.full((W * H, 4), -3.0, dtype=torch.float32, device="cuda")
            for r in range(G):
                mine = torch.from_numpy(rays[plan.ray_indices(r)]).cuda()
                scene.intersect_async(mine.data_ptr(), len(mine), None, gather.data_ptr() + int(plan.first_record[r]) * 16)
                torch.cuda.synchronize()
            fb = torch.full((W * H, 4), -5.0, dtype=torch.float32, device="cuda")
            # the library sees ONE rank that owns every record (they are already in place: mine == its slice of the gather buffer)
            whole = np.array([0, W * H], np.uint64)
            rt.gather_scatter(comm if G == 1 else None, 0, 1, 0, gpu, gather.data_ptr(), gather.data_ptr(), whole, 16, ids_d.data_ptr(),
                              first_d.data_ptr(), len(ids), tile, W, H, fb.data_ptr())
            torch.cuda.synchronize()
            assert torch.equal(fb.view(torch.int32), want.view(torch.int32)), G
            # the same with a separate record buffer (copied into the gather buffer by the call) and the 8-byte payload
            small = torch.empty((W * H, 2), dtype=torch.float32, device="cuda")
            rt.pack_hits8(gpu, gather.data_ptr(), W * H, small.data_ptr())
            gather8 = torch.full((W * H, 2), -3.0, dtype=torch.float32, device="cuda")
            fb8 = torch.full((W * H, 2), -5.0, dtype=torch.float32, device="cuda")
            rt.gather_scatter(comm, 0, 1, 0, gpu, small.data_ptr(), gather8.data_ptr(), whole, 8, ids_d.data_ptr(), first_d.data_ptr(), len(ids),
                              tile, W, H, fb8.data_ptr())
            torch.cuda.synchronize()
            assert torch.equal(fb8.view(torch.int32), want8.view(torch.int32)), G
        # argument errors
        assert rt.lib.rt_gather_scatter(None, 0, 2, 0, gpu, None, None, None, 16, None, None, 0, tile, W, H, None, None) == -1
    finally:
        comm.close()
        scene.close()


def test_device_composite_equals_the_host_blend(rt, gpu):
    """rt_composite_layers: the reference's two-layer display blend (render_client/img_main.cpp:434,512-526) on the device,
    against the same arithmetic in numpy float32, for the blended RGBA32F pixels and their 8-bit BGRA form."""
    import torch
    rng = np.random.default_rng(3)
    n = 100_003
    l0 = rng.uniform(-0.2, 1.3, (n, 4)).astype(np.float32)
    l1 = rng.uniform(-0.2, 1.3, (n, 4)).astype(np.float32)
    a, b = torch.from_numpy(l0).cuda(), torch.from_numpy(l1).cuda()
    out = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    pix = torch.zeros(n, dtype=torch.int32, device="cuda")
    rt.composite_layers(gpu, a.data_ptr(), b.data_ptr(), n, out.data_ptr(), pix.data_ptr())
    torch.cuda.synchronize()
    f32 = np.float32
    want = l1 * l1[:, 3:4] + l0 * (f32(1) - l0[:, 3:4])
    got = out.cpu().numpy()
    assert np.allclose(got, want, rtol=0, atol=2e-7)         # (the device may fuse the multiply-add of the blend)
    q = np.clip(got * f32(255) + f32(0.5), 0, 255).astype(np.uint32)
    want_pix = (q[:, 2] | (q[:, 1] << 8) | (q[:, 0] << 16) | np.uint32(0xFF000000)).astype(np.uint32)
    assert np.array_equal(pix.cpu().numpy().view(np.uint32), want_pix)
    # one layer only: the layer itself
    rt.composite_layers(gpu, a.data_ptr(), None, n, out.data_ptr(), None)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), l0)


def test_compact_records_and_8_bit_pixels_written_by_the_trace_kernel(rt, gpu):
    """rt_intersect_compact writes the 16-, 8- or 4-byte head of the compact hit; rt_trace_pass_bgra8 writes the pixel as
    8-bit BGRA.  Both must be the same bits as the two-step forms (rt_intersect_async's 16-byte record cut down,
    rt_trace_pass + rt_pack_bgra8), for the plain launch, the cost-sorted launch with split and cooperative quarter waves
    (from the third launch on) and the lane-refilling kernel; nothing is written past the records asked for."""
    import torch
    W, H = 611, 347
    for model, loader in (("dragon.obj", "tolerant"), ("cube_mirror_test.obj", "reference")):
        mesh = rt.Mesh.load_obj(os.path.join(MODELS, model), loader)
        bvh = rt.BVH(mesh, "fixed")
        rays = rt.primary_rays(W, H, rt.frame_camera(bvh.root_box, W, H))
        n = W * H
        scene = rt.Scene(mesh, bvh, device=gpu, width=4)
        try:
            scene.set_param("width", 4)
            scene.set_param("image_width", W)
            scene.set_param("feedback_period", 2)
            rays_d = torch.from_numpy(rays).cuda()
            want = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
            scene.intersect_async(rays_d.data_ptr(), n, None, want.data_ptr())
            torch.cuda.synchronize()
            for rec in (16, 8, 4):
                for rep in range(4):
                    out = torch.full((n + 8, rec // 4), -7, dtype=torch.int32, device="cuda")
                    scene.intersect_compact(rays_d.data_ptr(), n, out.data_ptr(), rec)
                    torch.cuda.synchronize()
                    assert torch.equal(out[:n], want[:, :rec // 4]), (model, rec, rep)
                    assert bool((out[n:] == -7).all())
            with pytest.raises(rt.RtError):
                scene.intersect_compact(rays_d.data_ptr(), n, out.data_ptr(), 12)
            # pixels
            scene.set_param("trace_split", 0)
            rgba = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
            px_want = torch.zeros(n, dtype=torch.int32, device="cuda")
            cam = np.eye(4, dtype=np.float32)
            scene.trace_pass(rays_d.data_ptr(), None, rgba.data_ptr(), n, cam, 0)
            rt.pack_bgra8(gpu, rgba.data_ptr(), n, px_want.data_ptr())
            px = torch.full((n + 8,), -7, dtype=torch.int32, device="cuda")
            recs = torch.zeros((n, 16), dtype=torch.float32, device="cuda")
            recs_want = torch.zeros((n, 16), dtype=torch.float32, device="cuda")
            scene.trace_pass(rays_d.data_ptr(), recs_want.data_ptr(), None, n, cam, 0)
            scene.trace_pass_bgra8(rays_d.data_ptr(), recs.data_ptr(), px.data_ptr(), n, cam, 0)
            torch.cuda.synchronize()
            assert torch.equal(px[:n], px_want) and bool((px[n:] == -7).all())
            assert torch.equal(recs.view(torch.int32), recs_want.view(torch.int32))
            if model == "cube_mirror_test.obj":
                assert len(torch.unique(px_want)) > 2
            # the full trace step with its costliest units split over cooperative quarter waves (always / by the launch's size)
            for ts in (1, -1):
                scene.set_param("trace_split", ts)
                for rep in range(4):
                    px.fill_(-7)
                    recs.zero_()
                    scene.trace_pass_bgra8(rays_d.data_ptr(), recs.data_ptr(), px.data_ptr(), n, cam, 0)
                    torch.cuda.synchronize()
                    assert torch.equal(px[:n], px_want) and bool((px[n:] == -7).all()), (model, ts, rep)
                    assert torch.equal(recs.view(torch.int32), recs_want.view(torch.int32)), (model, ts, rep)
            # the lane-refilling kernel (linear ray set, no image hint)
            scene.set_param("image_width", 0)
            scene.set_param("refill", 40)
            for rec in (8, 4):
                out = torch.full((n + 8, rec // 4), -7, dtype=torch.int32, device="cuda")
                scene.intersect_compact(rays_d.data_ptr(), n, out.data_ptr(), rec)
                torch.cuda.synchronize()
                assert torch.equal(out[:n], want[:, :rec // 4]) and bool((out[n:] == -7).all())
        finally:
            scene.close()


def test_scatter_of_large_squares_and_unaligned_tables(rt, gpu):
    """rt_scatter_tiles with squares large enough to be shared by several workgroups, with 4-, 8- and 16-byte records, and with
    a record table whose squares start at positions that are not multiples of 16 bytes (the 4-byte path)."""
    import torch
    from raytrace_amd.tiles import rays_of_tiles, tile_grid
    rng = np.random.default_rng(11)
    for W, H, tile in ((1024, 520, 256), (1001, 333, 128), (640, 360, 64)):
        tx, ty = tile_grid(W, H, tile)
        order = rng.permutation(tx * ty)
        idx, first = rays_of_tiles(W, H, tile, order)
        for words in (1, 2, 4):
            img = torch.from_numpy(rng.integers(-2**31, 2**31 - 1, (W * H, words), dtype=np.int64).astype(np.int32)).cuda()
            for shift in (0, 3):     # the table displaced by three records: unaligned starts for the small records
                stacked = torch.zeros((W * H + shift, words), dtype=torch.int32, device="cuda")
                stacked[shift:] = img[torch.from_numpy(idx).cuda()]
                ids_d = torch.from_numpy(order.astype(np.int32)).cuda()
                first_d = torch.from_numpy((first.astype(np.int64) + shift).astype(np.int32)).cuda()
                fb = torch.zeros_like(img)
                rt.scatter_tiles(gpu, stacked.data_ptr(), W * H + shift, ids_d.data_ptr(), first_d.data_ptr(), len(order), tile, W, H, 4 * words,
                                 fb.data_ptr())
                torch.cuda.synchronize()
                assert torch.equal(fb, img), (W, H, tile, words, shift)
