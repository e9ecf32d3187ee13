"""The N > 1 host logic on CPU: two gloo ranks split an image into interleaved tiles, each
"traces" its own rays (here with the CPU oracle standing in for the device call), rank 0 gathers
with point-to-point sends and rebuilds the framebuffer, which must equal the single-rank one byte
for byte."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import MODELS, ROOT

W, H, TILE = 70, 45, 16    # deliberately not multiples of the tile size


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    import raytrace_amd as rt
    from raytrace_amd.tiles import scatter_to_framebuffer, tile_indices
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, "teapot.obj"), "tolerant")
    bvh = rt.BVH(mesh, "fixed")
    rays = rt.primary_rays(W, H, rt.frame_camera(bvh.root_box, W, H))
    own = tile_indices(W, H, TILE, rank, world)
    ids, t, u, v = oracle.closest_hit(mesh.verts, mesh.faces, rays[own], threads=2)
    mine = torch.from_numpy(np.stack([ids.view(np.float32), t, u, v], axis=1).copy())
    if rank == 0:
        counts = [len(tile_indices(W, H, TILE, r, world)) for r in range(world)]
        bufs = [mine] + [torch.empty((c, 4), dtype=torch.float32) for c in counts[1:]]
        reqs = [dist.irecv(bufs[r], src=r) for r in range(1, world)]
        for q in reqs:
            q.wait()
        fb = scatter_to_framebuffer(W, H, TILE, world, [b.numpy() for b in bufs])
        np.save(out_path, fb)
    else:
        dist.isend(mine, dst=0).wait()
    dist.barrier()
    dist.destroy_process_group()


def test_partition_is_exact_cover():
    from raytrace_amd.tiles import tile_indices
    for world in (1, 2, 3, 8):
        allidx = np.concatenate([tile_indices(W, H, TILE, r, world) for r in range(world)])
        assert np.array_equal(np.sort(allidx), np.arange(W * H))
    sizes = [len(tile_indices(3840, 2160, 64, r, 8)) for r in range(8)]
    assert max(sizes) - min(sizes) <= 64 * 64 * 2     # balanced to within two tiles


def test_two_rank_gather_equals_single_rank(tmp_path):
    import oracle
    import raytrace_amd as rt
    out = str(tmp_path / "fb.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    fb = np.load(out)
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, "teapot.obj"), "tolerant")
    bvh = rt.BVH(mesh, "fixed")
    rays = rt.primary_rays(W, H, rt.frame_camera(bvh.root_box, W, H))
    ids, t, u, v = oracle.closest_hit(mesh.verts, mesh.faces, rays)
    single = np.stack([ids.view(np.float32), t, u, v], axis=1)
    assert fb.tobytes() == single.tobytes()
    assert int((ids != rt.MISS_ID).sum()) > 50
