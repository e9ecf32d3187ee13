"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on the same rays.

Bar: hit triangle ids bit-exact; t, u, v bit-exact as well (the kernel performs the reference's
operations in the reference's order with contraction off, so not even the 1e-5 relative tolerance
the spec allows is needed -- it is asserted at 0 ulp).  Hit point and normal, whose reference
formula is OpenCL float3 arithmetic with implementation-defined contraction, are held to 1e-5
relative."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from conftest import MODELS, ROOT, bits, framed_primary
from test_oracle import random_pairs

pytestmark = pytest.mark.gpu

f32 = np.float32
REL = 1e-5  # tolerance for the floating-point outputs that are not bit-pinned (hit point, normal)


def compare(rt, oracle, mesh, rays, hits, sel=None):
    ids, t, u, v = rt.hit_fields(hits)
    if sel is None:
        sel = np.arange(len(rays))
    oid, ot, ou, ov, second = oracle.closest_hit(mesh.verts, mesh.faces, rays[sel], want_second=True)
    assert np.array_equal(ids[sel], oid), f"{int((ids[sel] != oid).sum())} hit ids differ"
    h = oid != rt.MISS_ID
    for mine, ref in ((t, ot), (u, ou), (v, ov)):
        assert np.array_equal(bits(mine[sel][h]), bits(ref[h]))
    miss = ~h
    assert np.all(bits(t[sel][miss]) == 0x7F7FFFFF) and np.all(u[sel][miss] == 0) and np.all(v[sel][miss] == 0)
    near_ties = int((np.abs(second[h] - ot[h]) <= 1e-6 * np.abs(ot[h])).sum())
    return int(h.sum()), near_ties


@pytest.mark.parametrize("width", [2, 4])
@pytest.mark.parametrize("quant", [1, 0])
@pytest.mark.parametrize("model,loader,W,H,stride", [
    ("cube.obj", "tolerant", 512, 512, 1),             # exact ties on the shared face diagonals
    ("cube_mirror_test.obj", "reference", 200, 120, 1),
    ("teapot.obj", "reference", 160, 128, 1),
    ("bunny.obj", "tolerant", 1920, 1080, 257),
    ("dragon.obj", "tolerant", 1920, 1080, 257),
])
def test_primary_rays_match_oracle(rt, oracle, gpu, model, loader, W, H, stride, width, quant):
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, model), loader)
    bvh = rt.BVH(mesh, "fixed")
    scene = rt.Scene(mesh, bvh, device=gpu, width=width)
    scene.set_param("quantized", quant)
    if W % 8 == 0:
        scene.set_param("image_width", W)
    rays = framed_primary(rt, bvh, W, H)
    hits = scene.intersect(rays)
    sel = None if stride == 1 else np.arange(0, len(rays), stride)
    nhit, ties = compare(rt, oracle, mesh, rays, hits, sel)
    assert nhit > 0
    if model == "cube.obj":
        assert ties > 0          # the tie rule is really exercised
    scene.close()


def test_reference_bvh_and_reference_loader_give_the_same_hits(rt, oracle, gpu):
    """The literal reference tree (core/CentroidBVH.cpp:65 as written) is slower, the surface-area tree faster,
    neither is different."""
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, "teapot.obj"), "reference")
    rays = framed_primary(rt, rt.BVH(mesh, "fixed"), 128, 96)
    out = []
    for mode in ("reference", "fixed", "sah"):
        for width in (2, 4):
            s = rt.Scene(mesh, rt.BVH(mesh, mode), device=gpu, width=width)
            out.append(s.intersect(rays))
            s.close()
    for o in out[1:]:
        assert np.array_equal(bits(o), bits(out[0]))
    compare(rt, oracle, mesh, rays, out[0])


GOLDEN = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz")))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
@pytest.mark.parametrize("width", [2, 4])
def test_golden_vectors(rt, gpu, path, width):
    g = np.load(path)
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, str(g["model"])), str(g["loader"]))
    scene = rt.Scene(mesh, rt.BVH(mesh, "fixed"), device=gpu, width=width)
    hits = scene.intersect(rt.primary_rays(int(g["width"]), int(g["height"]), g["origin"]))
    ids, t, u, v = rt.hit_fields(hits)
    assert np.array_equal(ids, g["ids"])
    h = ids != rt.MISS_ID
    for mine, gold in ((t, g["t"]), (u, g["u"]), (v, g["v"])):
        assert np.array_equal(bits(mine[h]), bits(gold[h]))


def test_kernel_triangle_test_is_bitwise_the_oracle(rt, oracle, gpu):
    rng = np.random.default_rng(424242)
    accepted = 0
    for _ in range(3):
        tri, ray = random_pairs(rng, 2_000_000)
        h0, t0, uv0 = oracle.tri_batch(tri, ray)
        h1, t1, uv1 = rt.debug_tri_batch(tri, ray, device=gpu)
        assert np.array_equal(h0, h1)
        m = h0 == 1
        assert np.array_equal(bits(t0)[m], bits(t1)[m]) and np.array_equal(bits(uv0)[m], bits(uv1)[m])
        accepted += int(m.sum())
    assert accepted > 500_000


def test_incoherent_rays_match_oracle(rt, oracle, gpu):
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, "bunny_big.obj"), "tolerant")
    bvh = rt.BVH(mesh, "fixed")
    rays = rt.ao_rays(mesh, bvh.root_box, 1234, 0, 1 << 20)
    sel = np.arange(0, len(rays), 509)
    for width in (2, 4):
        scene = rt.Scene(mesh, bvh, device=gpu, width=width)
        nhit, _ = compare(rt, oracle, mesh, rays, scene.intersect(rays), sel)
        assert 0 < nhit < len(sel)
        scene.close()


def test_lane_refill_kernel_equals_the_batch_kernel_and_the_oracle(rt, oracle, gpu):
    """"refill" > 0 sends linear ray sets to the kernel whose waves stream through the rays and hand new ones to lanes
    as they finish.  Which rays share a wave, and when, must not show in any record: 64-byte records and 16-byte hits of
    1 048 576 + 777 incoherent rays equal the batch kernel's bit for bit for three thresholds, and every hit equals the
    oracle's tree walk."""
    import torch
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, "bunny_big.obj"), "tolerant")
    bvh = rt.BVH(mesh, "fixed")
    n = (1 << 20) + 777                      # not a multiple of the chunk a wave claims, nor of 64
    rays = rt.ao_rays(mesh, bvh.root_box, 4321, 0, n)
    d_rays = torch.from_numpy(rays).cuda()
    scene = rt.Scene(mesh, bvh, device=gpu, width=4)
    stream = torch.cuda.current_stream().cuda_stream
    full = torch.empty((n, 16), dtype=torch.float32, device="cuda")
    compact = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    scene.set_param("refill", 0)
    scene.intersect_async(d_rays.data_ptr(), n, full.data_ptr(), compact.data_ptr(), stream)
    torch.cuda.synchronize()
    want_full, want_compact = full.clone(), compact.clone()
    oid, ot, ou, ov = oracle.closest_hit_bvh(mesh.verts, mesh.faces, bvh.nodes, rays)
    got = want_compact.cpu().numpy()
    h = oid != rt.MISS_ID
    assert 0.02 * n < h.sum() < 0.9 * n
    assert np.array_equal(got[:, 0].view(np.uint32), oid)
    for k, want in ((1, ot), (2, ou), (3, ov)):
        assert np.array_equal(bits(got[h, k]), bits(want[h]))
    for threshold in (1, 32, 64):
        scene.set_param("refill", threshold)
        for outs in ((full, compact), (full, None), (None, compact)):
            full.fill_(-9.0)
            compact.fill_(-9.0)
            scene.intersect_async(d_rays.data_ptr(), n, outs[0].data_ptr() if outs[0] is not None else None,
                                  outs[1].data_ptr() if outs[1] is not None else None, stream)
            torch.cuda.synchronize()
            if outs[0] is not None:
                assert torch.equal(full.view(torch.int32), want_full.view(torch.int32)), threshold
            if outs[1] is not None:
                assert torch.equal(compact.view(torch.int32), want_compact.view(torch.int32)), threshold
    scene.close()


def test_incoherent_rays_at_the_full_64_mi(rt, oracle, gpu):
    """The divergence-stress configuration at its stated size: 64 Mi seeded AO-style rays through bunny_big in one
    call (4.3 GB of ray records resident in HBM, 1 GB of compact hits).  Every one of the 67 108 864 hits is compared
    (ids and the bits of t, u, v) with the oracle's tree walk, slice by slice; a sub-sample spread over the whole set is
    also held against the brute-force loop, and three more launches, the lane-refilling kernel and the two-wide layout
    must return the same bits."""
    import torch
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, "bunny_big.obj"), "tolerant")
    bvh = rt.BVH(mesh, "fixed")
    n, piece = 64 << 20, 4 << 20
    rays_d = torch.empty((n, 16), dtype=torch.float32, device="cuda")
    step = 16411                                     # sub-sample stride (prime), 4089 rays
    sample, walked = [], []
    for off in range(0, n, piece):                   # generated and uploaded in slices: no 4.3 GB host table
        part = rt.ao_rays(mesh, bvh.root_box, 1234, off, piece)
        first = (-off) % step
        sample.append(part[first::step].copy())
        rays_d[off:off + piece].copy_(torch.from_numpy(part))
        walked.append(np.stack([x.view(np.uint32) for x in oracle.closest_hit_bvh(mesh.verts, mesh.faces, bvh.nodes, part)], axis=1))
    sample = np.concatenate(sample)
    scene = rt.Scene(mesh, bvh, device=gpu, width=4)
    scene.set_param("feedback", 1)
    try:
        hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
        scene.intersect_async(rays_d.data_ptr(), n, None, hits.data_ptr())
        torch.cuda.synchronize()
        for k, want in enumerate(walked):            # every ray: ids everywhere, t/u/v where there is a hit
            got = hits[k * piece:(k + 1) * piece].cpu().numpy().view(np.uint32)
            assert np.array_equal(got[:, 0], want[:, 0]), k
            hk = want[:, 0] != rt.MISS_ID
            assert np.array_equal(got[hk], want[hk]), k
        del walked
        ids = hits[:, 0].view(torch.int32)
        hit = ids != -1
        assert bool(((ids >= 0) & (ids < len(mesh.faces)) | ~hit).all())
        share = float(hit.float().mean())
        assert 0.02 < share < 0.5, share
        again = torch.empty_like(hits)
        for rep in range(3):                         # launches 3 and 4 follow a cost-sorted order of 2^20 units, the costliest split
            again.zero_()
            scene.intersect_async(rays_d.data_ptr(), n, None, again.data_ptr())
            torch.cuda.synchronize()
            assert torch.equal(again.view(torch.int32), hits.view(torch.int32)), rep
        scene.set_param("refill", 40)                # the lane-refilling kernel: which rays share a wave must not show
        again.zero_()
        scene.intersect_async(rays_d.data_ptr(), n, None, again.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(again.view(torch.int32), hits.view(torch.int32))
        scene.set_param("refill", 0)
        scene.set_param("width", 2)
        scene.intersect_async(rays_d.data_ptr(), piece, None, again.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(again[:piece].view(torch.int32), hits[:piece].view(torch.int32))
        got = hits[0::step].cpu().numpy()
        assert len(got) == len(sample)
        oid, ot, ou, ov = oracle.closest_hit(mesh.verts, mesh.faces, sample)
        assert np.array_equal(got[:, 0].view(np.uint32), oid)
        h = oid != rt.MISS_ID
        assert h.sum() > 100
        for k, want in ((1, ot), (2, ou), (3, ov)):
            assert np.array_equal(bits(got[h, k]), bits(want[h]))
    finally:
        scene.close()


def test_every_ray_of_the_full_size_frames_matches_the_oracle(rt, oracle, gpu):
    """All 2 073 600 rays of the 1080p dragon frame, all 2 073 600 of the bunny frame and all 8 294 400 of the 4K dragon
    frame against the oracle's tree walk (which equals its brute-force loop bit for bit, tests/test_oracle.py): ids and the
    bits of t, u, v."""
    for model, W, H in (("dragon.obj", 1920, 1080), ("bunny.obj", 1920, 1080), ("dragon.obj", 3840, 2160)):
        mesh = rt.Mesh.load_obj(os.path.join(MODELS, model), "tolerant")
        bvh = rt.BVH(mesh, "fixed")
        rays = framed_primary(rt, bvh, W, H)
        s = rt.Scene(mesh, bvh, device=gpu, width=4)
        s.set_param("image_width", W)
        s.set_param("feedback", 1)       # bench.py's schedule: the fourth launch follows a cost-sorted order, 32 tiles split
        import torch
        d_rays = torch.from_numpy(rays).cuda()
        d_hits = torch.empty_like(d_rays)
        for rep in range(4):
            d_hits.fill_(-3.0)
            s.intersect_async(d_rays.data_ptr(), len(rays), d_hits.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        hits = d_hits.cpu().numpy()
        del d_rays, d_hits
        s.close()
        ids, t, u, v = rt.hit_fields(hits)
        oid, ot, ou, ov = oracle.closest_hit_bvh(mesh.verts, mesh.faces, bvh.nodes, rays)
        h = oid != rt.MISS_ID
        assert h.sum() > 0.2 * len(rays)
        assert np.array_equal(ids, oid)
        assert np.array_equal(bits(t[h]), bits(ot[h])) and np.array_equal(bits(u[h]), bits(ou[h])) and np.array_equal(bits(v[h]), bits(ov[h]))


@pytest.mark.parametrize("model,loader", [("teapot.obj", "reference"), ("sphere.obj", "tolerant"), ("wheel.obj", "tolerant"),
                                          ("cube_mirror_test.obj", "reference"), ("bunny_big.obj", "tolerant"),
                                          ("std_square.obj", "tolerant"), ("tri.obj", "tolerant")])
def test_every_model_every_layout_every_ray(rt, oracle, gpu, model, loader):
    """Each shipped mesh, each tree (centroid and surface-area), each record layout (two- / four-wide, quantized / float boxes),
    a 1280x720 frame with and without the image hint: all 921 600 hits against the oracle's tree walk, bit for bit."""
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, model), loader)
    W, H = 1280, 720
    for mode in ("fixed", "sah"):
        bvh = rt.BVH(mesh, mode)
        rays = framed_primary(rt, bvh, W, H)
        want = np.stack([x.view(np.uint32) for x in oracle.closest_hit_bvh(mesh.verts, mesh.faces, bvh.nodes, rays)], axis=1)
        h = want[:, 0] != rt.MISS_ID
        assert h.sum() > 1000, (model, mode)
        for width, quant, hint in ((4, 1, W), (2, 1, W), (4, 0, 0), (2, 0, W), (4, 1, 0)):
            s = rt.Scene(mesh, bvh, device=gpu, width=width)
            s.set_param("quantized", quant)
            s.set_param("image_width", hint)
            got = np.stack([x.view(np.uint32) for x in rt.hit_fields(s.intersect(rays))], axis=1)
            s.close()
            assert np.array_equal(got[:, 0], want[:, 0]), (model, mode, width, quant, hint)
            assert np.array_equal(got[h], want[h]), (model, mode, width, quant, hint)


def test_full_size_properties(rt, gpu):
    """Full benchmark sizes, checked through size-independent properties: the record layouts and both
    ray-to-lane mappings agree on every one of the 2 073 600 / 8 294 400 rays, hit records are
    self-consistent (origin + t*dir lands on the reported hit point, barycentrics inside the
    triangle), and tracing the hit set again is idempotent."""
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, "dragon.obj"), "tolerant")
    bvh = rt.BVH(mesh, "fixed")
    for (W, H) in ((1920, 1080), (3840, 2160)):
        rays = framed_primary(rt, bvh, W, H)
        ref = None
        for width, quant, hint in ((2, 1, 0), (4, 1, W), (2, 0, W), (4, 0, 0)):
            s = rt.Scene(mesh, bvh, device=gpu, width=width)
            s.set_param("quantized", quant)
            s.set_param("image_width", hint)
            hits = s.intersect(rays)
            s.close()
            if ref is None:
                ref = hits
            else:
                assert np.array_equal(bits(hits), bits(ref))
        ids, t, u, v = rt.hit_fields(ref)
        h = ids != rt.MISS_ID
        assert 0.3 < h.mean() < 0.45 and ids[h].max() < len(mesh.faces)
        assert np.all(u[h] >= 0) and np.all(v[h] >= 0) and np.all(u[h] + v[h] <= 1) and np.all(t[h] > 1e-6)
        p = rays[h, 0:3] + t[h, None] * rays[h, 4:7]
        scale = np.abs(rays[h, 0:3]).max()
        err = np.abs(p - ref[h, 0:3]).max(axis=1)
        # the triangle test's own t is ill-conditioned at grazing incidence, so the tail is loose
        assert np.median(err) <= 1e-6 * scale and np.quantile(err, 0.999) <= 2e-5 * scale and err.max() <= 1e-3 * scale
        # miss records carry the reference's defaults (tracekernel.ck:282-286)
        assert np.array_equal(bits(ref[~h, 0:4]), bits(rays[~h, 0:4])) and np.array_equal(bits(ref[~h, 8:12]), bits(rays[~h, 4:8]))


def test_hit_record_fields(rt, oracle, gpu):
    """Hit point, interpolated normal (the reference's u*N[n0] + v*N[n1] + (1-u-v)*N[n2] pairing,
    tracekernel.ck:376), direction, uv, id and t of the 64-byte record: bit for bit oracle.c's restatement of
    tracekernel.ck:282-286, :373-399 (itself parity-unpinned for the interpolated fields: OpenCL C cannot run
    here; the evaluation order is the one written in the kernel source, unfused)."""
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, "teapot.obj"), "reference")
    bvh = rt.BVH(mesh, "fixed")
    scene = rt.Scene(mesh, bvh, device=gpu)
    rays = framed_primary(rt, bvh, 96, 96)
    hits = scene.intersect(rays)
    ids, t, u, v = rt.hit_fields(hits)
    h = ids != rt.MISS_ID
    oid, ot, ou, ov = oracle.closest_hit(mesh.verts, mesh.faces, rays)
    want, _, defined = oracle.trace_epilogue(mesh, rays, np.eye(4, dtype=np.float32), 0, oid, ot, ou, ov)
    assert defined.all() and h.sum() > 500
    assert np.array_equal(bits(hits), bits(want))
    assert np.all(hits[h, 3] == 1) and np.all(hits[h, 11] == 0) and np.all(hits[h, 7] == 0)
    # a mesh without vn records gets unit geometric normals
    cube = rt.Mesh.load_obj(os.path.join(MODELS, "cube.obj"), "tolerant")
    cb = rt.BVH(cube, "fixed")
    cs = rt.Scene(cube, cb, device=gpu)
    ch = cs.intersect(framed_primary(rt, cb, 64, 64))
    hh = ch[:, 14].view(np.uint32) != rt.MISS_ID
    assert np.allclose(np.linalg.norm(ch[hh, 8:11], axis=1), 1, atol=1e-6)


def test_edge_cases(rt, oracle, gpu):
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, "cube.obj"), "tolerant")
    bvh = rt.BVH(mesh, "fixed")
    for width in (2, 4):
        scene = rt.Scene(mesh, bvh, device=gpu, width=width)
        assert scene.intersect(np.zeros((0, 16), np.float32)).shape == (0, 16)          # empty
        base = framed_primary(rt, bvh, 64, 64)
        for n in (1, 63, 64, 65, 257):                                                 # ragged sizes
            compare(rt, oracle, mesh, base[:n], scene.intersect(base[:n]))
        for (w, h) in ((70, 45), (8, 8), (9, 1), (1, 9), (64, 3)):                      # images that do not tile by 8x8
            scene.set_param("image_width", w)
            img = framed_primary(rt, bvh, w, h)
            compare(rt, oracle, mesh, img, scene.intersect(img))
        scene.set_param("image_width", 0)
        r = np.zeros((12, 16), np.float32)
        r[:, 0:3] = (0.5, 0.5, -3)
        r[0, 4:7] = (0, 0, 1)              # axis-parallel: two zero direction components
        r[1, 4:7] = (0, 0, -1)             # pointing away
        r[2, 0:3] = (0.5, 0.5, 0.5); r[2, 4:7] = (0, 1, 0)      # origin inside the cube
        r[3, 0:3] = (1.0, 0.5, -3); r[3, 4:7] = (0, 0, 1)       # exactly along the x = 1 face
        r[4, 0:3] = (0.0, 0.0, -3); r[4, 4:7] = (0, 0, 1)       # exactly through a corner edge
        r[5, 4:7] = (0, 0, 0)              # zero direction
        r[6, 4:7] = (np.nan, 0, 1)         # NaN direction
        r[7, 0:3] = (np.nan, 0, 0); r[7, 4:7] = (0, 0, 1)
        r[8, 4:7] = (0, 0, 1e-30)          # denormal-scale direction
        r[9, 4:7] = (0, 0, 1e30)
        r[10, 0:3] = (0.5, 0.5, 0.0); r[10, 4:7] = (0, 0, 1)    # origin on a face: t = 0 is rejected, far face hit
        r[11, 0:3] = (1e6, 1e6, -1e6); r[11, 4:7] = (-1, -1, 1)  # far-away origin
        compare(rt, oracle, mesh, r, scene.intersect(r))
        scene.close()
    # scenes of <= 4 triangles and coincident triangles
    sq = rt.Mesh.load_obj(os.path.join(MODELS, "std_square.obj"), "tolerant")
    sb = rt.BVH(sq, "fixed")
    ss = rt.Scene(sq, sb, device=gpu)
    rr = framed_primary(rt, sb, 33, 17)
    compare(rt, oracle, sq, rr, ss.intersect(rr))
    verts = np.array([[0, 0, 0, 1], [1, 0, 0, 1], [0, 1, 0, 1]], np.float32)
    faces = np.zeros((9, 8), np.uint32); faces[:, :3] = (0, 1, 2)
    co = rt.Mesh.from_arrays(verts, faces)
    cbv = rt.BVH(co, "fixed")
    cs = rt.Scene(co, cbv, device=gpu, width=4)
    rr = framed_primary(rt, cbv, 16, 16)
    nhit, ties = compare(rt, oracle, co, rr, cs.intersect(rr))
    assert nhit > 0 and ties == nhit      # nine coincident triangles: index 0 must win every time


def test_device_pointers_streams_and_errors(rt, oracle, gpu):
    import torch
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, "teapot.obj"), "reference")
    bvh = rt.BVH(mesh, "fixed")
    scene = rt.Scene(mesh, bvh, device=gpu)
    rays = framed_primary(rt, bvh, 128, 128)
    want = scene.intersect(rays)
    d_rays = torch.from_numpy(rays).cuda()
    d_hits = torch.zeros_like(d_rays)
    scene.intersect_device(d_rays.data_ptr(), len(rays), d_hits.data_ptr())       # device in, device out
    assert np.array_equal(bits(d_hits.cpu().numpy()), bits(want))
    d16 = torch.zeros((len(rays), 4), dtype=torch.float32, device="cuda")
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        scene.intersect_async(d_rays.data_ptr(), len(rays), None, d16.data_ptr(), st.cuda_stream)
    st.synchronize()
    h16 = d16.cpu().numpy()
    assert np.array_equal(bits(h16[:, 0]), bits(want[:, 14])) and np.array_equal(bits(h16[:, 1]), bits(want[:, 15]))
    assert np.array_equal(bits(h16[:, 2:4]), bits(want[:, 12:14]))
    # host in, device out and the reverse
    d_hits.zero_()
    assert rt.lib.ray_scene_intersection(C.c_void_p(rays.ctypes.data), len(rays), scene._h, C.c_void_p(d_hits.data_ptr())) == 0
    assert np.array_equal(bits(d_hits.cpu().numpy()), bits(want))
    # errors: message, negative status, nothing aborts
    assert rt.lib.ray_scene_intersection(None, 5, scene._h, C.c_void_p(d_hits.data_ptr())) == -1
    assert rt.lib.rt_intersect_async(scene._h, C.c_void_p(d_rays.data_ptr()), 5, None, None, None) == -1
    h = C.c_void_p()
    nodes = bvh.nodes
    assert rt.lib.rt_scene_create(C.c_void_p(mesh.verts.ctypes.data), len(mesh.verts), C.c_void_p(mesh.faces.ctypes.data),
                                  len(mesh.faces), None, 0, None, 0, C.c_void_p(nodes.ctypes.data), len(nodes),
                                  99, 0, C.byref(h)) == -3
    assert rt.lib.rt_scene_create(C.c_void_p(mesh.verts.ctypes.data), len(mesh.verts), C.c_void_p(mesh.faces.ctypes.data),
                                  len(mesh.faces), None, 0, None, 0, C.c_void_p(nodes.ctypes.data), len(nodes),
                                  gpu, 3, C.byref(h)) == -1


def test_mixed_host_device_calls_are_ordered_behind_the_null_stream(rt, oracle, gpu):
    """ray_scene_intersection keeps the reference's in-order queue semantics (one in-order queue, clFinish per pass:
    render_client/img_main.cpp:313,480) when one side of the call is a device buffer: device rays that earlier null-stream
    work is still producing are read only after it, and device hits that earlier null-stream work still writes are not
    overtaken.  Both directions, with a few milliseconds of null-stream work enqueued right before the call."""
    import ctypes as C
    import torch
    mesh, bvh, scene = rt.load_scene("teapot.obj", device=gpu, width=4)
    rays = framed_primary(rt, bvh, 512, 384)
    want = scene.intersect(rays)
    n = len(rays)
    src = torch.from_numpy(rays).cuda()
    ballast = torch.randn((4096, 4096), device="cuda")
    torch.cuda.synchronize()
    for rep in range(3):
        # device rays -> host hits: the rays reach their buffer at the END of a long null-stream chain
        d_rays = torch.zeros_like(src)
        hits_h = np.full_like(rays, -3.0)
        for _ in range(6):
            ballast = ballast @ ballast * 1e-4
        d_rays.copy_(src, non_blocking=True)
        assert rt.lib.ray_scene_intersection(C.c_void_p(d_rays.data_ptr()), n, scene._h, C.c_void_p(hits_h.ctypes.data)) == 0
        assert np.array_equal(bits(hits_h), bits(want)), rep
        # host rays -> device hits: a fill of the hit buffer sits at the end of the same kind of chain
        d_hits = torch.empty((n, 16), dtype=torch.float32, device="cuda")
        for _ in range(6):
            ballast = ballast @ ballast * 1e-4
        d_hits.fill_(-5.0)
        assert rt.lib.ray_scene_intersection(C.c_void_p(rays.ctypes.data), n, scene._h, C.c_void_p(d_hits.data_ptr())) == 0
        assert np.array_equal(bits(d_hits.cpu().numpy()), bits(want)), rep
    scene.close()


def test_a_tree_deeper_than_the_default_stack_grows_the_lds_part(rt, oracle, gpu):
    """A caller's degenerate tree: a two-wide chain of 90 inner nodes, every one holding a one-triangle leaf and the rest of
    the chain.  Its worst-case stack (91 two-wide, 3 * depth + 1 four-wide) is beyond the default LDS part plus the private
    part; the launch grows the LDS part instead of failing, and every hit equals the brute-force oracle."""
    D = 91   # triangles = leaves; D - 1 inner nodes
    verts = np.zeros((3 * D, 4), np.float32)
    faces = np.zeros((D, 8), np.uint32)
    for i in range(D):
        x = 2.0 * i
        verts[3 * i:3 * i + 3, :3] = [[x, 0, 0], [x + 1, 0, 0], [x, 1, 0.25]]
        verts[3 * i:3 * i + 3, 3] = 1.0
        faces[i, :3] = [3 * i, 3 * i + 1, 3 * i + 2]
    mesh = rt.Mesh.from_arrays(verts, faces)
    nodes = np.zeros((2 * D - 1, 20), np.uint32)

    def box_of(lo_tri, hi_tri):
        v = verts[3 * lo_tri:3 * hi_tri + 3, :3]
        lo, hi = v.min(axis=0) - 0.01, v.max(axis=0) + 0.01
        return np.concatenate([lo, [1.0], hi, [1.0]]).astype(np.float32).view(np.uint32)

    # node 2k = inner over triangles k..D-1 (children: leaf 2k+1 with triangle k, inner 2k+2), last node = leaf D-1
    for k in range(D - 1):
        inner, leaf, rest = 2 * k, 2 * k + 1, 2 * k + 2
        nodes[inner, 4:6] = [leaf, rest]
        nodes[inner, 8:16] = box_of(k, D - 1)
        nodes[inner, 16] = 0          # RT_BVH_INNER
        nodes[inner, 17] = max(0, inner - 2)
        nodes[leaf, 0] = k
        nodes[leaf, 4] = 1            # leaf.count
        nodes[leaf, 8:16] = box_of(k, k)
        nodes[leaf, 16] = 1           # RT_BVH_LEAF
        nodes[leaf, 17] = inner
    last = 2 * D - 2
    nodes[last, 0] = D - 1
    nodes[last, 4] = 1
    nodes[last, 8:16] = box_of(D - 1, D - 1)
    nodes[last, 16] = 1
    nodes[last, 17] = last - 2
    rng = np.random.default_rng(5)
    n = 4096
    rays = np.zeros((n, 16), np.float32)
    rays[:, 0] = rng.uniform(-1, 2 * D + 1, n)
    rays[:, 1] = rng.uniform(-0.5, 1.5, n)
    rays[:, 2] = -3.0
    rays[:, 3] = 1.0
    d = np.stack([rng.uniform(-0.8, 0.8, n), rng.uniform(-0.1, 0.1, n), np.ones(n)], axis=1)
    rays[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True)
    oid, ot, ou, ov = oracle.closest_hit(mesh.verts, mesh.faces, rays)
    assert (oid != rt.MISS_ID).sum() > 100
    for width in (2, 4):
        scene = rt.Scene(mesh, nodes, device=gpu, width=width)
        scene.set_param("width", width)
        got = scene.intersect(rays)
        scene.close()
        ids, t, u, v = rt.hit_fields(got)
        hit = oid != rt.MISS_ID
        assert np.array_equal(ids, oid), width
        assert np.array_equal(bits(t[hit]), bits(ot[hit])) and np.array_equal(bits(u[hit]), bits(ou[hit])), width


def test_one_scene_on_several_streams_with_the_cost_feedback(rt, gpu):
    """Launches of one scene may go to any streams.  With the cost feedback on, the sort that rewrites a dispatch-order
    table must wait for readers on every stream, not only the one it follows: sixty launches dealt to three streams
    (each with its own hit buffer, re-sorted every second launch, the costliest units split) all return the bits of a
    launch without feedback."""
    import torch
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, "bunny.obj"), "tolerant")
    bvh = rt.BVH(mesh, "fixed")
    W, H = 640, 360
    rays = framed_primary(rt, bvh, W, H)
    scene = rt.Scene(mesh, bvh, device=gpu, width=4)
    scene.set_param("image_width", W)
    scene.set_param("feedback", 0)
    d_rays = torch.from_numpy(rays).cuda()
    base = torch.zeros((len(rays), 4), dtype=torch.float32, device="cuda")
    scene.intersect_async(d_rays.data_ptr(), len(rays), None, base.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    scene.set_param("feedback", 1)
    scene.set_param("feedback_period", 2)
    streams = [torch.cuda.Stream() for _ in range(3)]
    outs = [torch.zeros_like(base) for _ in range(3)]
    for k in range(60):
        i = (k * 7 + k // 5) % 3          # an irregular deal, including runs on one stream
        scene.intersect_async(d_rays.data_ptr(), len(rays), None, outs[i].data_ptr(), streams[i].cuda_stream)
        if k % 20 == 19:
            torch.cuda.synchronize()
            for o in outs:
                assert torch.equal(o.view(torch.int32), base.view(torch.int32)), k
                o.zero_()
            # every buffer gets traced again before the next check
            for i2 in range(3):
                scene.intersect_async(d_rays.data_ptr(), len(rays), None, outs[i2].data_ptr(), streams[i2].cuda_stream)
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o.view(torch.int32), base.view(torch.int32))
    scene.close()


def test_fetch_counters_agree_between_host_and_device(rt, gpu):
    import torch
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, "bunny.obj"), "tolerant")
    bvh = rt.BVH(mesh, "fixed")
    rays = framed_primary(rt, bvh, 320, 200)
    d_rays = torch.from_numpy(rays).cuda()
    for width in (2, 4):
        scene = rt.Scene(mesh, bvh, device=gpu, width=width)
        for quant in (1, 0):
            scene.set_param("quantized", quant)
            dev = scene.count_fetches_device(d_rays.data_ptr(), len(rays))
            host = scene.count_fetches(rays)
            assert dev == host, (quant, dev, host)
        scene.close()


def test_stack_spill_to_private_memory_is_exact(rt, oracle, gpu):
    """The traversal stack keeps `lds_stack` entries per lane in LDS and spills deeper ones to private memory.
    Incoherent rays through bunny_big go deeper than 8 entries; with the LDS part at its minimum the spill path
    carries real traffic and every record must still equal the all-LDS launch and the oracle."""
    import torch
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, "bunny_big.obj"), "tolerant")
    bvh = rt.BVH(mesh, "fixed")
    rays = rt.ao_rays(mesh, bvh.root_box, 99, 0, 1 << 17)
    d_rays = torch.from_numpy(rays).cuda()
    for width in (4, 2):
        scene = rt.Scene(mesh, bvh, device=gpu, width=width)
        scene.set_param("lds_stack", 32)
        deep = scene.count_fetches_device(d_rays.data_ptr(), len(rays))["max_stack"]
        want = scene.intersect(rays)
        scene.set_param("lds_stack", 8)
        got = scene.intersect(rays)
        scene.close()
        if width == 4:
            assert deep > 8, f"test rays too shallow to spill (deepest stack {deep})"
            compare(rt, oracle, mesh, rays, got, np.arange(0, len(rays), 257))
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), width


def test_schedule_knobs_never_change_a_result(rt, oracle, gpu):
    """Dispatch order, cost feedback between launches, step scheduling threshold and workgroup size only
    decide WHEN a ray is traced.  Every setting, and every repeat of a launch (the feedback order changes
    from launch to launch), must return the same bits; one of them is checked against the oracle."""
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, "bunny.obj"), "tolerant")
    bvh = rt.BVH(mesh, "fixed")
    W, H = 640, 360
    rays = framed_primary(rt, bvh, W, H)
    import torch
    scene = rt.Scene(mesh, bvh, device=gpu, width=2)
    base = scene.intersect(rays)
    compare(rt, oracle, mesh, rays, base, np.arange(0, len(rays), 61))
    d_out = torch.empty((len(rays), 16), dtype=torch.float32, device="cuda")

    def launch(table):
        """rt_intersect_async on device buffers: the entry point whose launches follow the dispatch order and the cost
        feedback (the host-pointer call cuts its input into chunks and launches those plainly)."""
        d_in = torch.from_numpy(np.ascontiguousarray(table)).cuda()
        d_out.fill_(-3.0)
        scene.intersect_async(d_in.data_ptr(), len(table), d_out.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return d_out[:len(table)].cpu().numpy()

    settings = [
        dict(image_width=W, centre_first=0),
        dict(image_width=W, centre_first=1),
        dict(image_width=W, feedback=1),
        dict(image_width=0, feedback=1),
        dict(image_width=W, persistent=1),
        dict(image_width=W, persistent=1, feedback=1),
        dict(image_width=0, persistent=1, persistent_waves=3),
        dict(image_width=W, lds_stack=8),
        dict(image_width=W, feedback=1, block_threads=256),
        dict(image_width=W, feedback=1, block_threads=128, feedback_radius=0),
        dict(image_width=W, feedback=1, feedback_radius=3, feedback_share=128),
        dict(image_width=0, feedback=1, feedback_radius=2, feedback_share=10),
        dict(image_width=W, leaf_pct=1),
        dict(image_width=W, leaf_pct=100),
        dict(image_width=W, leaf_pct=37, block_threads=128, width=4),
        dict(image_width=W, quantized=0, feedback=1, width=4),
        dict(image_width=W, feedback=1, split_units=0, width=4),
        dict(image_width=W, feedback=1, split_max_units=100, width=4),      # 3600 units > 100: the launch is not split
        dict(image_width=W, feedback=1, split_max_units=3600, width=4),     # exactly at the limit: split
        dict(image_width=W, feedback=1, split_units=7, width=4),
        dict(image_width=W, feedback=1, split_units=65536, block_threads=256, width=4),   # capped: every unit is split
        dict(image_width=0, feedback=1, split_units=501, block_threads=128),
        dict(image_width=W, feedback=1, split_units=100, quantized=0, feedback_radius=0),
        dict(image_width=W, feedback=1, split_units=7, width=4, split_coop=0),       # quarter waves with one ray per lane
        dict(image_width=W, feedback=1, split_units=3600, width=4, split_coop=1),    # every unit as four cooperative quarter waves
        dict(image_width=0, feedback=1, split_units=300, width=4, split_coop=1, block_threads=256),
        dict(image_width=W, feedback=1, split_units=64, width=4, split_coop=1, lds_stack=8, leaf_pct=20),
        dict(image_width=W, feedback=1, feedback_period=1, width=4),
        dict(image_width=W, feedback=1, feedback_period=2, block_threads=128),
        dict(image_width=0, feedback=1, feedback_period=3, width=4),
        dict(image_width=W, feedback=1, feedback_period=1, width=4, side_stream=1),      # the order built on the library's side stream
        dict(image_width=W, feedback=1, feedback_period=3, width=4, side_stream=1, split_units=64),
    ]
    defaults = dict(width=2, quantized=1, block_threads=64, image_width=0, leaf_pct=50, persistent=0, split_units=32, split_coop=1,
                    feedback_period=4, persistent_waves=24, lds_stack=10, centre_first=1, feedback=0, feedback_radius=1, feedback_share=96,
                    side_stream=0, split_max_units=40960)
    for cfg in settings:
        for k, v in {**defaults, **cfg}.items():
            scene.set_param(k, v)
        for rep in range(9):      # from launch 2 on in an order sorted from earlier launches' costs, re-sorted every period
            out = launch(rays)
            assert np.array_equal(bits(out), bits(base)), (cfg, rep)
        assert np.array_equal(bits(scene.intersect(rays)), bits(base)), cfg      # and the host-pointer call
    # feedback state survives a change of shape and of rays between launches
    scene.set_param("feedback", 1)
    scene.set_param("image_width", W)
    for n in (len(rays), W * 8, len(rays), 1000, len(rays)):
        for rep in range(3):
            shifted = np.roll(rays[:n], rep * 17, axis=0)
            out = launch(shifted)
            assert np.array_equal(bits(out), bits(np.roll(base[:n], rep * 17, axis=0))), (n, rep)
    scene.close()


def test_cooperative_quarter_waves_are_exact(rt, oracle, gpu):
    """The costliest work units of a cost-sorted launch are traced by four waves of 16 rays each, FOUR LANES PER RAY (one
    child box or one leaf triangle per lane, quad-permute reductions).  Every record -- 64-byte and 16-byte, table rays and
    rays generated in the kernel, image tiles whose edges cut the 4x4 blocks, linear ray sets with a ragged end -- must equal
    the one-ray-per-lane launch bit for bit, and the oracle on a sample.  A tree too deep for the quads' LDS columns falls
    back to the one-ray-per-lane quarters (same bits again)."""
    import torch
    stream = torch.cuda.current_stream().cuda_stream
    for model, W, H in (("dragon.obj", 611, 347), ("teapot.obj", 256, 130)):
        mesh = rt.Mesh.load_obj(os.path.join(MODELS, model), "tolerant")
        bvh = rt.BVH(mesh, "fixed")
        origin = rt.frame_camera(bvh.root_box, W, H)
        rays = rt.primary_rays(W, H, origin)
        n = len(rays)
        d_rays = torch.from_numpy(rays).cuda()
        scene = rt.Scene(mesh, bvh, device=gpu, width=4)
        for k, v in dict(width=4, feedback=1, feedback_period=2, split_units=0).items():
            scene.set_param(k, v)
        want = scene.intersect(rays)
        compare(rt, oracle, mesh, rays, want, np.arange(0, n, 97))
        for hint, units, n_used in ((W, 1 << 16, n), (W, 40, n), (0, 1 << 16, n - 13), (0, 9, n)):
            scene.set_param("image_width", hint)
            scene.set_param("split_units", units)
            for coop, stack in ((1, 10), (0, 10), (1, 8)):
                scene.set_param("split_coop", coop)
                scene.set_param("lds_stack", stack)
                for rep in range(5):     # launches 0 and 1 build the order; from launch 2 on the head of it is split
                    full = torch.full((n, 16), -3.0, dtype=torch.float32, device="cuda")
                    small = torch.full((n, 4), -3.0, dtype=torch.float32, device="cuda")
                    scene.intersect_async(d_rays.data_ptr(), n_used, full.data_ptr(), small.data_ptr(), stream)
                    torch.cuda.synchronize()
                    got, got16 = full.cpu().numpy(), small.cpu().numpy()
                    assert np.array_equal(bits(got[:n_used]), bits(want[:n_used])), (model, hint, units, coop, stack, rep)
                    assert np.all(got[n_used:] == -3.0) and np.all(got16[n_used:] == -3.0)
                    assert np.array_equal(bits(got16[:n_used, 0]), bits(want[:n_used, 14])) and np.array_equal(bits(got16[:n_used, 1]), bits(want[:n_used, 15]))
                    assert np.array_equal(bits(got16[:n_used, 2:4]), bits(want[:n_used, 12:14]))
        # rays generated in the kernel, cooperative quarters included
        scene.set_param("image_width", W)
        scene.set_param("split_units", 1 << 16)
        scene.set_param("split_coop", 1)
        for rep in range(4):
            full = torch.full((n, 16), -3.0, dtype=torch.float32, device="cuda")
            scene.trace_primary(W, H, origin, full.data_ptr(), None, stream)
            torch.cuda.synchronize()
            assert np.array_equal(bits(full.cpu().numpy()), bits(want)), (model, rep)
        # the instrumented launch follows the same split and counts the same records as the host mirror of the traversal
        dev = scene.count_fetches_device(d_rays.data_ptr(), n)
        assert dev["rays"] == n and dev["rays_hit"] == int((want[:, 14].view(np.uint32) != rt.MISS_ID).sum())
        scene.close()


@pytest.mark.parametrize("W,H", [(640, 360), (203, 97), (8, 8), (1, 1)])
def test_rays_generated_in_the_kernel_equal_the_table(rt, gpu, W, H):
    """rt_trace_primary computes ray (i, j) of the reference's camera table inside the kernel; its records
    must be the bits that tracing the host-generated table gives, ragged image edges included."""
    import torch
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, "teapot.obj"), "reference")
    bvh = rt.BVH(mesh, "sah")
    scene = rt.Scene(mesh, bvh, device=gpu)
    origin = rt.frame_camera(bvh.root_box, max(W, 16), max(H, 16))
    table = scene.intersect(rt.primary_rays(W, H, origin))
    for feedback, split in ((0, 32), (1, 32), (1, 65536)):   # the last: every tile traced as four quarter waves
        scene.set_param("feedback", feedback)
        scene.set_param("split_units", split)
        for rep in range(4):
            full = torch.zeros((W * H, 16), dtype=torch.float32, device="cuda")
            compact = torch.zeros((W * H, 4), dtype=torch.float32, device="cuda")
            scene.trace_primary(W, H, origin, full.data_ptr(), compact.data_ptr())
            torch.cuda.synchronize()
            got = full.cpu().numpy()
            assert np.array_equal(bits(got), bits(table)), (feedback, split, rep)
            ids, t, u, v = rt.hit_fields(table)
            c = compact.cpu().numpy()
            assert np.array_equal(c[:, 0].view(np.uint32), ids) and np.array_equal(bits(c[:, 1]), bits(t))
    with pytest.raises(rt.RtError):
        scene.trace_primary(0, 4, origin, 1, None)
    scene.close()


@pytest.mark.parametrize("seed", range(12))
def test_random_triangle_soups_at_many_scales(rt, oracle, gpu, seed):
    """Seeded triangle soups (slivers, axis-aligned flats, duplicates, far-off and tiny clusters) at scene scales
    from 1e-3 to 1e5 and ray families that stress the conservative box test -- grazing, axis-parallel, origins
    inside the scene, origins far away: every builder, width and box encoding must return the oracle's bits."""
    rng = np.random.default_rng(1000 + seed)
    scale = f32(10.0 ** rng.uniform(-3, 5))
    offset = (rng.uniform(-1, 1, 3) * scale * f32(10.0 ** rng.uniform(-1, 2))).astype(f32)
    nt = int(rng.integers(40, 400))
    c = rng.uniform(-1, 1, (nt, 1, 3))
    tri = c + rng.normal(0, 1, (nt, 3, 3)) * 10.0 ** rng.uniform(-3, -0.5, (nt, 1, 1))
    tri[::7, :, int(rng.integers(3))] = tri[::7, :1, int(rng.integers(3))]        # axis-aligned flat triangles
    tri[5::11, 2] = tri[5::11, 0] + (tri[5::11, 1] - tri[5::11, 0]) * 0.5 + 1e-7  # slivers
    tri[3::13] = tri[2::13][: len(tri[3::13])]                                    # exact duplicates
    verts = np.ones((nt * 3, 4), f32)
    verts[:, :3] = (tri.reshape(-1, 3) * scale).astype(f32) + offset
    faces = np.zeros((nt, 8), np.uint32)
    faces[:, :3] = np.arange(nt * 3).reshape(nt, 3)
    mesh = rt.Mesh.from_arrays(verts, faces)
    lo, hi = verts[:, :3].min(0), verts[:, :3].max(0)
    n = 6000
    o = np.empty((n, 3), f32); d = np.empty((n, 3), f32)
    k = n // 4
    pick = rng.integers(0, nt, n)
    bary = rng.dirichlet((1, 1, 1), n).astype(f32)
    corners = verts[:, :3].reshape(nt, 3, 3)[pick]
    target = (corners * bary[:, :, None]).sum(1).astype(f32)                                 # a point on a triangle
    target[::5] += (rng.normal(0, 1, (len(target[::5]), 3)) * scale * 0.01).astype(f32)      # ... or just beside one
    o[:k] = (lo + (hi - lo) * rng.uniform(0, 1, (k, 3))).astype(f32)                       # inside the scene
    o[k:2 * k] = (lo - (hi - lo) * rng.uniform(1, 50, (k, 3))).astype(f32)                  # far away
    o[2 * k:3 * k] = target[2 * k:3 * k] + (rng.normal(0, 1, (k, 3)) * scale * 3).astype(f32)
    o[3 * k:] = target[3 * k:].copy()
    d[:] = target - o
    axis = rng.integers(0, 3, n - 3 * k)
    o[3 * k:, :] -= 0                                                                       # axis-parallel family:
    d[3 * k:] = 0
    d[np.arange(3 * k, n), axis] = rng.choice([-1, 1], n - 3 * k) * scale
    o[3 * k:] = target[3 * k:] - d[3 * k:] * f32(2.5)
    d[::17] *= f32(1e-6); d[1::17] *= f32(1e6)                                              # unnormalised directions
    rays = np.zeros((n, 16), f32)
    rays[:, 0:3] = o; rays[:, 3] = 1; rays[:, 4:7] = d
    oid, ot, ou, ov = oracle.closest_hit(mesh.verts, mesh.faces, rays)
    assert (oid != rt.MISS_ID).sum() > 50      # tiny scales lose most hits to the reference's |det| <= 1e-6 rejection
    for mode in ("fixed", "sah"):
        bvh = rt.BVH(mesh, mode)
        for width in (2, 4):
            scene = rt.Scene(mesh, bvh, device=gpu, width=width)
            for quant in (1, 0):
                scene.set_param("quantized", quant)
                ids, t, u, v = rt.hit_fields(scene.intersect(rays))
                bad = np.flatnonzero(ids != oid)
                assert len(bad) == 0, (mode, width, quant, float(scale), bad[:5], ids[bad[:5]], oid[bad[:5]])
                h = oid != rt.MISS_ID
                for a, b in ((t, ot), (u, ou), (v, ov)):
                    assert np.array_equal(bits(a[h]), bits(b[h])), (mode, width, quant)
            scene.close()


def test_dispatch_order_kernels_build_the_stable_bin_sort(rt, gpu):
    """The costliest-first dispatch order (count, scan and scatter kernels) against numpy: units sorted by cost bin (cost / 8
    capped at 31, costliest first), index order inside a bin -- a permutation of all units, and the cost table is left zeroed.
    Sizes around the kernels' chunking (1024 keys per chunk, 2048 bin counters per scan tile) and the benchmark shapes
    (32 400 units at 1080p, 129 600 at 4K)."""
    rng = np.random.default_rng(11)
    for n in (1, 63, 1024, 4096, 4097, 32400, 129600, 300000, 700000):   # the last one takes the separate scan launch
        costs = np.where(rng.random(n) < 0.5, rng.integers(0, 8, n), rng.integers(0, 400, n)).astype(np.uint32)
        bins = 31 - np.minimum(costs >> 3, 31)
        want = np.argsort(bins, kind="stable").astype(np.uint32)
        order = np.full(n, 0xFFFFFFFF, np.uint32)
        after = np.full(n, 7, np.uint32)
        rt._check(rt.lib.rt_debug_build_order(gpu, rt._ptr(costs), n, rt._ptr(order), rt._ptr(after)))
        assert np.array_equal(order, want), n
        assert not after.any(), n
