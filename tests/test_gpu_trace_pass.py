"""The full `trace` step (render_client/tracekernel.ck:192-414) through rt_trace_pass: camera
transform (op 0), reflection regeneration (op 1), output record and RGBA32F pixel.

Ray setup, output record and pixel are compared BIT FOR BIT with oracle.c's restatement of
render_client/tracekernel.ck:224-265, :282-286, :373-413 (unfused float, sums left to right as written
there).  That restatement is parity-unpinned for the interpolated outputs and the colour: the kernel is
OpenCL C, cannot run here, and leaves the rounding of dot() to the implementation; ids and t/u/v are pinned
through the brute-force oracle as everywhere else."""
import os

import numpy as np
import pytest

from conftest import MODELS, bits, framed_primary

pytestmark = pytest.mark.gpu
REL = 1e-5
f32 = np.float32


def run_pass(scene, rays_in, cam, op):
    import torch
    d_in = torch.from_numpy(np.ascontiguousarray(rays_in)).cuda()
    d_out = torch.zeros_like(d_in)
    d_rgba = torch.full((len(rays_in), 4), -1.0, dtype=torch.float32, device="cuda")
    scene.trace_pass(d_in.data_ptr(), d_out.data_ptr(), d_rgba.data_ptr(), len(rays_in), cam, op)
    torch.cuda.synchronize()
    return d_out.cpu().numpy(), d_rgba.cpu().numpy()


@pytest.mark.parametrize("width", [2, 4])
def test_primary_then_reflection_pass(rt, oracle, gpu, width):
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, "cube_mirror_test.obj"), "reference")
    bvh = rt.BVH(mesh, "fixed")
    scene = rt.Scene(mesh, bvh, device=gpu, width=width)
    W, H = 160, 120
    # camera table in camera space, placed by the matrix: a rotation about y plus a translation
    c, s = f32(np.cos(2.6)), f32(np.sin(2.6))   # from the +z side, where the cube sits above the mirror slab
    box = bvh.root_box
    centre = 0.5 * (box[0, :3] + box[1, :3])
    dist = f32(8.0) * f32(np.abs(box[1, :3] - box[0, :3]).max())
    cam = np.array([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1]], f32)
    cam[:3, 3] = centre - dist * np.array([s, 0, c], f32)   # eye sits `dist` behind the centre along the view axis
    rays = rt.primary_rays(W, H, (0, 0, 0))
    out0, rgba0 = run_pass(scene, rays, cam, 0)

    # ---- op 0: the rays the kernel must have traced, then the whole output record and pixel ----------------
    world = oracle.trace_setup(rays, cam, 0)
    oid, ot, ou, ov = oracle.closest_hit(mesh.verts, mesh.faces, world)
    ids, t, u, v = rt.hit_fields(out0)
    assert np.array_equal(ids, oid)
    h = oid != rt.MISS_ID
    assert 0.05 < h.mean() < 0.95
    for a, b in ((t, ot), (u, ou), (v, ov)):
        assert np.array_equal(bits(a[h]), bits(b[h]))
    want_rec, want_rgba, defined = oracle.trace_epilogue(mesh, rays, cam, 0, oid, ot, ou, ov)
    assert defined.all()                                   # every face of this model carries vertex normals
    assert np.array_equal(bits(out0), bits(want_rec))      # hit point, normal, direction, uv, id, t; miss defaults
    assert np.array_equal(bits(rgba0), bits(want_rgba))    # |n.d| * material.color, alpha = material.Rf; (0,0,0,0) on a miss
    assert np.all(rgba0[~h] == 0) and rgba0[h, :3].max() > 0

    # ---- op 1: reflect the pass-0 records --------------------------------------------
    out1, rgba1 = run_pass(scene, out0, cam, 1)
    rec = oracle.trace_setup(out0, cam, 1)
    rid, rtt, ru, rv = oracle.closest_hit(mesh.verts, mesh.faces, rec)
    ids1, t1, u1, v1 = rt.hit_fields(out1)
    live = h
    assert np.array_equal(ids1[live], rid[live])
    assert np.all(ids1[~live] == rt.MISS_ID) and np.all(rgba1[~live] == 0)   # terminated rays stay terminated
    hh = live & (rid != rt.MISS_ID)
    assert hh.sum() > 0
    for a, b in ((t1, rtt), (u1, ru), (v1, rv)):
        assert np.array_equal(bits(a[hh]), bits(b[hh]))
    want_rec1, want_rgba1, defined1 = oracle.trace_epilogue(mesh, out0, cam, 1, rid, rtt, ru, rv)
    assert defined1.all()
    assert np.array_equal(bits(out1[live]), bits(want_rec1[live]))
    assert np.array_equal(bits(rgba1[live]), bits(want_rgba1[live]))
    # a terminated record passes through: origin, direction and normal as they came in, id = miss
    assert np.array_equal(bits(out1[~live, 0:12]), bits(out0[~live, 0:12]))


def test_frame_loop_camera_matrices_drive_the_primary_pass(rt, oracle, gpu):
    """op 0 fed the matrices the reference's frame loop produces (rt_camera_matrix = core/Affine.h rotate3 /
    multiply as render_client/img_main.cpp:627-660 uses them): a camera-space ray table, the eye placed by the
    translation, the view turned by the cursor angles.  Every frame's ids and t/u/v equal the oracle's."""
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, "teapot.obj"), "reference")
    bvh = rt.BVH(mesh, "fixed")
    scene = rt.Scene(mesh, bvh, device=gpu, width=4)
    W, H = 160, 100
    scene.set_param("image_width", W)
    rays = rt.primary_rays(W, H, (0, 0, 0))
    eye = rt.frame_camera(bvh.root_box, W, H)
    seen = set()
    for rx, ry in ((0.0, 0.0), (0.02, -0.05), (-0.04, 0.08), (0.0, 0.3)):
        cam = rt.camera_matrix(rx, ry, eye)
        out, rgba = run_pass(scene, rays, cam, 0)
        world = oracle.trace_setup(rays, cam, 0)
        oid, ot, ou, ov = oracle.closest_hit(mesh.verts, mesh.faces, world)
        ids, t, u, v = rt.hit_fields(out)
        assert np.array_equal(ids, oid), (rx, ry)
        h = oid != rt.MISS_ID
        for a, b in ((t, ot), (u, ou), (v, ov)):
            assert np.array_equal(bits(a[h]), bits(b[h]))
        want_rec, want_rgba, defined = oracle.trace_epilogue(mesh, rays, cam, 0, oid, ot, ou, ov)
        assert np.array_equal(bits(out[defined]), bits(want_rec[defined]))
        seen.add(int(h.sum()))
    assert len(seen) >= 3 and max(seen) > 1000       # the view really changed between frames


def test_identity_camera_equals_plain_intersection(rt, gpu):
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, "teapot.obj"), "reference")
    bvh = rt.BVH(mesh, "fixed")
    scene = rt.Scene(mesh, bvh, device=gpu)
    rays = framed_primary(rt, bvh, 128, 80)
    out, rgba = run_pass(scene, rays, np.eye(4, dtype=f32), 0)
    plain = scene.intersect(rays)
    assert np.array_equal(bits(out), bits(plain))
    h = plain[:, 14].view(np.uint32) != rt.MISS_ID
    # teapot has no usemtl: every face uses material 0, the all-zero sky material
    assert np.all(rgba[h, :3] == 0)


def test_trace_pass_argument_errors(rt, gpu):
    import ctypes as C
    import torch
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, "cube.obj"), "tolerant")
    scene = rt.Scene(mesh, rt.BVH(mesh, "fixed"), device=gpu)
    buf = torch.zeros((64, 16), dtype=torch.float32, device="cuda")
    cam = (C.c_float * 16)(*np.eye(4, dtype=f32).reshape(-1))
    p = C.c_void_p(buf.data_ptr())
    assert rt.lib.rt_trace_pass(scene._h, p, p, None, 64, cam, 0, None) == -1          # aliasing in/out
    assert rt.lib.rt_trace_pass(scene._h, p, None, None, 64, cam, 0, None) == -1       # no output at all
    assert rt.lib.rt_trace_pass(scene._h, p, p, None, 64, cam, 2, None) == -1          # unknown operation
    assert rt.lib.rt_trace_pass(scene._h, p, None, None, 0, cam, 0, None) == 0         # empty launch is fine


@pytest.mark.parametrize("model", ["cube_mirror_test.obj", "teapot.obj"])
def test_five_pass_bounce_chain_full_size(rt, oracle, gpu, model):
    """1920x1080 primary pass then four reflection passes, each fed by the previous pass's records
    (render_client/tracekernel.ck:227-233 regeneration; rays that missed stay terminated).  Every pass is
    checked against the oracle on a sub-grid, starting from the records the GPU itself produced."""
    import torch
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, model), "reference")
    bvh = rt.BVH(mesh, "fixed")
    scene = rt.Scene(mesh, bvh, device=gpu)
    W, H = 1920, 1080
    scene.set_param("image_width", W)
    scene.set_param("feedback", 1)
    sel = (np.arange(0, H, 24)[:, None] * W + np.arange(0, W, 24)[None, :]).reshape(-1)
    if model == "cube_mirror_test.obj":
        # from the +z side, where the cube sits above the mirror slab (same pose as the small test above)
        c, s = f32(np.cos(2.6)), f32(np.sin(2.6))
        box = bvh.root_box
        centre = 0.5 * (box[0, :3] + box[1, :3])
        dist = f32(8.0) * f32(np.abs(box[1, :3] - box[0, :3]).max())
        cam = np.array([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1]], f32)
        cam[:3, 3] = centre - dist * np.array([s, 0, c], f32)
        rays = rt.primary_rays(W, H, (0, 0, 0))
    else:
        cam = np.eye(4, dtype=f32)
        rays = framed_primary(rt, bvh, W, H)
    bufs = [torch.from_numpy(rays).cuda(), torch.zeros((len(rays), 16), dtype=torch.float32, device="cuda")]
    rgba = torch.zeros((len(rays), 4), dtype=torch.float32, device="cuda")
    prev = rays
    alive = np.ones(len(rays), bool)
    hits_per_pass = []
    for k in range(5):
        scene.trace_pass(bufs[k & 1].data_ptr(), bufs[(k + 1) & 1].data_ptr(), rgba.data_ptr(), len(rays), cam, 0 if k == 0 else 1)
        torch.cuda.synchronize()
        out = bufs[(k + 1) & 1].cpu().numpy()
        ids, t, u, v = rt.hit_fields(out)
        want_rays = oracle.trace_setup(prev[sel], cam, 0 if k == 0 else 1)
        oid, ot, ou, ov = oracle.closest_hit(mesh.verts, mesh.faces, want_rays)
        live = alive[sel]
        assert np.array_equal(ids[sel][live], oid[live]), f"pass {k}"
        assert np.all(ids[sel][~live] == rt.MISS_ID), f"pass {k}: a terminated ray came back"
        hh = live & (oid != rt.MISS_ID)
        for a, b in ((t, ot), (u, ou), (v, ov)):
            assert np.array_equal(bits(a[sel][hh]), bits(b[hh])), f"pass {k}"
        want_rec, want_rgba, defined = oracle.trace_epilogue(mesh, prev[sel], cam, 0 if k == 0 else 1, oid, ot, ou, ov)
        ok = live & defined
        assert np.array_equal(bits(out[sel][ok]), bits(want_rec[ok])), f"pass {k}: output record"
        assert np.array_equal(bits(rgba.cpu().numpy()[sel][ok]), bits(want_rgba[ok])), f"pass {k}: pixel"
        alive &= ids != rt.MISS_ID
        hits_per_pass.append(int(alive.sum()))
        prev = out
    assert hits_per_pass[0] > 0 and all(a >= b for a, b in zip(hits_per_pass, hits_per_pass[1:]))
    if model == "cube_mirror_test.obj":
        assert hits_per_pass[1] > 0      # the mirror slab really reflects onto the cube


def test_headless_driver_frame_matches_the_library_path(rt, gpu, tmp_path):
    """tools/rt_headless (the stand-in for the reference's windowed client, written against the C header only)
    renders primary + one bounce and writes a BMP; the same frame composed here from rt_trace_pass layers with
    the reference's blend (render_client/img_main.cpp:434,512-526) must give the same bytes."""
    import subprocess
    import torch
    exe = os.path.join(os.path.dirname(MODELS.rstrip("/")), "..", "tools", "rt_headless")
    exe = os.path.normpath(exe)
    if not os.path.exists(exe):
        pytest.fail(f"{exe} is missing: run make")
    W, H = 160, 96
    model = os.path.join(MODELS, "cube_mirror_test.obj")
    out = str(tmp_path / "frame.bmp")
    r = subprocess.run([exe, "--model", model, "--width", str(W), "--height", str(H), "--loader", "reference", "--bvh", "sah",
                        "--bounces", "1", "--device", str(gpu), "--out", out], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Triangles: 24" in r.stdout
    data = open(out, "rb").read()
    assert data[:2] == b"BM" and len(data) == 58 + W * H * 4
    got = np.frombuffer(data[58:], np.uint8).reshape(H, W, 4)[::-1]          # rows are stored bottom-up

    mesh = rt.Mesh.load_obj(model, "reference")
    bvh = rt.BVH(mesh, "sah")
    scene = rt.Scene(mesh, bvh, device=gpu)
    rays = framed_primary(rt, bvh, W, H)
    out0, rgba0 = run_pass(scene, rays, np.eye(4, dtype=f32), 0)
    out1, rgba1 = run_pass(scene, out0, np.eye(4, dtype=f32), 1)
    frame = rgba0.copy()
    sa, da = rgba1[:, 3:4], frame[:, 3:4].copy()
    frame = rgba1 * sa + frame * (f32(1) - da)
    q = np.clip(frame * f32(255) + f32(0.5), 0, 255).astype(np.uint8).reshape(H, W, 4)
    want = np.stack([q[..., 2], q[..., 1], q[..., 0], np.full((H, W), 255, np.uint8)], axis=-1)
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1            # float blend order may differ in the last bit
    assert (got[..., :3].sum(axis=-1) > 0).mean() > 0.02                      # the frame shows something
    # the primary pass alone, plainly and through the sharded-frame object (one rank, 32 x 32 squares): the same file
    frames = []
    for extra in ([], ["--tiled", "32"]):
        path = str(tmp_path / f"primary{len(extra)}.bmp")
        r = subprocess.run([exe, "--model", model, "--width", str(W), "--height", str(H), "--loader", "reference", "--bvh", "sah",
                            "--bounces", "0", "--device", str(gpu), "--out", path] + extra, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        frames.append(open(path, "rb").read())
    assert frames[0] == frames[1]
    lit = np.frombuffer(frames[0][58:], np.uint8).reshape(H, W, 4)[..., :3].any(axis=-1)
    assert 0.02 < lit.mean() < 0.98                                          # a silhouette: a misplaced square would show
