"""The CPU oracle: known-answer tests from the reference (cpu_raycast/raycast.cpp:8-21 `main0`)
and bit-equality with the reference's own core/ray_tri.cpp / core/raymath.cpp where
oracle/_ref has been built (it is built wherever /root/reference exists)."""
import os

import numpy as np
import pytest

from conftest import MODELS, bits


def _ray(o, d):
    return np.array([*o, 0, *d, 0], np.float32)


def _tri(a, b, c):
    return np.array([[*a, 0], [*b, 0], [*c, 0]], np.float32)


KAT_TRI = _tri((0, 1, 1), (0, -1, 0), (0, 1, -1))


def test_main0_known_answer(oracle):
    hit, t, uv = oracle.tri_batch(KAT_TRI[None], _ray((-1, 0, 0), (1, 0, 0))[None])
    assert hit[0] == 1
    assert bits(t)[0] == 0x3F800000 and bits(uv)[0, 0] == 0x3F000000 and bits(uv)[0, 1] == 0x3E800000


def test_two_sided_parallel_behind_and_negative_zero(oracle):
    rays = np.stack([
        _ray((1, 0, 0), (-1, 0, 0)),      # from the back side: still a hit
        _ray((-1, 0, 0), (0, 1, 0)),      # parallel to the plane
        _ray((-1, 0, 0), (-1, 0, 0)),     # triangle behind the origin
        _ray((-1, 0, 0.5), (1, 0, 0)),    # on the edge: v == -0.0 passes the `< 0.f` test
    ])
    hit, t, uv = oracle.tri_batch(np.repeat(KAT_TRI[None], 4, 0), rays)
    assert hit.tolist() == [1, 0, 0, 1]
    assert t[0] == 1.0 and uv[0, 0] == 0.5 and uv[0, 1] == 0.25
    assert uv[3, 0] == 0.5 and bits(uv)[3, 1] in (0x80000000, 0x00000000)


def test_epsilon_is_a_double_literal(oracle):
    # |det| <= (float)1e-6 is "near zero"; the next float up is not.  Triangle in the plane x=0 with
    # e1 x e2 chosen so det = dir.x * area-ish value we can dial exactly.
    eps = np.float32(1e-6)
    up = np.nextafter(eps, np.float32(1))
    for det, expect in ((eps, 0), (up, 1), (-eps, 0), (-up, 1)):
        # tri a=(0,0,0), b=(0,1,0), c=(0,0,1): P = d x e2 = (dy*1-0, 0-dx*1, 0) -> det = e1.P = -dx ... use dx = -det
        tri = _tri((0, 0, 0), (0, 1, 0), (0, 0, 1))
        ray = _ray((det * 4, 0.25, 0.25), (-det, 0, 0))
        hit, t, uv = oracle.tri_batch(tri[None], ray[None])
        assert hit[0] == expect, (det, hit, t)
    # t must be strictly greater than the double 1e-6
    tri = _tri((0, 0, 0), (0, 1, 0), (0, 0, 1))
    for t0, expect in ((eps, 0), (up, 1)):
        hit, t, _ = oracle.tri_batch(tri[None], _ray((t0, 0.25, 0.25), (-1, 0, 0))[None])
        assert hit[0] == expect


def random_pairs(rng, n):
    """Triangles and rays with a healthy share of hits, grazing rays and degenerate inputs."""
    tri = rng.uniform(-1, 1, (n, 3, 4)).astype(np.float32)
    tri[..., 3] = 1
    target = (tri[:, 0, :3] * 0.3 + tri[:, 1, :3] * 0.3 + tri[:, 2, :3] * 0.4)
    k = n // 4
    w = rng.uniform(-0.2, 1.2, (k, 2)).astype(np.float32)     # some just outside the edges
    target[:k] = tri[:k, 0, :3] + w[:, :1] * (tri[:k, 1, :3] - tri[:k, 0, :3]) + w[:, 1:] * (tri[:k, 2, :3] - tri[:k, 0, :3])
    target[k:k + k // 4] = tri[k:k + k // 4, 1, :3]            # aimed exactly at a vertex
    org = rng.uniform(-4, 4, (n, 3)).astype(np.float32)
    d = target - org
    d[n // 2:] /= np.linalg.norm(d[n // 2:], axis=1, keepdims=True).astype(np.float32)
    ray = np.zeros((n, 8), np.float32)
    ray[:, 0:3] = org
    ray[:, 4:7] = d
    # sprinkle degenerate cases
    tri[-50:-40, 2] = tri[-50:-40, 1]          # zero-area triangles
    ray[-40:-30, 4:7] = 0                      # zero directions
    ray[-30:-20, 4] = np.nan                   # NaN direction
    ray[-20:-10, 0:3] = tri[-20:-10, 0, :3]    # origin on a vertex
    tri[-10:] *= np.float32(1e-4)              # tiny triangles: |det| near 1e-6
    return tri, ray


def test_oracle_matches_reference_ray_tri_bitwise(oracle):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    rng = np.random.default_rng(20241111)
    total = 0
    for _ in range(5):
        tri, ray = random_pairs(rng, 2_000_000)
        h0, t0, uv0 = oracle.tri_batch(tri, ray)
        h1, t1, uv1 = oracle.ref_tri_batch(tri, ray)
        assert np.array_equal(h0, h1)
        m = h0 == 1
        assert np.array_equal(bits(t0)[m], bits(t1)[m]) and np.array_equal(bits(uv0)[m], bits(uv1)[m])
        total += int(m.sum())
    assert total > 1_000_000  # the sweep really exercises the accept path


def test_oracle_aabb_and_normalize_match_reference(oracle):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built")
    import ctypes as C
    rng = np.random.default_rng(7)
    n = 500_000
    lo = rng.uniform(-2, 1, (n, 3)).astype(np.float32)
    box = np.zeros((n, 8), np.float32)
    box[:, 0:3] = lo
    box[:, 4:7] = lo + rng.uniform(0, 2, (n, 3)).astype(np.float32)
    ray = np.zeros((n, 8), np.float32)
    ray[:, 0:3] = rng.uniform(-4, 4, (n, 3))
    ray[:, 4:7] = rng.normal(size=(n, 3))
    ray[:1000, 4] = 0      # axis-parallel rays: 1/0 = inf
    ray[1000:2000, 0] = box[1000:2000, 0]  # origin exactly on a slab plane
    assert np.array_equal(oracle.aabb_batch(box, ray), oracle.ref_aabb_batch(box, ray))
    # normalize via the primary-ray table vs the reference's normalize
    W, H = 37, 23
    tab = oracle.primary_rays(W, H, (0, 0, -5))
    r = oracle.ref()
    for (i, j) in ((0, 0), (36, 22), (18, 11), (5, 20)):
        v = np.array([i - W / 2.0, j - H / 2.0, 5.0 * W, 0], np.float32)
        out = np.zeros(4, np.float32)
        r.ref_normalize(C.c_void_p(v.ctypes.data), C.c_void_p(out.ctypes.data))
        assert np.array_equal(bits(tab[i + j * W, 4:7]), bits(out[:3]))


def test_tree_walk_returns_the_brute_force_answer_bit_for_bit(rt, oracle):
    """The oracle's closest hit through the reference-format tree (every builder, coherent and incoherent rays, and the
    coplanar tie) equals its brute-force loop in ids and in the bits of t, u, v: the accelerated CPU figure of the
    benchmark and the full-size GPU parity tests lean on it."""
    for model, W, H in (("dragon.obj", 96, 54), ("teapot.obj", 128, 72), ("cube.obj", 48, 48)):
        mesh = rt.Mesh.load_obj(os.path.join(MODELS, model), "tolerant")
        for mode in ("fixed", "sah") + (("reference",) if model != "dragon.obj" else ()):
            bvh = rt.BVH(mesh, mode)
            rays = rt.primary_rays(W, H, rt.frame_camera(bvh.root_box, W, H))
            if model == "teapot.obj":
                rays = np.concatenate([rays, rt.ao_rays(mesh, bvh.root_box, 3, 0, 4000)])
            a = oracle.closest_hit(mesh.verts, mesh.faces, rays, threads=8)
            b = oracle.closest_hit_bvh(mesh.verts, mesh.faces, bvh.nodes, rays, threads=8)
            assert (a[0] != rt.MISS_ID).sum() > 100
            for x, y in zip(a, b):
                assert np.array_equal(bits(x), bits(y)), (model, mode)
    # the coplanar pair of the tie test below, through a hand-made two-leaf tree
    verts = np.array([[0, 0, 0, 1], [1, 0, 0, 1], [1, 1, 0, 1], [0, 1, 0, 1]], np.float32)
    faces = np.zeros((2, 8), np.uint32)
    faces[0, :3] = (0, 1, 2)
    faces[1, :3] = (0, 2, 3)
    mesh = rt.Mesh.from_arrays(verts, faces)
    bvh = rt.BVH(mesh, "fixed")
    rays = np.zeros((2, 16), np.float32)
    rays[:, 0:3] = [(0.5, 0.5, -1), (0.25, 0.75, -1)]
    rays[:, 6] = 1
    ids, t, _, _ = oracle.closest_hit_bvh(verts, faces, bvh.nodes, rays, threads=1)
    assert ids.tolist() == [0, 1] and t.tolist() == [1.0, 1.0]


def test_brute_force_tie_goes_to_lowest_index_and_second_t(oracle):
    # two coplanar triangles sharing the diagonal of a unit square; a ray through the diagonal ties
    verts = np.array([[0, 0, 0, 1], [1, 0, 0, 1], [1, 1, 0, 1], [0, 1, 0, 1]], np.float32)
    faces = np.zeros((2, 8), np.uint32)
    faces[0, :3] = (0, 1, 2)
    faces[1, :3] = (0, 2, 3)
    rays = np.zeros((2, 16), np.float32)
    rays[:, 0:3] = [(0.5, 0.5, -1), (0.25, 0.75, -1)]
    rays[:, 6] = 1
    ids, t, u, v, second = oracle.closest_hit(verts, faces, rays, threads=1, want_second=True)
    assert ids.tolist() == [0, 1] and t.tolist() == [1.0, 1.0]
    assert second[0] == 1.0 and second[1] > 1e30
    if oracle.have_ref():
        rid, rtt, ru, rv = oracle.ref_closest_hit(verts, faces, rays)
        assert rid.tolist() == [0, 1] and np.array_equal(bits(rtt), bits(t))


def test_camera_matrix_is_the_reference_frame_loops(rt, oracle):
    """rt_camera_matrix against core/Affine.h's own rotate3 / multiply (header-only templates instantiated in
    oracle/_ref) driven as render_client/img_main.cpp:627-660 drives them: all 16 floats, bit for bit."""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    rng = np.random.default_rng(11)
    cases = [(0.0, 0.0, (0, 0, -100)), (0.3, -1.2, (0, 0, -100)), (6.0, 2.5, (1.5, -2, 40))]
    cases += [(float(a), float(b), tuple(rng.uniform(-50, 50, 3))) for a, b in rng.uniform(-7, 7, (200, 2))]
    for rx, ry, tr in cases:
        got, want = rt.camera_matrix(rx, ry, tr), oracle.ref_camera_matrix(rx, ry, tr)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (rx, ry, tr)
    # the out[1][0] = 0 slip of rotate3 (core/Affine.h:180) is part of the matrices the reference uploads
    m = rt.camera_matrix(0.0, 0.7, (0, 0, 0))
    assert m[0, 2] != 0 and abs(m[0, 0] - np.cos(np.float32(0.7))) < 1e-6


def test_bmp_bytes_equal_the_reference_writer(rt, oracle, tmp_path):
    """rt_write_bmp32 against cpu_raycast/bmp.c's printImage (compiled where it lies), byte for byte."""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    rng = np.random.default_rng(3)
    for (w, h) in ((1, 1), (7, 5), (64, 48), (360, 20)):
        img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        a, b = str(tmp_path / f"a_{w}x{h}.bmp"), str(tmp_path / f"b_{w}x{h}.bmp")
        rt.write_bmp32(a, img)
        oracle.ref_print_image(img, b)
        assert open(a, "rb").read() == open(b, "rb").read(), (w, h)
