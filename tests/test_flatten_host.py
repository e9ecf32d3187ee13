"""Host-side checks that need no GPU: the oracle reproduces the committed golden vectors, the BVH
re-flattening validates its input, and the host traversal of both flattened layouts reaches the same
hits the brute-force oracle finds (so no triangle was lost in flattening or widening)."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from conftest import MODELS, ROOT, bits, framed_primary

GOLDEN = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz")))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_reproduces_golden(rt, oracle, path):
    g = np.load(path)
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, str(g["model"])), str(g["loader"]))
    rays = rt.primary_rays(int(g["width"]), int(g["height"]), g["origin"])
    assert np.array_equal(bits(rays), bits(oracle.primary_rays(int(g["width"]), int(g["height"]), g["origin"])))
    ids, t, u, v = oracle.closest_hit(mesh.verts, mesh.faces, rays)
    assert np.array_equal(ids, g["ids"])
    for mine, gold in ((t, g["t"]), (u, g["u"]), (v, g["v"])):
        assert np.array_equal(bits(mine), bits(gold))


def test_there_are_golden_files():
    assert len(GOLDEN) >= 4


def _host_scene(rt, mesh, nodes, width=2):
    class HostScene(rt.Scene):
        pass
    return rt.Scene(mesh, nodes, device=-1, width=width)


@pytest.mark.parametrize("model,W,H", [("cube.obj", 96, 96), ("teapot.obj", 96, 64), ("bunny.obj", 48, 40)])
@pytest.mark.parametrize("mode", ["fixed", "reference"])
def test_flattened_layouts_lose_no_hits(rt, oracle, model, W, H, mode):
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, model), "tolerant")
    bvh = rt.BVH(mesh, mode)
    rays = framed_primary(rt, bvh, W, H)
    ids, *_ = oracle.closest_hit(mesh.verts, mesh.faces, rays)
    want = int((ids != rt.MISS_ID).sum())
    assert want > 0
    for width in (2, 4):
        s = _host_scene(rt, mesh, bvh, width)
        for quant in (1, 0):
            s.set_param("quantized", quant)
            c = s.count_fetches(rays)
            assert c["rays"] == len(rays) and c["rays_hit"] == want
            info = s.info()
            assert info["leaf_tris"] == len(mesh.faces) and info["nodes4"] <= info["nodes2"]
            assert (info["node2_bytes"], info["node4_bytes"], info["tri_bytes"]) == ((32, 64, 48) if quant else (64, 128, 48))
            assert c["nodes"] >= c["rays"]      # every ray reads at least the root record
            if quant == 0:
                exact = c["nodes"]
            else:
                loose = c["nodes"]
        assert exact <= loose <= 1.10 * exact   # the 16-bit grid inflates boxes by well under a cell's worth of work
        with pytest.raises(rt.RtError, match="without a device"):
            s.intersect(rays[:4])
    # the fixed tree is what makes the traversal cheap (core/CentroidBVH.cpp:65)
    if model == "bunny.obj" and mode == "reference":
        fixed = _host_scene(rt, mesh, rt.BVH(mesh, "fixed")).count_fetches(rays)
        assert c["nodes"] > 20 * fixed["nodes"]


def test_flatten_rejects_malformed_trees(rt):
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, "teapot.obj"), "tolerant")
    good = rt.BVH(mesh, "fixed").nodes

    def expect(nodes, msg):
        with pytest.raises(rt.RtError, match=msg):
            rt.Scene(mesh, nodes, device=-1)

    bad = good.copy(); bad[0, 16] = 1
    expect(bad, "root must be an inner node")
    bad = good.copy(); bad[0, 5] = len(good) + 7
    expect(bad, "child index out of range")
    bad = good.copy(); bad[0, 5] = bad[0, 4]
    expect(bad, "not a tree")
    leaf = int(np.flatnonzero(good[:, 16] == 1)[0])
    bad = good.copy(); bad[leaf, 4] = 5
    expect(bad, "more than 4 triangles")
    bad = good.copy(); bad[leaf, 0] = len(mesh.faces)
    expect(bad, "triangle index out of range")
    bad = good.copy(); bad[leaf, 16] = 7
    expect(bad, "unknown node type")
    expect(good[:2], "root with two children")
    # boxes that do not hold their triangles: a shrunken leaf box, a shrunken inner box, a root box that is too small
    fview = lambda a: a.view(np.float32)
    bad = good.copy(); fview(bad)[leaf, 12:15] = fview(bad)[leaf, 8:11]            # leaf box max := min
    expect(bad, "does not contain triangle")
    inner = int(np.flatnonzero(good[:, 16] == 0)[1])
    bad = good.copy(); fview(bad)[inner, 12:15] = fview(bad)[inner, 8:11]
    expect(bad, f"box of node {inner} does not contain")
    bad = good.copy(); fview(bad)[0, 8] = fview(bad)[0, 8] + 0.25 * (fview(bad)[0, 12] - fview(bad)[0, 8])
    expect(bad, "box of node 0 does not contain")
    m2 = rt.Mesh.from_arrays(mesh.verts[:10], mesh.faces)
    expect_nodes = good
    with pytest.raises(rt.RtError, match="vertex index out of range"):
        rt.Scene(m2, expect_nodes, device=-1)


def test_ray_generators(rt, oracle):
    box = np.array([[-1, -2, 0, 1], [3, 2, 5, 1]], np.float32)
    o = rt.frame_camera(box, 160, 90)
    assert o[0] == 1.0 and o[1] == 0.0 and o[2] < 0
    rays = rt.primary_rays(160, 90, o)
    assert np.array_equal(bits(rays), bits(oracle.primary_rays(160, 90, o)))
    # record defaults of render_client/img_main.cpp:261-265
    assert np.all(rays[:, 3] == 1) and np.all(rays[:, 8:12] == np.array([0, 0, 1, 0], np.float32))
    assert np.all(rays[:, 14].view(np.uint32) == 0xFFFFFFFF) and np.all(rays[:, 12:14] == 0)
    # the frustum really frames the box: corner rays pass outside it at its near face
    d = rays[0, 4:7]; tz = (box[0, 2] - o[2]) / d[2]
    assert o[0] + tz * d[0] < box[0, 0] and o[1] + tz * d[1] < box[0, 1]
    mesh = rt.Mesh.load_obj(os.path.join(MODELS, "teapot.obj"), "tolerant")
    bvh = rt.BVH(mesh, "fixed")
    a = rt.ao_rays(mesh, bvh.root_box, 1234, 0, 5000)
    b = rt.ao_rays(mesh, bvh.root_box, 1234, 1000, 1000)
    assert np.array_equal(bits(a[1000:2000]), bits(b))                      # numbered, order-independent
    assert np.allclose(np.linalg.norm(a[:, 4:7], axis=1), 1, atol=1e-5)
    assert np.all(np.einsum("ij,ij->i", a[:, 4:7], a[:, 8:11]) > -1e-4)     # in the normal's hemisphere
    assert not np.array_equal(a[:100], rt.ao_rays(mesh, bvh.root_box, 99, 0, 100))


def test_bmp_writer_layout(rt, tmp_path):
    W, H = 5, 3
    px = np.arange(W * H * 4, dtype=np.uint8).reshape(H, W, 4)
    p = str(tmp_path / "x.bmp")
    rt.write_bmp32(p, px)
    raw = open(p, "rb").read()
    assert len(raw) == 58 + W * H * 4 and raw[:2] == b"BM"
    u32 = lambda o: int.from_bytes(raw[o:o + 4], "little")
    u16 = lambda o: int.from_bytes(raw[o:o + 2], "little")
    assert (u32(2), u32(6), u32(10), u32(14), u32(18), u32(22)) == (len(raw), 0, 58, 40, W, H)
    assert (u16(26), u16(28), u32(30), u32(34), u32(38), u32(42), u32(46), u32(50), u32(54)) == \
        (1, 32, 0, 0, 0xFFFFFF, 0xFFFFFF, 0, 0, 0)
    assert raw[58:58 + W * 4] == px[H - 1].tobytes() and raw[-W * 4:] == px[0].tobytes()   # bottom-up rows
    q = str(tmp_path / "x.pfm")
    rt.write_pfm(q, np.ones((H, W, 4), np.float32))
    assert open(q, "rb").read().startswith(b"PF\n5 3\n-1.0\n")


def test_host_code_is_clean_under_address_and_ub_sanitizers():
    """Loader, the three BVH builders, re-flattening, fetch counting, ray generators and image writers, driven by
    tools/host_selftest.cpp in a build with -fsanitize=address,undefined (`make sanitize-host`)."""
    import shutil
    import subprocess
    if shutil.which("g++") is None or shutil.which("make") is None:
        pytest.skip("no host toolchain")
    r = subprocess.run(["make", "sanitize-host"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "host_selftest: 0 failure(s)" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
