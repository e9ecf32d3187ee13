"""Centroid-split BVH: node-for-node equality with the reference's own CentroidBVHBuild
(core/CentroidBVH.cpp:226-253, compiled into oracle/_ref), the statistics the reference prints
(render_client/img_main.cpp:287-290) and structural properties of both modes."""
import os

import numpy as np
import pytest

from conftest import MODELS

# (model, loader, bvh mode) -> (nodes, leaves, depth, leaf histogram 1/2/3/4)
STATS = {
    ("cube_mirror_test.obj", "reference", "reference"): (15, 8, 2, [0, 4, 0, 4]),
    ("teapot.obj", "reference", "reference"): (1969, 985, 11, [355, 113, 185, 332]),
    ("sphere.obj", "reference", "reference"): (167, 84, 6, [0, 40, 32, 12]),
    ("tri.obj", "reference", "reference"): (3, 2, 0, [0, 0, 1, 1]),
    ("wheel.obj", "reference", "reference"): (231, 116, 7, [0, 20, 68, 28]),
    ("bunny.obj", "reference", "reference"): (49743, 24872, 16, [2390, 7745, 7162, 7575]),
    ("dragon.obj", "reference", "reference"): (68795, 34398, 16, [0, 13190, 11212, 9996]),
    ("bunny_big.obj", "reference", "reference"): (56929, 28465, 20, [11141, 1648, 7477, 8199]),
    ("cube.obj", "tolerant", "reference"): (7, 4, 1, None),
    ("bunny.obj", "tolerant", "reference"): (47489, None, 15, None),
    ("dragon.obj", "tolerant", "reference"): (61991, None, 16, None),
    ("bunny_big.obj", "tolerant", "reference"): (49183, None, 20, None),
    ("teapot.obj", "tolerant", "fixed"): (1639, None, 12, None),
    ("bunny.obj", "tolerant", "fixed"): (45633, None, 17, None),
    ("dragon.obj", "tolerant", "fixed"): (66137, None, 18, None),
    ("bunny_big.obj", "tolerant", "fixed"): (45819, None, 16, None),
}


@pytest.mark.parametrize("key", sorted(STATS))
def test_build_statistics(rt, key):
    m = rt.Mesh.load_obj(os.path.join(MODELS, key[0]), key[1])
    b = rt.BVH(m, key[2])
    nodes, leaves, depth, hist = STATS[key]
    assert len(b.nodes) == nodes and b.depth == depth and b.fallback_splits == 0
    if leaves is not None:
        assert b.leafcount == leaves
    if hist is not None:
        lv = b.nodes[b.nodes[:, 16] == 1]
        assert np.bincount(lv[:, 4], minlength=5)[1:5].tolist() == hist


def _masked(nodes):
    """Zero the words the reference leaves uninitialised (core/CentroidBVH.cpp:177,201)."""
    n = nodes.copy()
    n[:, 18:20] = 0                       # node padding
    inner = n[:, 16] == 0
    n[inner, 3] = 0                       # plane_norm.w is written as 0 by the reference; keep deterministic
    n[inner, 6:8] = 0                     # inner padding
    leaf = ~inner
    n[leaf, 5:8] = 0                      # leaf padding
    for k in range(4):                    # leaf slots beyond count
        n[leaf & (n[:, 4] <= k), k] = 0
    return n


@pytest.mark.parametrize("model,loader", [("cube_mirror_test.obj", "reference"), ("teapot.obj", "reference"),
                                          ("sphere.obj", "reference"), ("wheel.obj", "reference"),
                                          ("tri.obj", "reference"), ("cube.obj", "tolerant"),
                                          ("bunny.obj", "reference"), ("dragon.obj", "tolerant"),
                                          ("bunny_big.obj", "tolerant")])
def test_node_for_node_equal_to_reference_builder(rt, oracle, model, loader):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    m = rt.Mesh.load_obj(os.path.join(MODELS, model), loader)
    mine = rt.BVH(m, "reference")
    ref_nodes, ref_depth, ref_leaves = oracle.ref_bvh_build(m.verts, m.faces)
    assert len(ref_nodes) == len(mine.nodes) and ref_depth == mine.depth and ref_leaves == mine.leafcount
    a, b = _masked(mine.nodes), _masked(ref_nodes)
    diff = np.argwhere(a != b)
    assert diff.size == 0, f"first differing (node, word): {diff[:5].tolist()}"


@pytest.mark.parametrize("mode", ["reference", "fixed", "sah"])
@pytest.mark.parametrize("model", ["teapot.obj", "cube.obj", "bunny.obj"])
def test_tree_properties(rt, model, mode):
    m = rt.Mesh.load_obj(os.path.join(MODELS, model), "tolerant")
    b = rt.BVH(m, mode)
    n = b.nodes
    boxes = n[:, 8:16].view(np.float32)
    inner = np.flatnonzero(n[:, 16] == 0)
    leaf = np.flatnonzero(n[:, 16] == 1)
    # pre-order numbering, parent links, every triangle in exactly one leaf
    kids = n[inner][:, 4:6]
    assert np.all(kids[:, 0] == inner + 1) and np.all(kids > inner[:, None])
    assert np.all(n[kids[:, 0], 17] == inner) and np.all(n[kids[:, 1], 17] == inner) and n[0, 17] == 0xFFFFFFFF
    tris = np.concatenate([n[i, :n[i, 4]] for i in leaf])
    assert np.array_equal(np.sort(tris), np.arange(len(m.faces)))
    # a leaf's box holds every vertex of its triangles (the reference asserts one vertex, CentroidBVH.cpp:145-172)
    e = 1e-6
    for i in leaf[:: max(1, len(leaf) // 500)]:
        v = m.verts[m.faces[n[i, :n[i, 4]], :3].reshape(-1), :3]
        assert np.all(v >= boxes[i, 0:3] - e) and np.all(v <= boxes[i, 4:7] + e)
    # split axis is recorded as -e_axis
    pn = n[inner][:, 0:3].view(np.float32)
    assert np.all(np.sort(pn, axis=1) == np.array([-1, 0, 0], np.float32))


def test_small_and_degenerate_scenes(rt):
    sq = rt.Mesh.load_obj(os.path.join(MODELS, "std_square.obj"), "tolerant")
    b = rt.BVH(sq, "fixed")  # <= 4 triangles: root + one real leaf + one empty leaf
    assert len(b.nodes) == 3 and b.nodes[1, 4] == 2 and b.nodes[2, 4] == 0 and b.leafcount == 1
    # > 4 coincident triangles: no axis separates them; the reference would never return
    verts = np.array([[0, 0, 0, 1], [1, 0, 0, 1], [0, 1, 0, 1]], np.float32)
    faces = np.zeros((9, 8), np.uint32)
    faces[:, :3] = (0, 1, 2)
    m = rt.Mesh.from_arrays(verts, faces)
    for mode in ("fixed", "sah"):
        b = rt.BVH(m, mode)
        assert b.fallback_splits >= 1 and b.leafcount >= 3
        tris = np.concatenate([b.nodes[i, :b.nodes[i, 4]] for i in np.flatnonzero(b.nodes[:, 16] == 1)])
        assert np.array_equal(np.sort(tris), np.arange(9))
    assert len(rt.BVH(sq, "sah").nodes) == 3      # four or fewer triangles: the reference's three-node layout
    with pytest.raises(rt.RtError):
        bad = faces.copy(); bad[0, 0] = 99
        rt.BVH(rt.Mesh.from_arrays(verts, bad), "fixed")


def test_sah_tree_is_cheaper_to_traverse_and_flattens(rt):
    """Same node format, better topology: the re-flattening accepts it, every inner box contains its children,
    and the traversal restatement opens fewer boxes and tests fewer triangles than on the centroid tree."""
    m = rt.Mesh.load_obj(os.path.join(MODELS, "bunny.obj"), "tolerant")
    fixed, sah = rt.BVH(m, "fixed"), rt.BVH(m, "sah")
    n = sah.nodes
    boxes = n[:, 8:16].view(np.float32)
    inner = np.flatnonzero(n[:, 16] == 0)
    for c in (n[inner, 4], n[inner, 5]):
        # (an exactly flat child box is given 0.01 of thickness, as the reference does)
        assert np.all(boxes[c, 0:3] >= boxes[inner, 0:3]) and np.all(boxes[c, 4:7] <= boxes[inner, 4:7] + np.float32(0.0101))
    origin = rt.frame_camera(fixed.root_box, 320, 240)
    rays = rt.primary_rays(320, 240, origin)
    cost = []
    for b in (fixed, sah):
        sc = rt.Scene(m, b, device=-1)
        c = sc.count_fetches(rays)
        cost.append((c["nodes"], c["tris"], c["rays_hit"]))
        sc.close()
    assert cost[0][2] == cost[1][2]                       # the same rays hit something
    assert cost[1][0] < cost[0][0] and cost[1][1] < cost[0][1]
