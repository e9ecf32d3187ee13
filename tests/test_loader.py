"""OBJ/MTL loader: token-level behaviour of both tokenizer modes (core/ObjLoader.cpp:12-34,
:63-220, :244-354) and the per-model counts the reference loader produces."""
import os

import numpy as np
import pytest

from conftest import MODELS


def _write(tmp_path, name, text):
    p = tmp_path / name
    p.write_text(text)
    return str(p)


def test_reference_tokenizer_keeps_empty_tokens(rt, tmp_path):
    p = _write(tmp_path, "a.obj", "v  1 2 3\nv 4 5 6\nv 7 8 9\nf 1 2 3\nf  1 2 3\nf 1 2 3  \n")
    ref = rt.Mesh.load_obj(p, "reference")
    # "v  1 2 3" -> tokens v,"",1,2,3 -> (0,1,2); the double-spaced and trailing-space faces have 5 tokens
    assert ref.verts[0].tolist() == [0.0, 1.0, 2.0, 1.0] and ref.verts[1].tolist() == [4.0, 5.0, 6.0, 1.0]
    assert len(ref.faces) == 1 and ref.log.count("Quads and ngons not supported, ignoring.") == 2
    tol = rt.Mesh.load_obj(p, "tolerant")
    assert tol.verts[0].tolist() == [1.0, 2.0, 3.0, 1.0] and len(tol.faces) == 3


def test_face_forms_and_records(rt, tmp_path):
    text = ("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1\nvn 0 0 1\nvn 0 1\nvt 0.5 0.5\n"
            "f 1 2 3\nf 1/1 2/1 3/1\nf 1/1/1 2/1/2 3/1/1\nf 1//1 2//1 3//2\nf 1 2\nf 1 2 3 4\nf /1 2 3\n# c\n\nf 3 2 1")
    m = rt.Mesh.load_obj(_write(tmp_path, "b.obj", text), "reference")
    assert m.verts[3].tolist() == [0, 0, 0, 1]               # incomplete vertex -> origin
    assert m.normals.tolist() == [[0, 0, 1, 0], [1, 0, 0, 0]]  # incomplete normal -> <1,0,0>
    assert m.faces[:, :3].tolist() == [[0, 1, 2]] * 4 + [[2, 1, 0]]  # last line has no trailing newline
    assert m.faces[2, 4:7].tolist() == [0, 1, 0] and m.faces[3, 4:7].tolist() == [0, 0, 1]
    assert m.faces[0, 4:8].tolist() == [0, 0, 0, 0]
    for msg in ("incomplete face record, ignoring", "Quads and ngons not supported, ignoring.",
                "face record missing vertex index, skipping", "incomplete vertex record, using (0,0,0)"):
        assert msg in m.log


def test_materials_and_usemtl(rt, tmp_path):
    _write(tmp_path, "m.mtl", "newmtl A\ncolor 1 0.5 0.25\nrough 0.3\nind_ref 1.5\n\nnewmtl B\ncolor 0 1 0\nnewmtl A\ncolor 9 9 9\n")
    text = "mtllib m.mtl\nmtllib missing.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl B\nf 1 2 3\nusemtl nope\nf 1 2 3\nusemtl A\nf 1 2 3\n"
    m = rt.Mesh.load_obj(_write(tmp_path, "c.obj", text), "tolerant")
    # sky, A, B, then the trailing redefinition of A is appended and re-pointed (no duplicate check at EOF)
    assert len(m.materials) == 4
    assert m.materials[0].tolist() == [0.0] * 12
    assert m.materials[1, :3].tolist() == [1.0, 0.5, 0.25] and m.materials[1, 8] == 1.5 and abs(m.materials[1, 9] - 0.3) < 1e-7
    assert m.materials[2, :3].tolist() == [0, 1, 0] and m.materials[2, 8] == 1 and m.materials[2, 9] == 1
    assert m.material_index("A") == 3 and m.material_index("B") == 2 and m.material_index("zzz") == -1
    assert m.faces[:, 3].tolist() == [2, 0, 3]               # unknown name -> 0 (map::operator[])


# (model, loader) -> (verts, tris); the reference-loader rows are what the reference itself loads
COUNTS = {
    ("cube.obj", "reference"): (8, 0), ("cube.obj", "tolerant"): (8, 12),
    ("cube_mirror_test.obj", "reference"): (16, 24), ("teapot.obj", "reference"): (1292, 2464),
    ("sphere.obj", "reference"): (114, 224), ("tri.obj", "reference"): (24, 7), ("tri.obj", "tolerant"): (24, 8),
    ("wheel.obj", "reference"): (180, 356), ("std_square.obj", "reference"): (4, 2),
    ("bunny.obj", "reference"): (34835, 69666), ("dragon.obj", "reference"): (50000, 100000),
    ("bunny_big.obj", "reference"): (34834, 69664),
}


@pytest.mark.parametrize("key", sorted(COUNTS))
def test_model_counts(rt, key):
    m = rt.Mesh.load_obj(os.path.join(MODELS, key[0]), key[1])
    assert (len(m.verts), len(m.faces)) == COUNTS[key]


def test_reference_loader_flattens_double_spaced_vertices(rt):
    ref = rt.Mesh.load_obj(os.path.join(MODELS, "dragon.obj"), "reference")
    tol = rt.Mesh.load_obj(os.path.join(MODELS, "dragon.obj"), "tolerant")
    assert np.all(ref.verts[:, 0] == 0) and np.array_equal(ref.verts[:, 1:3], tol.verts[:, 0:2])
    cube = rt.Mesh.load_obj(os.path.join(MODELS, "cube.obj"), "reference")
    assert np.all(cube.verts[:, [0, 2]] == 0)               # "v  x  y  z" -> (0, x, 0)


def test_cube_mirror_materials(rt):
    m = rt.Mesh.load_obj(os.path.join(MODELS, "cube_mirror_test.obj"), "reference")
    assert len(m.materials) == 6 and len(m.normals) == 12
    assert m.faces[0].tolist() == [1, 2, 0, 4, 0, 0, 0, 0]
    assert m.material_index("Mirror") == 2 and m.materials[2, 9] == 1.0 and m.materials[1, 9] == 0.0


def test_extended_loader_fans_polygons_and_resolves_relative_indices(rt, tmp_path):
    """The reference drops every face with more than three corners (core/ObjLoader.cpp:151-154); the extended mode
    turns them into triangle fans and accepts OBJ's negative indices.  Triangles-only files load identically."""
    text = ("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0.5 1.5 0\nvn 0 0 1\n"
            "f 1 2 3 4\n"            # quad -> (1,2,3) (1,3,4)
            "f 1//1 2//1 3//1 4//1 5//1\n"   # pentagon -> three triangles, normals kept
            "f -5 -4 -3\n"           # relative: vertices 1 2 3
            "f 1 2 3\n")
    p = _write(tmp_path, "poly.obj", text)
    ext = rt.Mesh.load_obj(p, "extended")
    assert len(ext.faces) == 2 + 3 + 1 + 1
    assert ext.faces[:2, :3].tolist() == [[0, 1, 2], [0, 2, 3]]
    assert ext.faces[2:5, :3].tolist() == [[0, 1, 2], [0, 2, 3], [0, 3, 4]] and np.all(ext.faces[2:5, 4:7] == 0)
    assert ext.faces[5, :3].tolist() == [0, 1, 2] and ext.faces[6, :3].tolist() == [0, 1, 2]
    tol = rt.Mesh.load_obj(p, "tolerant")
    assert len(tol.faces) == 2 and "Quads and ngons not supported, ignoring." in tol.log   # polygons dropped, as the reference does
    for model in ("teapot.obj", "cube_mirror_test.obj", "bunny.obj"):
        a = rt.Mesh.load_obj(os.path.join(MODELS, model), "tolerant")
        b = rt.Mesh.load_obj(os.path.join(MODELS, model), "extended")
        assert np.array_equal(a.faces, b.faces) and np.array_equal(a.verts, b.verts)
    wheel_t = rt.Mesh.load_obj(os.path.join(MODELS, "wheel.obj"), "tolerant")
    wheel_e = rt.Mesh.load_obj(os.path.join(MODELS, "wheel.obj"), "extended")
    assert len(wheel_e.faces) >= len(wheel_t.faces)
    bvh = rt.BVH(wheel_e, "sah")
    sc = rt.Scene(wheel_e, bvh, device=-1)     # the fanned mesh flattens
    sc.close()


def test_texture_coordinates_and_their_face_indices(rt, tmp_path):
    """`vt` records as core/ObjLoader.cpp:96-118 parses them, plus the per-corner indices it leaves commented out
    (:178-186): 0-based, 0xFFFFFFFF where a face names none."""
    p = tmp_path / "uv.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\n"
                 "vt 0.25 0.5\nvt 1 0 0.75\nvt 0.5\nvt 0 1\n"
                 "vn 0 0 1\n"
                 "f 1/1/1 2/2/1 3/4/1\n"      # v/vt/vn
                 "f 2/2 4/3 3/1\n"            # v/vt
                 "f 1//1 2//1 4//1\n"         # no vt
                 "f 1 2 3\n")
    for mode in ("reference", "tolerant", "extended"):
        m = rt.Mesh.load_obj(str(p), mode)
        assert m.uvws.shape == (4, 4)
        assert np.allclose(m.uvws[:, :3], [[0.25, 0.5, 0], [1, 0, 0.75], [0, 0, 0], [0, 1, 0]])
        assert "incomplete texture coordinate record, using (0,0)" in m.log
        miss = 0xFFFFFFFF
        assert m.face_uvws.tolist() == [[0, 1, 3], [1, 2, 0], [miss] * 3, [miss] * 3]
        assert m.faces[:, :3].tolist() == [[0, 1, 2], [1, 3, 2], [0, 1, 3], [0, 1, 2]]
    teapot = rt.Mesh.load_obj(os.path.join(MODELS, "teapot.obj"), "reference")
    assert len(teapot.face_uvws) == len(teapot.faces)
