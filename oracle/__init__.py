"""oracle -- TEST INFRASTRUCTURE.  ctypes bindings of liboracle.so (the repo's C
restatement of the reference's closest-hit arithmetic, oracle/oracle.c) and, when it
has been built, of oracle/_ref/libref.so (the reference's own core/*.cpp).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product (raytrace_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_PATH = os.path.join(_HERE, "liboracle.so")
REF_PATH = os.path.join(_HERE, "_ref", "libref.so")

_vp, _sz, _i32 = C.c_void_p, C.c_size_t, C.c_int


def _p(a):
    return C.c_void_p(a.ctypes.data) if a is not None else None


def _load_oracle():
    if not os.path.exists(ORACLE_PATH):
        raise ImportError(f"{ORACLE_PATH} missing: run `make -C oracle`")
    lib = C.CDLL(ORACLE_PATH)
    lib.oracle_closest_hit.restype = _i32
    lib.oracle_closest_hit.argtypes = [_vp, _vp, _sz, _vp, _sz, _vp, _vp, _i32]
    lib.oracle_closest_hit_bvh.restype = _i32
    lib.oracle_closest_hit_bvh.argtypes = [_vp, _vp, _vp, _sz, _vp, _sz, _vp, _i32]
    lib.oracle_tri_batch.restype = None
    lib.oracle_tri_batch.argtypes = [_vp, _vp, _sz, _vp, _vp, _vp]
    lib.oracle_aabb_batch.restype = None
    lib.oracle_aabb_batch.argtypes = [_vp, _vp, _sz, _vp]
    lib.oracle_primary_rays.restype = None
    lib.oracle_primary_rays.argtypes = [_i32, _i32, C.POINTER(C.c_float), _vp]
    lib.oracle_reflect_ray.restype = None
    lib.oracle_reflect_ray.argtypes = [_vp, _vp]
    lib.oracle_trace_setup_batch.restype = None
    lib.oracle_trace_setup_batch.argtypes = [_vp, _sz, C.POINTER(C.c_float), C.c_uint, _vp]
    lib.oracle_trace_batch.restype = None
    lib.oracle_trace_batch.argtypes = [_vp, _vp, _vp, _sz, _vp, _sz, _vp, _sz, C.POINTER(C.c_float), C.c_uint, _vp, _vp, _vp, _vp]
    return lib


lib = _load_oracle()


def have_ref():
    return os.path.exists(REF_PATH)


_ref = None


def ref():
    """The reference's own code (built from /root/reference by oracle/Makefile)."""
    global _ref
    if _ref is None:
        r = C.CDLL(REF_PATH)
        r.ref_tri_batch.restype = None
        r.ref_tri_batch.argtypes = [_vp, _vp, _sz, _vp, _vp, _vp]
        r.ref_aabb_batch.restype = None
        r.ref_aabb_batch.argtypes = [_vp, _vp, _sz, _vp]
        r.ref_normalize.restype = None
        r.ref_normalize.argtypes = [_vp, _vp]
        r.ref_bvh_build.restype = _sz
        r.ref_bvh_build.argtypes = [_vp, _vp, _sz, _sz, _vp, _sz, C.POINTER(_i32)]
        r.ref_closest_hit.restype = None
        r.ref_closest_hit.argtypes = [_vp, _vp, _sz, _vp, _sz, _vp]
        r.ref_camera_matrix.restype = None
        r.ref_camera_matrix.argtypes = [C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        r.ref_print_image.restype = _i32
        r.ref_print_image.argtypes = [_vp, _i32, _i32, C.c_char_p]
        r.ref_min_size_aabb.restype = None
        r.ref_min_size_aabb.argtypes = [_vp]
        _ref = r
    return _ref


def default_threads():
    return max(1, min(16, os.cpu_count() or 1))


def closest_hit(verts, faces, rays, threads=None, want_second=False):
    """Brute-force closest hit: returns (ids u32, t f32, u f32, v f32[, second_t f32])."""
    verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 4)
    faces = np.ascontiguousarray(faces, np.uint32).reshape(-1, 8)
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 16)
    n = len(rays)
    hits = np.zeros((n, 4), np.float32)
    second = np.zeros(n, np.float32) if want_second else None
    rc = lib.oracle_closest_hit(_p(verts), _p(faces), len(faces), _p(rays), n, _p(hits), _p(second),
                                threads or default_threads())
    if rc != 0:
        raise RuntimeError("oracle_closest_hit failed")
    out = (hits[:, 0].view(np.uint32).copy(), hits[:, 1].copy(), hits[:, 2].copy(), hits[:, 3].copy())
    return out + (second,) if want_second else out


def closest_hit_bvh(verts, faces, nodes, rays, threads=None):
    """The same closest hit through the reference-format tree `nodes` ((n, 20) u32 = 80-byte CentroidBVHNode records):
    returns (ids u32, t f32, u f32, v f32), equal to closest_hit()'s."""
    verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 4)
    faces = np.ascontiguousarray(faces, np.uint32).reshape(-1, 8)
    nodes = np.ascontiguousarray(nodes, np.uint32).reshape(-1, 20)
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 16)
    n = len(rays)
    hits = np.zeros((n, 4), np.float32)
    rc = lib.oracle_closest_hit_bvh(_p(verts), _p(faces), _p(nodes), len(nodes), _p(rays), n, _p(hits), threads or default_threads())
    if rc != 0:
        raise RuntimeError("oracle_closest_hit_bvh failed (malformed or too deep a tree)")
    return hits[:, 0].view(np.uint32).copy(), hits[:, 1].copy(), hits[:, 2].copy(), hits[:, 3].copy()


def _batch(fn, tri, ray):
    tri = np.ascontiguousarray(tri, np.float32).reshape(-1, 3, 4)
    ray = np.ascontiguousarray(ray, np.float32).reshape(-1, 8)
    n = len(ray)
    hit = np.zeros(n, np.int32)
    t = np.zeros(n, np.float32)
    uv = np.zeros((n, 2), np.float32)
    fn(_p(tri), _p(ray), n, _p(hit), _p(t), _p(uv))
    return hit, t, uv


def tri_batch(tri, ray):
    return _batch(lib.oracle_tri_batch, tri, ray)


def ref_tri_batch(tri, ray):
    return _batch(ref().ref_tri_batch, tri, ray)


def _aabb(fn, box, ray):
    box = np.ascontiguousarray(box, np.float32).reshape(-1, 8)
    ray = np.ascontiguousarray(ray, np.float32).reshape(-1, 8)
    hit = np.zeros(len(ray), np.int32)
    fn(_p(box), _p(ray), len(ray), _p(hit))
    return hit


def aabb_batch(box, ray):
    return _aabb(lib.oracle_aabb_batch, box, ray)


def ref_aabb_batch(box, ray):
    return _aabb(ref().ref_aabb_batch, box, ray)


def primary_rays(width, height, origin):
    out = np.zeros((width * height, 16), np.float32)
    o = (C.c_float * 3)(*[float(x) for x in origin])
    lib.oracle_primary_rays(width, height, o, _p(out))
    return out


def reflect_rays(records):
    records = np.ascontiguousarray(records, np.float32).reshape(-1, 16)
    out = np.zeros((len(records), 8), np.float32)
    for i in range(len(records)):
        lib.oracle_reflect_ray(C.c_void_p(records[i].ctypes.data), C.c_void_p(out[i].ctypes.data))
    return out


def _cam(cam):
    return (C.c_float * 16)(*[float(x) for x in np.asarray(cam, np.float32).reshape(16)])


def trace_setup(records, cam, op):
    """The rays the `trace` kernel traces for these input records (op 0: camera matrix, op 1: reflection), as
    (n,16) records with origin and direction filled (render_client/tracekernel.ck:224-265)."""
    records = np.ascontiguousarray(records, np.float32).reshape(-1, 16)
    out = np.zeros_like(records)
    lib.oracle_trace_setup_batch(_p(records), len(records), _cam(cam), op, _p(out))
    return out


def trace_epilogue(mesh, records, cam, op, ids, t, u, v):
    """Output records (n,16), RGBA pixels (n,4) and a `defined` mask for the given closest hits
    (render_client/tracekernel.ck:282-286, :373-413)."""
    records = np.ascontiguousarray(records, np.float32).reshape(-1, 16)
    n = len(records)
    hits = np.zeros((n, 4), np.float32)
    hits[:, 0] = np.asarray(ids, np.uint32).view(np.float32)
    hits[:, 1], hits[:, 2], hits[:, 3] = t, u, v
    verts = np.ascontiguousarray(mesh.verts, np.float32)
    faces = np.ascontiguousarray(mesh.faces, np.uint32)
    normals = np.ascontiguousarray(mesh.normals, np.float32)
    mats = np.ascontiguousarray(mesh.materials, np.float32)
    rec = np.zeros((n, 16), np.float32)
    rgba = np.zeros((n, 4), np.float32)
    defined = np.zeros(n, np.int32)
    lib.oracle_trace_batch(_p(verts), _p(faces), _p(normals) if len(normals) else None, len(normals),
                           _p(mats) if len(mats) else None, len(mats), _p(records), n, _cam(cam), op, _p(hits), _p(rec), _p(rgba),
                           _p(defined))
    return rec, rgba, defined.astype(bool)


def ref_closest_hit(verts, faces, rays):
    """Index-order brute force over the REFERENCE's intersect_ray_triangle (single thread)."""
    verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 4)
    faces = np.ascontiguousarray(faces, np.uint32).reshape(-1, 8)
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 16)
    hits = np.zeros((len(rays), 4), np.float32)
    ref().ref_closest_hit(_p(verts), _p(faces), len(faces), _p(rays), len(rays), _p(hits))
    return hits[:, 0].view(np.uint32).copy(), hits[:, 1].copy(), hits[:, 2].copy(), hits[:, 3].copy()


def ref_camera_matrix(rot_x, rot_y, trans):
    """core/Affine.h's rotate3 / multiply driven as render_client/img_main.cpp:627-660 drives them: (4,4) f32."""
    t = (C.c_float * 3)(*[float(x) for x in trans])
    out = (C.c_float * 16)()
    ref().ref_camera_matrix(float(rot_x), float(rot_y), t, out)
    return np.array(list(out), dtype=np.float32).reshape(4, 4)


def ref_print_image(bgra, path):
    """cpu_raycast/bmp.c printImage on an (H,W,4) u8 image, file opened in binary mode."""
    bgra = np.ascontiguousarray(bgra, np.uint8)
    if ref().ref_print_image(_p(bgra), bgra.shape[1], bgra.shape[0], os.fsencode(path)) != 0:
        raise OSError(f"cannot write {path}")


def ref_bvh_build(verts, faces):
    """CentroidBVHBuild of the reference: (nodes (n,20) u32, depth, leafcount)."""
    verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 4)
    faces = np.ascontiguousarray(faces, np.uint32).reshape(-1, 8)
    stats = (_i32 * 3)()
    n = ref().ref_bvh_build(_p(faces), _p(verts), len(faces), len(verts), None, 0, stats)
    nodes = np.zeros((n, 20), np.uint32)
    ref().ref_bvh_build(_p(faces), _p(verts), len(faces), len(verts), _p(nodes), n, stats)
    return nodes, stats[0], stats[2]
