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
    lib.oracle_tri_batch.restype = None
    lib.oracle_tri_batch.argtypes = [_vp, _vp, _sz, _vp, _vp, _vp]
    lib.oracle_aabb_batch.restype = None
    lib.oracle_aabb_batch.argtypes = [_vp, _vp, _sz, _vp]
    lib.oracle_primary_rays.restype = None
    lib.oracle_primary_rays.argtypes = [_i32, _i32, C.POINTER(C.c_float), _vp]
    lib.oracle_reflect_ray.restype = None
    lib.oracle_reflect_ray.argtypes = [_vp, _vp]
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


def ref_closest_hit(verts, faces, rays):
    """Index-order brute force over the REFERENCE's intersect_ray_triangle (single thread)."""
    verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 4)
    faces = np.ascontiguousarray(faces, np.uint32).reshape(-1, 8)
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 16)
    hits = np.zeros((len(rays), 4), np.float32)
    ref().ref_closest_hit(_p(verts), _p(faces), len(faces), _p(rays), len(rays), _p(hits))
    return hits[:, 0].view(np.uint32).copy(), hits[:, 1].copy(), hits[:, 2].copy(), hits[:, 3].copy()


def ref_bvh_build(verts, faces):
    """CentroidBVHBuild of the reference: (nodes (n,20) u32, depth, leafcount)."""
    verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 4)
    faces = np.ascontiguousarray(faces, np.uint32).reshape(-1, 8)
    stats = (_i32 * 3)()
    n = ref().ref_bvh_build(_p(faces), _p(verts), len(faces), len(verts), None, 0, stats)
    nodes = np.zeros((n, 20), np.uint32)
    ref().ref_bvh_build(_p(faces), _p(verts), len(faces), len(verts), _p(nodes), n, stats)
    return nodes, stats[0], stats[2]
