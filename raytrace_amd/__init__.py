"""raytrace_amd -- Python face of librtamd.so, the MI355X closest-hit ray/scene path.

This module holds no algorithm.  It binds the C ABI of ``include/rt_amd.h`` with ctypes
and mirrors, for the parity tests and the benchmark, the host-side calling sequence of the
reference's ``cl_init``/``cl_update`` (render_client/img_main.cpp:120-316, :437-507):
load a mesh, build the BVH, create the device scene, trace ray records.

The shared library must have been built (``make`` or ``__graft_entry__.build()``); if it
is missing, importing this package raises -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_LIB") or os.path.join(_HERE, "librtamd.so")   # RT_LIB: an A/B build of the same library
REPO_ROOT = os.path.dirname(_HERE)
MODELS_DIR = os.path.join(REPO_ROOT, "assets", "models")

RAY_FLOATS = 16          # one 64-byte RayIntersection record = 16 x 4 bytes
MISS_ID = 0xFFFFFFFF
LOADER = {"reference": 0, "tolerant": 1, "extended": 2}
BVH_MODE = {"reference": 0, "fixed": 1, "sah": 2}


class RtError(RuntimeError):
    pass


def _preload_hip_runtime():
    """Make sure the process ends up with ONE HIP runtime.

    PyTorch's ROCm wheel ships its own libamdhip64.so (same SONAME as the system one).  If
    this library bound the system copy first and torch were imported later, the process
    would hold two runtimes and device pointers could not be shared.  When torch is
    installed but not yet imported, its copy is loaded first so both sides bind it.
    """
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.submodule_search_locations:
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make` (or __graft_entry__.build()); "
            "raytrace_amd has no fallback implementation")
    _preload_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    vp, sz, u32, u64, i32 = C.c_void_p, C.c_size_t, C.c_uint, C.c_uint64, C.c_int
    sig = {
        "rt_last_error": (C.c_char_p, []),
        "rt_device_count": (i32, []),
        "rt_mesh_load_obj": (i32, [C.c_char_p, u32, C.POINTER(vp)]),
        "rt_mesh_from_arrays": (i32, [vp, sz, vp, sz, vp, sz, vp, sz, C.POINTER(vp)]),
        "rt_mesh_nverts": (sz, [vp]), "rt_mesh_ntris": (sz, [vp]),
        "rt_mesh_nnormals": (sz, [vp]), "rt_mesh_nmaterials": (sz, [vp]),
        "rt_mesh_verts": (vp, [vp]), "rt_mesh_faces": (vp, [vp]),
        "rt_mesh_normals": (vp, [vp]), "rt_mesh_materials": (vp, [vp]),
        "rt_mesh_nuvws": (sz, [vp]), "rt_mesh_uvws": (vp, [vp]), "rt_mesh_face_uvws": (vp, [vp]),
        "rt_mesh_material_index": (i32, [vp, C.c_char_p]),
        "rt_mesh_log": (C.c_char_p, [vp]),
        "rt_mesh_destroy": (None, [vp]),
        "rt_bvh_build": (i32, [vp, vp, sz, sz, u32, C.POINTER(vp)]),
        "rt_bvh_nodes": (vp, [vp]), "rt_bvh_nnodes": (sz, [vp]),
        "rt_bvh_depth": (i32, [vp]), "rt_bvh_leafcount": (i32, [vp]),
        "rt_bvh_fallback_splits": (i32, [vp]),
        "rt_bvh_destroy": (None, [vp]),
        "rt_primary_rays": (i32, [i32, i32, C.POINTER(C.c_float), vp]),
        "rt_camera_matrix": (i32, [C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
        "rt_frame_camera": (i32, [vp, i32, i32, C.POINTER(C.c_float)]),
        "rt_ao_rays": (i32, [vp, vp, u64, sz, sz, vp]),
        "rt_scene_create": (i32, [vp, sz, vp, sz, vp, sz, vp, sz, vp, sz, i32, u32, C.POINTER(vp)]),
        "rt_scene_destroy": (None, [vp]),
        "rt_scene_info": (i32, [vp, C.POINTER(u64)]),
        "rt_scene_records": (i32, [vp, i32, vp, sz, C.POINTER(sz)]),
        "rt_scene_set_param": (i32, [vp, C.c_char_p, i32]),
        "ray_scene_intersection": (i32, [vp, sz, vp, vp]),
        "rt_intersect_async": (i32, [vp, vp, sz, vp, vp, vp]),
        "rt_intersect_static": (i32, [vp, vp, sz, vp, vp, vp]),
        "rt_intersect_compact": (i32, [vp, vp, sz, vp, sz, vp]),
        "rt_trace_pass_bgra8": (i32, [vp, vp, vp, vp, sz, C.POINTER(C.c_float), u32, vp]),
        "rt_trace_pass": (i32, [vp, vp, vp, vp, sz, C.POINTER(C.c_float), u32, vp]),
        "rt_trace_primary": (i32, [vp, i32, i32, C.POINTER(C.c_float), vp, vp, vp]),
        "rt_scene_count_fetches": (i32, [vp, vp, sz, sz, C.POINTER(u64), vp]),
        "rt_scene_count_fetches_device": (i32, [vp, vp, sz, C.POINTER(u64)]),
        "rt_scene_unit_costs": (i32, [vp, vp, sz, vp, sz, C.POINTER(sz)]),
        "rt_scene_unit_costs_primary": (i32, [vp, i32, i32, C.POINTER(C.c_float), vp, sz, C.POINTER(sz)]),
        "rt_scatter_tiles": (i32, [i32, vp, sz, vp, vp, sz, i32, i32, i32, sz, vp, vp]),
        "rt_pack_bgra8": (i32, [i32, vp, sz, vp, vp]),
        "rt_pack_hits8": (i32, [i32, vp, sz, vp, vp]),
        "rt_composite_layers": (i32, [i32, vp, vp, sz, vp, vp, vp]),
        "rt_shard_plan": (i32, [i32, i32, i32, i32, vp, vp, vp, vp, vp, vp]),
        "rt_shard_square_costs": (i32, [vp, sz, i32, sz, vp, vp, sz, vp, sz]),
        "rt_comm_sum_f64": (i32, [vp, i32, i32, vp, sz]),
        "rt_shard_ray_indices": (i32, [i32, i32, i32, vp, sz, u32, vp, vp]),
        "rt_comm_unique_id": (i32, [vp]),
        "rt_comm_create": (i32, [vp, i32, i32, i32, C.POINTER(vp)]),
        "rt_comm_destroy": (None, [vp]),
        "rt_comm_check": (i32, [vp, i32, i32, i32]),
        "rt_debug_build_order": (i32, [i32, vp, sz, vp, vp]),
        "rt_gather_scatter": (i32, [vp, i32, i32, i32, i32, vp, vp, vp, sz, vp, vp, sz, i32, i32, i32, vp, vp]),
        "rt_tiled_create": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, i32, C.c_double, C.POINTER(vp)]),
        "rt_tiled_render": (i32, [vp, vp, vp, vp]),
        "rt_tiled_redeal": (i32, [vp, vp, C.POINTER(sz)]),
        "rt_tiled_info": (i32, [vp, C.POINTER(u64)]),
        "rt_tiled_destroy": (None, [vp]),
        "rt_scene_wave_timeline": (i32, [vp, vp, sz, vp, sz]),
        "rt_scene_wave_model": (i32, [vp, vp, sz, u32, u32, sz, i32, C.POINTER(u64)]),
        "rt_debug_tri_batch": (i32, [i32, vp, vp, sz, vp, vp, vp]),
        "rt_write_bmp32": (i32, [C.c_char_p, vp, i32, i32]),
        "rt_write_pfm": (i32, [C.c_char_p, vp, i32, i32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError here means the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib, sorted(sig)


lib, EXPORTS = _load()


def _check(rc):
    if rc != 0:
        raise RtError(f"rt error {rc}: {lib.rt_last_error().decode(errors='replace')}")


def _ptr(a):
    """Address of a numpy array's buffer, or a raw integer device address, or None."""
    if a is None:
        return None
    if isinstance(a, (int, np.integer)):
        return C.c_void_p(int(a))
    return C.c_void_p(a.ctypes.data)


def _copy(ptr, count, width, dtype):
    if not ptr or count == 0:
        return np.zeros((0, width), dtype=dtype)
    raw = (C.c_char * (count * width * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(raw, dtype=dtype).reshape(count, width).copy()


class Mesh:
    """Vertices (n,4) f32, faces (n,8) u32 {v0,v1,v2,mat,n0,n1,n2,0}, normals (n,4) f32,
    materials (n,12) f32-view of the 48-byte records.  core/ObjLoader.h:12-14."""

    def __init__(self, handle):
        self._h = handle
        self.verts = _copy(lib.rt_mesh_verts(handle), lib.rt_mesh_nverts(handle), 4, np.float32)
        self.faces = _copy(lib.rt_mesh_faces(handle), lib.rt_mesh_ntris(handle), 8, np.uint32)
        self.normals = _copy(lib.rt_mesh_normals(handle), lib.rt_mesh_nnormals(handle), 4, np.float32)
        self.materials = _copy(lib.rt_mesh_materials(handle), lib.rt_mesh_nmaterials(handle), 12, np.float32)
        self.uvws = _copy(lib.rt_mesh_uvws(handle), lib.rt_mesh_nuvws(handle), 4, np.float32)
        fu = lib.rt_mesh_face_uvws(handle)
        self.face_uvws = _copy(fu, len(self.faces), 3, np.uint32) if fu else np.zeros((0, 3), np.uint32)
        self.log = lib.rt_mesh_log(handle).decode(errors="replace").splitlines()

    @classmethod
    def load_obj(cls, path, loader="tolerant"):
        h = C.c_void_p()
        _check(lib.rt_mesh_load_obj(os.fsencode(path), LOADER[loader], C.byref(h)))
        return cls(h)

    @classmethod
    def from_arrays(cls, verts, faces, normals=None, materials=None):
        verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 4)
        faces = np.ascontiguousarray(faces, dtype=np.uint32).reshape(-1, 8)
        normals = np.zeros((0, 4), np.float32) if normals is None else np.ascontiguousarray(normals, np.float32).reshape(-1, 4)
        materials = np.zeros((0, 12), np.float32) if materials is None else np.ascontiguousarray(materials, np.float32).reshape(-1, 12)
        h = C.c_void_p()
        _check(lib.rt_mesh_from_arrays(_ptr(verts), len(verts), _ptr(faces), len(faces),
                                       _ptr(normals) if len(normals) else None, len(normals),
                                       _ptr(materials) if len(materials) else None, len(materials), C.byref(h)))
        return cls(h)

    def material_index(self, name):
        return lib.rt_mesh_material_index(self._h, name.encode())

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:
            lib.rt_mesh_destroy(h)


class BVH:
    """The 80-byte CentroidBVHNode array (n,20) u32-view.  core/CentroidBVH.h:90-94."""

    def __init__(self, mesh, mode="fixed"):
        h = C.c_void_p()
        _check(lib.rt_bvh_build(_ptr(mesh.faces), _ptr(mesh.verts), len(mesh.faces), len(mesh.verts),
                                BVH_MODE[mode], C.byref(h)))
        self._h = h
        self.nodes = _copy(lib.rt_bvh_nodes(h), lib.rt_bvh_nnodes(h), 20, np.uint32)
        self.depth = lib.rt_bvh_depth(h)
        self.leafcount = lib.rt_bvh_leafcount(h)
        self.fallback_splits = lib.rt_bvh_fallback_splits(h)

    @property
    def root_box(self):
        """(2,4) f32: the root AABB min and max rows."""
        return self.nodes[0, 8:16].view(np.float32).reshape(2, 4).copy()

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:
            lib.rt_bvh_destroy(h)


def primary_rays(width, height, origin):
    """Camera ray table of render_client/img_main.cpp:253-267 as an (W*H, 16) f32 array."""
    out = np.zeros((width * height, RAY_FLOATS), dtype=np.float32)
    o = (C.c_float * 3)(*[float(x) for x in origin])
    _check(lib.rt_primary_rays(width, height, o, _ptr(out)))
    return out


def camera_matrix(rot_x, rot_y, trans):
    """(4,4) f32 camera matrix of the reference's frame loop (render_client/img_main.cpp:627-660)."""
    t = (C.c_float * 3)(*[float(x) for x in trans])
    out = (C.c_float * 16)()
    _check(lib.rt_camera_matrix(float(rot_x), float(rot_y), t, out))
    return np.array(list(out), dtype=np.float32).reshape(4, 4)


def frame_camera(root_box, width, height):
    box = np.ascontiguousarray(root_box, dtype=np.float32).reshape(2, 4)
    o = (C.c_float * 3)()
    _check(lib.rt_frame_camera(_ptr(box), width, height, o))
    return np.array([o[0], o[1], o[2]], dtype=np.float32)


def ao_rays(mesh, root_box, seed, first, count):
    out = np.zeros((count, RAY_FLOATS), dtype=np.float32)
    box = np.ascontiguousarray(root_box, dtype=np.float32).reshape(2, 4)
    _check(lib.rt_ao_rays(mesh._h, _ptr(box), seed, first, count, _ptr(out)))
    return out


def hit_fields(hits):
    """Splits (n,16) f32 hit records into (triangle_id u32, t f32, u f32, v f32)."""
    ids = hits[:, 14].view(np.uint32)
    t = hits[:, 15]
    return ids.copy(), t.copy(), hits[:, 12].copy(), hits[:, 13].copy()


class Scene:
    """Device-resident scene.  Stands in for the buffer uploads and static kernel-argument
    bindings of render_client/img_main.cpp:243-249,293,298-311."""

    def __init__(self, mesh, bvh, device=0, width=2, flatten_on_device=False):
        h = C.c_void_p()
        nodes = np.ascontiguousarray(bvh.nodes if isinstance(bvh, BVH) else bvh, dtype=np.uint32).reshape(-1, 20)
        _check(lib.rt_scene_create(_ptr(mesh.verts), len(mesh.verts), _ptr(mesh.faces), len(mesh.faces),
                                   _ptr(mesh.normals) if len(mesh.normals) else None, len(mesh.normals),
                                   _ptr(mesh.materials) if len(mesh.materials) else None, len(mesh.materials),
                                   _ptr(nodes), len(nodes), device, (4 if width == 4 else 0) | (0x100 if flatten_on_device else 0),
                                   C.byref(h)))
        self._h = h
        self.device = device

    def info(self):
        v = (C.c_uint64 * 8)()
        _check(lib.rt_scene_info(self._h, v))
        keys = ["nodes2", "nodes4", "leaf_tris", "depth2", "depth4", "node2_bytes", "node4_bytes", "tri_bytes"]
        return dict(zip(keys, [int(x) for x in v]))

    def records(self, which):
        """Raw bytes (u8 array) of one re-flattened record array: 0 nodes2, 1 nodesq2, 2 nodes4, 3 nodesq4, 4 triangles."""
        n = C.c_size_t()
        _check(lib.rt_scene_records(self._h, which, None, 0, C.byref(n)))
        out = np.zeros(n.value, np.uint8)
        _check(lib.rt_scene_records(self._h, which, _ptr(out), len(out), C.byref(n)))
        return out

    def set_param(self, key, value):
        _check(lib.rt_scene_set_param(self._h, key.encode(), int(value)))

    def intersect(self, rays):
        """ray_scene_intersection on host arrays: (n,16) f32 rays -> (n,16) f32 hit records."""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, RAY_FLOATS)
        hits = np.empty_like(rays)
        _check(lib.ray_scene_intersection(_ptr(rays), len(rays), self._h, _ptr(hits)))
        return hits

    def intersect_device(self, rays_ptr, n, hits_ptr):
        """ray_scene_intersection on raw device addresses (synchronous)."""
        _check(lib.ray_scene_intersection(_ptr(rays_ptr), n, self._h, _ptr(hits_ptr)))

    def intersect_async(self, rays_ptr, n, hits_ptr=None, hits16_ptr=None, stream=0):
        _check(lib.rt_intersect_async(self._h, _ptr(rays_ptr), n, _ptr(hits_ptr), _ptr(hits16_ptr),
                                      C.c_void_p(stream) if stream else None))

    def intersect_static(self, rays_ptr, n, hits_ptr=None, hits16_ptr=None, stream=0):
        """rt_intersect_async's allocation-free twin (fixed schedule tables inside the library)."""
        _check(lib.rt_intersect_static(self._h, _ptr(rays_ptr), n, _ptr(hits_ptr), _ptr(hits16_ptr),
                                       C.c_void_p(stream) if stream else None))

    def intersect_compact(self, rays_ptr, n, out_ptr, record_bytes, stream=0):
        """Only a compact record per ray: 16 = {id, t, u, v}, 8 = {id, t}, 4 = {id} bytes."""
        _check(lib.rt_intersect_compact(self._h, _ptr(rays_ptr), n, _ptr(out_ptr), record_bytes,
                                        C.c_void_p(stream) if stream else None))

    def trace_primary(self, width, height, origin, hits_ptr=None, hits16_ptr=None, stream=0):
        o = (C.c_float * 3)(*[float(x) for x in origin])
        _check(lib.rt_trace_primary(self._h, width, height, o, _ptr(hits_ptr), _ptr(hits16_ptr),
                                    C.c_void_p(stream) if stream else None))

    def trace_pass(self, rays_in_ptr, rays_out_ptr, rgba_ptr, n, cam, op, stream=0):
        cam_arr = (C.c_float * 16)(*[float(x) for x in np.asarray(cam, dtype=np.float32).reshape(16)])
        _check(lib.rt_trace_pass(self._h, _ptr(rays_in_ptr), _ptr(rays_out_ptr), _ptr(rgba_ptr), n, cam_arr, op,
                                 C.c_void_p(stream) if stream else None))

    def trace_pass_bgra8(self, rays_in_ptr, rays_out_ptr, bgra_ptr, n, cam, op, stream=0):
        """rt_trace_pass with the pixel written as 8-bit BGRA (4 bytes per ray)."""
        cam_arr = (C.c_float * 16)(*[float(x) for x in np.asarray(cam, dtype=np.float32).reshape(16)])
        _check(lib.rt_trace_pass_bgra8(self._h, _ptr(rays_in_ptr), _ptr(rays_out_ptr), _ptr(bgra_ptr), n, cam_arr, op,
                                       C.c_void_p(stream) if stream else None))

    def count_fetches(self, rays, stride=1, per_ray=False):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, RAY_FLOATS)
        v = (C.c_uint64 * 6)()
        pr = np.zeros(((len(rays) + stride - 1) // stride, 2), np.uint32) if per_ray else None
        _check(lib.rt_scene_count_fetches(self._h, _ptr(rays), len(rays), stride, v, _ptr(pr)))
        out = dict(zip(["nodes", "tris", "leaves", "rays_hit", "rays", "max_stack"], [int(x) for x in v]))
        return (out, pr) if per_ray else out

    def wave_model(self, rays, image_width=0, refill=0, max_waves=0, policy=0):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, RAY_FLOATS)
        v = (C.c_uint64 * 16)()
        _check(lib.rt_scene_wave_model(self._h, _ptr(rays), len(rays), image_width, refill, max_waves, policy, v))
        keys = ["waves", "rays", "wave_node_steps", "lane_node_steps", "wave_tri_steps", "lane_tri_steps",
                "wave_leaf_phases", "lane_leaf_phases", "refills", "refill_lanes", "max_wave_steps", "uniform_node_steps",
                "iterations", "max_wave_iterations", "two_node_steps", "three_node_steps"]
        return dict(zip(keys, [int(x) for x in v]))

    def wave_timeline(self, rays_ptr, n):
        """(waves, 4) u64: start tick, end tick (100 MHz), HW_ID, XCC_ID of every wave of one instrumented launch."""
        cap = ((n + 63) // 64 + 64) * 4 * 4
        out = np.zeros(cap, np.uint64)
        _check(lib.rt_scene_wave_timeline(self._h, _ptr(rays_ptr), n, _ptr(out), cap))
        out = out.reshape(-1, 4)
        return out[out[:, 1] != 0]

    def count_fetches_device(self, rays_ptr, n):
        v = (C.c_uint64 * 6)()
        _check(lib.rt_scene_count_fetches_device(self._h, _ptr(rays_ptr), n, v))
        return dict(zip(["nodes", "tris", "leaves", "rays_hit", "rays", "max_stack"], [int(x) for x in v]))

    def unit_costs(self, rays_ptr, n):
        """Per work unit (8x8 tile with the image hint, else 64 records) traversal cost of one launch, as u32."""
        nu = C.c_size_t()
        _check(lib.rt_scene_unit_costs(self._h, _ptr(rays_ptr), n, None, 0, C.byref(nu)))
        out = np.zeros(nu.value, np.uint32)
        _check(lib.rt_scene_unit_costs(self._h, _ptr(rays_ptr), n, _ptr(out), len(out), C.byref(nu)))
        return out

    def unit_costs_primary(self, width, height, origin):
        """The same for the primary-ray table of a width x height image: (tiles_y, tiles_x) u32 of 8x8-tile costs."""
        o = (C.c_float * 3)(*[float(x) for x in origin])
        nu = C.c_size_t()
        _check(lib.rt_scene_unit_costs_primary(self._h, width, height, o, None, 0, C.byref(nu)))
        out = np.zeros(nu.value, np.uint32)
        _check(lib.rt_scene_unit_costs_primary(self._h, width, height, o, _ptr(out), len(out), C.byref(nu)))
        return out.reshape((height + 7) // 8, (width + 7) // 8)

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:
            lib.rt_scene_destroy(h)

    def __del__(self):
        self.close()


def scatter_tiles(device, src_ptr, src_records, tile_ids_ptr, tile_first_ptr, ntiles, tile, width, height, record_bytes, dst_ptr, stream=0):
    """rt_scatter_tiles on raw device addresses (the caller keeps the index arrays alive until the launch has run)."""
    _check(lib.rt_scatter_tiles(device, _ptr(src_ptr), src_records, _ptr(tile_ids_ptr), _ptr(tile_first_ptr), ntiles, tile, width, height,
                                record_bytes, _ptr(dst_ptr), C.c_void_p(stream) if stream else None))


def pack_bgra8(device, rgba_ptr, n, bgra_ptr, stream=0):
    """rt_pack_bgra8 on raw device addresses: n RGBA32F pixels -> n 8-bit BGRA pixels."""
    _check(lib.rt_pack_bgra8(device, _ptr(rgba_ptr), n, _ptr(bgra_ptr), C.c_void_p(stream) if stream else None))


def pack_hits8(device, hits16_ptr, n, hits8_ptr, stream=0):
    """rt_pack_hits8 on raw device addresses: n 16-byte {id,t,u,v} hits -> n 8-byte {id,t} records."""
    _check(lib.rt_pack_hits8(device, _ptr(hits16_ptr), n, _ptr(hits8_ptr), C.c_void_p(stream) if stream else None))


def composite_layers(device, layer0_ptr, layer1_ptr, n, rgba_out_ptr=None, bgra_out_ptr=None, stream=0):
    """rt_composite_layers on raw device addresses (layer1_ptr None = one layer)."""
    _check(lib.rt_composite_layers(device, _ptr(layer0_ptr), _ptr(layer1_ptr), n, _ptr(rgba_out_ptr), _ptr(bgra_out_ptr),
                                   C.c_void_p(stream) if stream else None))


class ShardPlan:
    """rt_shard_plan: which squares of a width x height image (cut into tile x tile squares) each of `nranks` ranks traces.
    tiles[r] = rank r's squares in the order dealt; first_record[r] = where its records start in the gather buffer."""

    def __init__(self, width, height, tile, nranks, unit_costs=None, square_costs=None, head_start=None):
        self.width, self.height, self.tile, self.nranks = width, height, tile, nranks
        tx, ty = (width + tile - 1) // tile, (height + tile - 1) // tile
        self.tile_order = np.zeros(tx * ty, np.uint32)
        self.first_tile = np.zeros(nranks + 1, np.uint64)
        self.first_record = np.zeros(nranks + 1, np.uint64)
        costs = None if unit_costs is None else np.ascontiguousarray(unit_costs, dtype=np.uint32)
        sq = None if square_costs is None else np.ascontiguousarray(square_costs, dtype=np.float64)
        assert sq is None or len(sq) == tx * ty
        hs = None if head_start is None else np.ascontiguousarray(head_start, dtype=np.float64)
        assert hs is None or len(hs) == nranks
        _check(lib.rt_shard_plan(width, height, tile, nranks, _ptr(costs), _ptr(sq), _ptr(hs), _ptr(self.tile_order), _ptr(self.first_tile),
                                 _ptr(self.first_record)))
        self.tiles = [self.tile_order[int(self.first_tile[r]):int(self.first_tile[r + 1])] for r in range(nranks)]

    def square_costs_of_rank(self, rank, unit_costs, image_width, out=None):
        """rt_shard_square_costs: adds rank's own work-unit costs (of a launch over its records with the given image hint)
        to a per-square array (all squares of the image; made here if not given)."""
        tx, ty = (self.width + self.tile - 1) // self.tile, (self.height + self.tile - 1) // self.tile
        out = np.zeros(tx * ty, np.float64) if out is None else out
        t = np.ascontiguousarray(self.tiles[rank])
        first = np.zeros(len(t), np.uint32)
        _check(lib.rt_shard_ray_indices(self.width, self.height, self.tile, _ptr(t), len(t), 0, None, _ptr(first)))
        u = np.ascontiguousarray(unit_costs, dtype=np.uint32).reshape(-1)
        _check(lib.rt_shard_square_costs(_ptr(u), len(u), image_width, self.records(rank), _ptr(t), _ptr(first), len(t), _ptr(out), len(out)))
        return out

    def records(self, rank):
        return int(self.first_record[rank + 1] - self.first_record[rank])

    def ray_indices(self, rank):
        """Row-major ray / pixel indices of rank's records, square after square."""
        idx = np.zeros(self.records(rank), np.uint32)
        t = np.ascontiguousarray(self.tiles[rank])
        _check(lib.rt_shard_ray_indices(self.width, self.height, self.tile, _ptr(t), len(t), 0, _ptr(idx), None))
        return idx

    def scatter_tables(self):
        """(tile ids, first record of each square in the gather buffer) for all ranks' squares: what rt_scatter_tiles takes."""
        first = np.zeros(len(self.tile_order), np.uint32)
        _check(lib.rt_shard_ray_indices(self.width, self.height, self.tile, _ptr(self.tile_order), len(self.tile_order), 0, None, _ptr(first)))
        return self.tile_order.copy(), first


class Comm:
    """RCCL communicator made through the library (rt_comm_unique_id / rt_comm_create)."""

    def __init__(self, comm_id, nranks, rank, device):
        h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(comm_id))
        _check(lib.rt_comm_create(buf, nranks, rank, device, C.byref(h)))
        self._h, self.nranks, self.rank, self.device = h, nranks, rank, device

    def sum_f64(self, values):
        """In place: values[i] := sum over the ranks (rt_comm_sum_f64; a collective)."""
        assert values.dtype == np.float64 and values.flags.c_contiguous
        _check(lib.rt_comm_sum_f64(self._h, self.nranks, self.device, _ptr(values), values.size))
        return values

    def check(self):
        """A real exchange around the ring of ranks and a sum over them (rt_comm_check; a collective)."""
        _check(lib.rt_comm_check(self._h, self.nranks, self.rank, self.device))

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        _check(lib.rt_comm_unique_id(buf))
        return bytes(buf)

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:
            lib.rt_comm_destroy(h)

    def __del__(self):
        self.close()


class TiledFrame:
    """rt_tiled_*: one rank's part of a frame rendered by `nranks` processes, one per GPU (the deal of the image squares, this
    rank's rays, the exchange buffers).  `frame_rays` is the frame's whole ray table (W*H records); the scene gets the squares'
    width as its "image_width" and should be used by this object only."""

    def __init__(self, scene, comm, rank, nranks, width, height, tile, frame_rays, device, root_head_start=0.0):
        assert frame_rays.dtype == np.float32 and frame_rays.flags.c_contiguous and frame_rays.shape == (width * height, RAY_FLOATS)
        h = C.c_void_p()
        _check(lib.rt_tiled_create(scene._h, comm._h if comm is not None else None, rank, nranks, width, height, tile, _ptr(frame_rays), device,
                                   float(root_head_start), C.byref(h)))
        self._h, self.scene, self.comm, self.rank, self.nranks = h, scene, comm, rank, nranks

    def info(self):
        v = (C.c_uint64 * 4)()
        _check(lib.rt_tiled_info(self._h, v))
        return {"rays": int(v[0]), "squares": int(v[1]), "squares_of_frame": int(v[2]), "image_width": int(v[3])}

    def render(self, cam, framebuffer_ptr=None, stream=0):
        """One frame, enqueued on `stream` (a collective: every rank calls it); rank 0's framebuffer receives W*H BGRA8 pixels."""
        m = np.ascontiguousarray(cam, dtype=np.float32).reshape(16)
        _check(lib.rt_tiled_render(self._h, _ptr(m), _ptr(framebuffer_ptr), C.c_void_p(stream) if stream else None))

    def redeal(self, cam):
        """Deals the squares again from the traversal costs measured under `cam` (a collective, synchronous); returns the number
        of squares that changed rank."""
        m = np.ascontiguousarray(cam, dtype=np.float32).reshape(16)
        moved = C.c_size_t(0)
        _check(lib.rt_tiled_redeal(self._h, _ptr(m), C.byref(moved)))
        return int(moved.value)

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:
            lib.rt_tiled_destroy(h)

    def __del__(self):
        self.close()


def gather_scatter(comm, rank, nranks, root, device, mine_ptr, gather_ptr, first_record, record_bytes, tile_ids_ptr, tile_first_ptr, ntiles,
                   tile, width, height, framebuffer_ptr, stream=0):
    """rt_gather_scatter: every rank's records to root over RCCL (one grouped point-to-point call), then the device scatter."""
    fr = np.ascontiguousarray(first_record, dtype=np.uint64)
    _check(lib.rt_gather_scatter(comm._h if comm is not None else None, rank, nranks, root, device, _ptr(mine_ptr), _ptr(gather_ptr), _ptr(fr),
                                 record_bytes, _ptr(tile_ids_ptr), _ptr(tile_first_ptr), ntiles, tile, width, height, _ptr(framebuffer_ptr),
                                 C.c_void_p(stream) if stream else None))


def debug_tri_batch(tri, ray, device=0):
    """Runs the kernel's triangle test on n (triangle, ray) pairs: tri (n,3,4) f32, ray (n,8) f32."""
    tri = np.ascontiguousarray(tri, dtype=np.float32).reshape(-1, 3, 4)
    ray = np.ascontiguousarray(ray, dtype=np.float32).reshape(-1, 8)
    n = len(ray)
    hit = np.zeros(n, np.int32)
    t = np.zeros(n, np.float32)
    uv = np.zeros((n, 2), np.float32)
    _check(lib.rt_debug_tri_batch(device, _ptr(tri), _ptr(ray), n, _ptr(hit), _ptr(t), _ptr(uv)))
    return hit, t, uv


def write_bmp32(path, bgra):
    bgra = np.ascontiguousarray(bgra, dtype=np.uint8)
    h, w = bgra.shape[0], bgra.shape[1]
    _check(lib.rt_write_bmp32(os.fsencode(path), _ptr(bgra), w, h))


def write_pfm(path, rgba):
    rgba = np.ascontiguousarray(rgba, dtype=np.float32)
    h, w = rgba.shape[0], rgba.shape[1]
    _check(lib.rt_write_pfm(os.fsencode(path), _ptr(rgba), w, h))


def load_scene(model, loader="tolerant", bvh_mode="fixed", device=0, width=2):
    """Convenience: mesh + BVH + device scene for a model under assets/models."""
    path = model if os.path.exists(model) else os.path.join(MODELS_DIR, model)
    mesh = Mesh.load_obj(path, loader)
    bvh = BVH(mesh, bvh_mode)
    return mesh, bvh, Scene(mesh, bvh, device=device, width=width)
