"""Generates tests/golden/*.npz from the REFERENCE's own arithmetic.

Run where /root/reference exists (after `make -C oracle`): every triangle test below is the
reference's intersect_ray_triangle (core/ray_tri.cpp:6-53) compiled into oracle/_ref/libref.so,
driven in index order with a strict `<` (cpu_raycast/raycast.cpp:113-122); ray directions come
from the reference's normalize (core/raymath.cpp:6-15) via the repo's primary-ray table, which
tests/test_oracle.py pins to it.  Output per case: the rays' origin, image size, and id/t/u/v.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
import raytrace_amd as rt  # noqa: E402  (host-side loader / ray table only; nothing runs on a GPU)

CASES = [("cube.obj", "tolerant", 64, 64), ("cube_mirror_test.obj", "reference", 64, 64),
         ("teapot.obj", "reference", 64, 64), ("sphere.obj", "reference", 48, 32)]

for model, loader, W, H in CASES:
    mesh = rt.Mesh.load_obj(os.path.join(rt.MODELS_DIR, model), loader)
    bvh = rt.BVH(mesh, "reference")
    origin = rt.frame_camera(bvh.root_box, W, H)
    rays = rt.primary_rays(W, H, origin)
    ids, t, u, v = oracle.ref_closest_hit(mesh.verts, mesh.faces, rays)
    out = os.path.join(ROOT, "tests", "golden", model.replace(".obj", f"_{W}x{H}.npz"))
    np.savez_compressed(out, model=model, loader=loader, width=W, height=H, origin=origin,
                        ids=ids, t=t, u=u, v=v)
    print(out, "hits:", int((ids != 0xFFFFFFFF).sum()), "of", len(ids))
