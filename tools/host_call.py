#!/usr/bin/env python3
"""Times ray_scene_intersection with HOST ray and hit buffers (the PCIe-inclusive drop-in call)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytrace_amd as rt

mesh = rt.Mesh.load_obj(os.path.join(rt.MODELS_DIR, "dragon.obj"), "tolerant")
bvh = rt.BVH(mesh, "fixed")
scene = rt.Scene(mesh, bvh, device=0)
W, H = 1920, 1080
scene.set_param("image_width", W)
rays = rt.primary_rays(W, H, rt.frame_camera(bvh.root_box, W, H))
hits = np.zeros_like(rays)     # allocated and touched once: the timed calls pay for the transfer, not for page faults
for _ in range(3):
    rt._check(rt.lib.ray_scene_intersection(rt._ptr(rays), len(rays), scene._h, rt._ptr(hits)))
ts = []
for _ in range(10):
    t0 = time.perf_counter()
    rt._check(rt.lib.ray_scene_intersection(rt._ptr(rays), len(rays), scene._h, rt._ptr(hits)))
    ts.append(time.perf_counter() - t0)
t = float(np.median(ts))
want = scene.intersect(rays)
assert np.array_equal(hits.view(np.uint32), want.view(np.uint32))
print(f"host-pointer call, dragon 1080p: median {t*1e3:.2f} ms = {len(rays)/t/1e6:.1f} Mrays/s "
      f"({2*64*len(rays)/t/1e9:.1f} GB/s over PCIe, pageable host memory both ways)")
