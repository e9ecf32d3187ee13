#!/usr/bin/env python3
"""Scene creation time with the records built on the host and on the device (dragon, centroid tree)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytrace_amd as rt
mesh = rt.Mesh.load_obj(os.path.join(rt.MODELS_DIR, "dragon.obj"), "tolerant")
bvh = rt.BVH(mesh, "fixed")
for dev_flat in (False, True, False, True):
    t0 = time.perf_counter()
    s = rt.Scene(mesh, bvh, device=0, width=4, flatten_on_device=dev_flat)
    dt = time.perf_counter() - t0
    s.close()
    print(f"rt_scene_create, dragon, records built on the {'device' if dev_flat else 'host'}: {dt*1e3:.1f} ms")
