#!/usr/bin/env python3
"""Distribution of the per-tile traversal cost (wave iterations) of a primary-ray frame:
   tools/cost_hist.py [model] [W] [H]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytrace_amd as rt  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "dragon.obj"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
mesh = rt.Mesh.load_obj(os.path.join(rt.MODELS_DIR, model), "tolerant")
bvh = rt.BVH(mesh, "fixed")
scene = rt.Scene(mesh, bvh, device=0, width=4)
c = np.asarray(scene.unit_costs_primary(W, H, rt.frame_camera(bvh.root_box, W, H))).ravel().astype(np.int64)
s = np.sort(c)[::-1]
print(f"{model} {W}x{H}: {c.size} tiles, cost sum {c.sum()}, mean {c.mean():.1f}, max {s[0]}")
for q in (0.5, 0.9, 0.95, 0.99, 0.995, 0.999):
    print(f"  quantile {q}: {np.quantile(c, q):.0f}")
for k in (8, 32, 64, 128, 256, 512, 1024):
    print(f"  top {k}: cost >= {s[k - 1]}, share of all iterations {s[:k].sum() / c.sum():.4f}")
