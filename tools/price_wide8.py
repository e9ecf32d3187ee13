#!/usr/bin/env python3
"""Prices an eight-wide (128-byte, quantized) node against the four-wide one the kernel walks, without a GPU: the lockstep wave
model (rt_scene_wave_model) run over the same rays with the tree collapsed three binary levels per node instead of two
("model_width" 8).  Prints, per workload: wave-level node steps, triangle steps, loop iterations in total and of the longest
wave, and the lane utilisation of the node steps.   usage: tools/price_wide8.py [max_waves]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytrace_amd as rt

max_waves = int(sys.argv[1]) if len(sys.argv) > 1 else 0
out = {}
for model, W, H in (("dragon.obj", 1920, 1080), ("bunny.obj", 1920, 1080)):
    mesh = rt.Mesh.load_obj(os.path.join(rt.MODELS_DIR, model), "tolerant")
    bvh = rt.BVH(mesh, "fixed")
    scene = rt.Scene(mesh, bvh, device=-1, width=4)
    leaf_pct = 50
    rays = rt.primary_rays(W, H, rt.frame_camera(bvh.root_box, W, H))
    row = {}
    for mw in (0, 8):
        scene.set_param("model_width", mw)
        m = scene.wave_model(rays, image_width=W, max_waves=max_waves, policy=leaf_pct)
        row["wide8" if mw else "wide4"] = {
            "waves": m["waves"], "node_steps_per_wave": round(m["wave_node_steps"] / m["waves"], 2),
            "tri_steps_per_wave": round(m["wave_tri_steps"] / m["waves"], 2),
            "iterations_per_wave": round(m["iterations"] / m["waves"], 2), "longest_wave_iterations": m["max_wave_iterations"],
            "longest_wave_steps": m["max_wave_steps"],
            "node_step_lane_utilisation": round(m["lane_node_steps"] / max(1, m["wave_node_steps"]) / 64, 3),
            "box_tests_per_ray": round(m["lane_node_steps"] * (8 if mw else 4) / m["rays"], 1)}
    a, b = row["wide4"], row["wide8"]
    row["wide8_over_wide4"] = {k: round(b[k] / a[k], 3) for k in ("node_steps_per_wave", "tri_steps_per_wave", "iterations_per_wave",
                                                                   "longest_wave_iterations", "longest_wave_steps", "box_tests_per_ray")}
    out[f"{model} {W}x{H}"] = row
    scene.close()
print(json.dumps(out, indent=1))
