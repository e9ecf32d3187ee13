#!/usr/bin/env python3
"""Interleaved A/B timing of scene parameter sets in ONE process (same device, same data):
   tools/ab.py [--workload dragon1080] [--rounds 7] [--steps 40] "k=v,k=v" "k=v" ...
Prints median / min kernel time per configuration and Mrays/s at the median."""
import argparse
import os
import statistics
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytrace_amd as rt  # noqa: E402

WORK = {"dragon1080": ("dragon.obj", 1920, 1080), "bunny1080": ("bunny.obj", 1920, 1080),
        "dragon4k": ("dragon.obj", 3840, 2160), "dragon1440": ("dragon.obj", 2560, 1440), "dragon1620": ("dragon.obj", 2880, 1620), "ao": ("bunny_big.obj", 0, 0), "teapot1080": ("teapot.obj", 1920, 1080)}
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="dragon1080")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--rays", type=int, default=8 << 20)
ap.add_argument("--out", default="full")
ap.add_argument("--bvh", default="sah")
ap.add_argument("configs", nargs="+")
a = ap.parse_args()

model, W, H = WORK[a.workload]
mesh = rt.Mesh.load_obj(os.path.join(rt.MODELS_DIR, model), "tolerant")
bvh = rt.BVH(mesh, a.bvh)
scene = rt.Scene(mesh, bvh, device=0)
if W:
    rays = rt.primary_rays(W, H, rt.frame_camera(bvh.root_box, W, H))
else:
    rays = rt.ao_rays(mesh, bvh.root_box, 1234, 0, a.rays)
n = len(rays)
d_rays = torch.from_numpy(rays).cuda()
d_hits = torch.empty((n, 16 if a.out == "full" else 4), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
DEFAULTS = {"width": 2, "quantized": 1, "block_threads": 64, "image_width": 0, "extra_stack": 0, "leaf_pct": 50, "centre_first": 1, "feedback": 0, "waves_per_cu": 0, "feedback_radius": 1, "feedback_share": 96}


def apply(cfg):
    p = dict(DEFAULTS)
    for kv in cfg.split(","):
        if kv:
            k, v = kv.split("=")
            p[k] = int(v) if k != "image_width" or v != "W" else W
    for k, v in p.items():
        scene.set_param(k, v)


def run(steps):
    for _ in range(steps):
        if a.out == "full":
            scene.intersect_async(d_rays.data_ptr(), n, d_hits.data_ptr(), None, stream)
        else:
            scene.intersect_async(d_rays.data_ptr(), n, None, d_hits.data_ptr(), stream)


times = {c: [] for c in a.configs}
ref = None
for c in a.configs:       # warm-up + cross-check that every configuration returns the same hits
    apply(c)
    run(3)
    torch.cuda.synchronize()
    h = d_hits.cpu().numpy().view(np.uint32)
    if ref is None:
        ref = h.copy()
    elif not np.array_equal(ref, h):
        print(f"WARNING: configuration '{c}' differs from '{a.configs[0]}' in {int((ref != h).any(axis=1).sum())} records")
for _ in range(a.rounds):
    for c in a.configs:
        apply(c)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(a.steps)
        e1.record()
        torch.cuda.synchronize()
        times[c].append(e0.elapsed_time(e1) / a.steps)
for c in a.configs:
    med, mn = statistics.median(times[c]), min(times[c])
    print(f"{c:50s} median {med*1e3:8.1f} us  min {mn*1e3:8.1f} us  {n/med/1e3:9.1f} Mrays/s", flush=True)
