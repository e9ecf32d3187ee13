#!/usr/bin/env python3
"""Per-launch durations of consecutive rt_intersect_async launches on one stream (an event between every two launches), to see
what the launches that record costs and the order build cost, and how the first launches of a scene differ from the later
ones: tools/per_launch.py [warm=N] [n=N] [k=v ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytrace_amd as rt  # noqa: E402

W, H = 1920, 1080
mesh = rt.Mesh.load_obj(os.path.join(rt.MODELS_DIR, "dragon.obj"), "tolerant")
bvh = rt.BVH(mesh, "fixed")
scene = rt.Scene(mesh, bvh, device=0, width=4)
scene.set_param("width", 4)
scene.set_param("image_width", W)
WARM, N, SPIN_MS = 40, 96, 0
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    if k == "spin":
        SPIN_MS = int(v)   # unrelated GPU work before the first launch (a large matrix product repeated for about this long)
    elif k == "warm":
        WARM = int(v)      # launches before the timed ones (bench.py --warmup)
    elif k == "n":
        N = int(v)
    else:
        scene.set_param(k, int(v))
box = bvh.root_box
tables = []
for k in range(4):
    o = rt.frame_camera(box, W, H)
    o[0] += k * 0.002 * float(box[1, 0] - box[0, 0])
    tables.append(torch.from_numpy(rt.primary_rays(W, H, o)).cuda())
hits = torch.empty((W * H, 16), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
if SPIN_MS:
    import time
    a_ = torch.randn(4096, 4096, device="cuda")
    t_ = time.perf_counter()
    while (time.perf_counter() - t_) * 1e3 < SPIN_MS:
        b_ = a_ @ a_
        torch.cuda.synchronize()
for i in range(WARM):
    scene.intersect_async(tables[i % 4].data_ptr(), W * H, hits.data_ptr(), None, stream)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
ev[0].record()
for i in range(N):
    scene.intersect_async(tables[(WARM + i) % 4].data_ptr(), W * H, hits.data_ptr(), None, stream)
    ev[i + 1].record()
torch.cuda.synchronize()
d = np.array([ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(N)])
print("per-launch us (an event record between launches adds a few us to each):")
for r in range(0, N, 8):
    print(" ".join(f"{x:6.1f}" for x in d[r:r + 8]))
print("mean by position in the period of 8:", " ".join(f"{d[k::8].mean():6.1f}" for k in range(8)))
print("mean", round(float(d.mean()), 2))
scene.close()
