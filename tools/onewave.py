#!/usr/bin/env python3
"""Latency of single heavy waves that have the GPU to themselves: traces the 8x8 tile with the longest
ray (and a few lighter ones) as a 64-ray launch and divides the kernel time by the wave's step count from
the lockstep model.  tools/onewave.py [k=v,...]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytrace_amd as rt

mesh = rt.Mesh.load_obj(os.path.join(rt.MODELS_DIR, "dragon.obj"), "tolerant")
bvh = rt.BVH(mesh, os.environ.get("RT_BVH", "fixed"))
scene = rt.Scene(mesh, bvh, device=0, width=4)
for kv in (sys.argv[1] if len(sys.argv) > 1 else "").split(","):
    if kv:
        k, v = kv.split("=")
        scene.set_param(k, int(v))
W, H = 1920, 1080
rays = rt.primary_rays(W, H, rt.frame_camera(bvh.root_box, W, H))
_, per = scene.count_fetches(rays, per_ray=True)
steps = (per[:, 0].astype(int) + per[:, 1]).reshape(H // 8, 8, W // 8, 8)
tile_max = steps.max(axis=(1, 3)).reshape(-1)
order = np.argsort(-tile_max)
stream = torch.cuda.current_stream().cuda_stream
for rank in (0, 1, 10, 100, 1000):
    ty, tx = divmod(int(order[rank]), W // 8)
    idx = ((ty * 8 + np.arange(8))[:, None] * W + tx * 8 + np.arange(8)[None, :]).reshape(-1)
    sub = np.ascontiguousarray(rays[idx])
    m = scene.wave_model(sub, image_width=0, policy=50)
    d = torch.from_numpy(sub).cuda()
    out = torch.empty((64, 16), dtype=torch.float32, device="cuda")
    for _ in range(5):
        scene.intersect_async(d.data_ptr(), 64, out.data_ptr(), None, stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(20):
        e0.record(); scene.intersect_async(d.data_ptr(), 64, out.data_ptr(), None, stream); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    t = float(np.median(ts))
    print(f"tile rank {rank}: longest ray {int(tile_max[order[rank]])} steps, wave steps {m['max_wave_steps']}, "
          f"kernel {t:.1f} us -> {t / max(1, m['max_wave_steps']) * 1e3:.0f} ns per wave step")
