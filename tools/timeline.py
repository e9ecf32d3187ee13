#!/usr/bin/env python3
"""Occupancy-over-time of one instrumented launch: tools/timeline.py [k=v,...]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytrace_amd as rt

cfg = sys.argv[1] if len(sys.argv) > 1 else ""
mesh = rt.Mesh.load_obj(os.path.join(rt.MODELS_DIR, "dragon.obj"), "tolerant")
bvh = rt.BVH(mesh, os.environ.get("RT_BVH", "fixed"))
scene = rt.Scene(mesh, bvh, device=0)
W, H = 1920, 1080
for kv in cfg.split(","):
    if kv:
        k, v = kv.split("=")
        scene.set_param(k, W if v == "W" else int(v))
rays = rt.primary_rays(W, H, rt.frame_camera(bvh.root_box, W, H))
d = torch.from_numpy(rays).cuda()
out = torch.empty((len(rays), 16), dtype=torch.float32, device="cuda")
for _ in range(4):   # lets a cost-feedback order settle before the instrumented launch follows it
    scene.intersect_async(d.data_ptr(), len(rays), out.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
for _ in range(4):
    tl = scene.wave_timeline(d.data_ptr(), len(rays))
long = (tl[:, 1] - tl[:, 0]) > 100  # > 1 us: waves that actually traced
print('slots', len(tl), 'of which traced', int(long.sum()))
t0 = tl[:, 0].min()
start = (tl[:, 0] - t0).astype(np.float64) / 100.0   # us
end = (tl[:, 1] - t0).astype(np.float64) / 100.0
life = end - start
print(f"{cfg}: waves {len(tl)}, span {end.max():.1f} us, wave lifetime mean {life.mean():.1f} us p50 {np.median(life):.1f} p99 {np.quantile(life, .99):.1f} max {life.max():.1f}")
print(f"  wave-time sum {life.sum():.0f} us -> mean concurrency {life.sum()/end.max():.0f} waves ({life.sum()/end.max()/256:.1f} per CU)")
cu = (tl[:, 3].astype(np.int64) << 16) | ((tl[:, 2].astype(np.int64) >> 8) & 0xFF)
print(f"  distinct (xcc, se/cu) keys: {len(np.unique(cu))}; xcc ids {np.unique(tl[:,3]).tolist()}")
edges = np.linspace(0, end.max(), 21)
conc = [int(((start < b) & (end > a)).sum()) for a, b in zip(edges[:-1], edges[1:])]
print("  waves alive per 5% time slice:", conc)
last = [float(end[tl[:, 3] == x].max()) for x in np.unique(tl[:, 3])]
print("  last wave end per XCC (us):", [round(x, 1) for x in last])
order = np.argsort(start)
top = np.argsort(-life)[:8]
print("  longest waves (start us, life us):", [(round(float(start[i]), 1), round(float(life[i]), 1)) for i in top])
print("  launch rate: first 10%% of waves started within %.1f us; median start %.1f us; last start %.1f us" % (start[order][len(order)//10], np.median(start), start.max()))
