#!/usr/bin/env python3
"""Wave timeline of one instrumented launch, split by kind of wave: tools/timeline2.py [k=v,...] [model W H]
Prints, for the quarter waves of the split units (the first 4 x split_units dispatch slots) and for the whole-unit waves,
how long they live and when the last one ends, and the occupancy of the launch over time."""
import ctypes as C
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytrace_amd as rt

cfg = sys.argv[1] if len(sys.argv) > 1 else ""
model = sys.argv[2] if len(sys.argv) > 2 else "dragon.obj"
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
H = int(sys.argv[4]) if len(sys.argv) > 4 else 1080
mesh = rt.Mesh.load_obj(os.path.join(rt.MODELS_DIR, model), "tolerant")
bvh = rt.BVH(mesh, os.environ.get("RT_BVH", "fixed"))
scene = rt.Scene(mesh, bvh, device=0, width=4)
params = {"width": 4, "image_width": W, "feedback": 1, "split_units": 32}
for kv in cfg.split(","):
    if kv:
        k, v = kv.split("=")
        params[k] = int(v)
for k, v in params.items():
    scene.set_param(k, v)
rays = rt.primary_rays(W, H, rt.frame_camera(bvh.root_box, W, H))
n = len(rays)
d = torch.from_numpy(rays).cuda()
out = torch.empty((n, 16), dtype=torch.float32, device="cuda")
for _ in range(12):
    scene.intersect_async(d.data_ptr(), n, out.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
cap = ((n + 63) // 64 + 64) * 4 * 4
for _ in range(3):
    raw = np.zeros(cap, np.uint64)
    rt._check(rt.lib.rt_scene_wave_timeline(scene._h, rt._ptr(d.data_ptr()), n, rt._ptr(raw), cap))
tl = raw.reshape(-1, 4)
S = params["split_units"]
used = tl[:, 1] != 0
t0 = tl[used, 0].min()
start = (tl[:, 0].astype(np.float64) - float(t0)) / 100.0
end = (tl[:, 1].astype(np.float64) - float(t0)) / 100.0
life = end - start
span = end[used].max()
print(f"{model} {W}x{H} {cfg}: {int(used.sum())} waves, span {span:.1f} us, wave-time sum {life[used].sum():.0f} us -> "
      f"mean concurrency {life[used].sum() / span / 256:.1f} of 28 waves per CU")
q = np.zeros(len(tl), bool)
q[:4 * S] = True
for name, sel in (("quarter waves of split units", q & used), ("whole-unit waves", ~q & used)):
    if sel.sum():
        l = life[sel]
        print(f"  {name:30s}: {int(sel.sum()):6d} waves, life mean {l.mean():6.1f} p50 {np.median(l):6.1f} p99 {np.quantile(l, .99):6.1f} max {l.max():6.1f} us; "
              f"last start {start[sel].max():6.1f}, last end {end[sel].max():6.1f} us")
edges = np.linspace(0, span, 21)
print("  waves alive per 5% time slice:", [int(((start[used] < b) & (end[used] > a)).sum()) for a, b in zip(edges[:-1], edges[1:])])
top = np.argsort(-np.where(used, life, 0))[:10]
print("  longest waves (slot, start us, life us):", [(int(i), round(float(start[i]), 1), round(float(life[i]), 1)) for i in top])
