#!/usr/bin/env python3
"""How much would re-ordering incoherent rays buy?  Traces the AO-style ray set as generated and after a host-side
sort by (direction octant, Morton code of the origin), and prints both rates.  Planning aid."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytrace_amd as rt

def part1by2(x):
    x = x.astype(np.uint64) & 0x3FF
    x = (x | (x << 16)) & 0x30000FF
    x = (x | (x << 8)) & 0x300F00F
    x = (x | (x << 4)) & 0x30C30C3
    x = (x | (x << 2)) & 0x9249249
    return x

mesh = rt.Mesh.load_obj(os.path.join(rt.MODELS_DIR, "bunny_big.obj"), "tolerant")
bvh = rt.BVH(mesh, "fixed")
scene = rt.Scene(mesh, bvh, device=0, width=4)
n = 8 << 20
rays = rt.ao_rays(mesh, bvh.root_box, 1234, 0, n)
box = bvh.root_box
q = ((rays[:, 0:3] - box[0, :3]) / (box[1, :3] - box[0, :3]) * 1023).clip(0, 1023).astype(np.uint32)
morton = part1by2(q[:, 0]) | (part1by2(q[:, 1]) << 1) | (part1by2(q[:, 2]) << 2)
octant = ((rays[:, 4] < 0).astype(np.uint64) | ((rays[:, 5] < 0).astype(np.uint64) << 1) | ((rays[:, 6] < 0).astype(np.uint64) << 2))
orders = {"as generated": np.arange(n), "octant, morton(origin)": np.argsort((octant << 30) | morton, kind="stable"),
          "morton(origin), octant": np.argsort((morton << 3) | octant, kind="stable"),
          "morton(origin >> 9 bits), octant, morton": np.argsort(((morton >> 9) << 33) | (octant << 30) | morton, kind="stable")}
stream = torch.cuda.current_stream().cuda_stream
out = torch.empty((n, 16), dtype=torch.float32, device="cuda")
for name, perm in orders.items():
    d = torch.from_numpy(np.ascontiguousarray(rays[perm])).cuda()
    for fb in (0,):
        scene.set_param("feedback", fb)
        for _ in range(3):
            scene.intersect_async(d.data_ptr(), n, out.data_ptr(), None, stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            scene.intersect_async(d.data_ptr(), n, out.data_ptr(), None, stream)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{name:45s} {ms:.3f} ms  {n / ms / 1e3:.0f} Mrays/s", flush=True)
