#!/usr/bin/env python3
"""Lane-refill kernel against the batch kernel on incoherent rays: same bits, and the time of each."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytrace_amd as rt

mesh = rt.Mesh.load_obj(os.path.join(rt.MODELS_DIR, "bunny_big.obj"), "tolerant")
bvh = rt.BVH(mesh, "fixed")
scene = rt.Scene(mesh, bvh, device=0, width=4)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4 << 20
rays = torch.from_numpy(rt.ao_rays(mesh, bvh.root_box, 1234, 0, n)).cuda()
stream = torch.cuda.current_stream().cuda_stream
for width, label in ((4, "16-byte hits"), (16, "64-byte records")):
    base = None
    for refill in [0] + [int(x) for x in sys.argv[2:]]:
        scene.set_param("refill", refill)
        out = torch.full((n, width), -5.0, dtype=torch.float32, device="cuda")
        args = (None, out.data_ptr()) if width == 4 else (out.data_ptr(), None)
        for _ in range(3):
            scene.intersect_async(rays.data_ptr(), n, args[0], args[1], stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            scene.intersect_async(rays.data_ptr(), n, args[0], args[1], stream)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        if base is None:
            base = out
        same = bool(torch.equal(out.view(torch.int32), base.view(torch.int32)))
        print(f"{label}, refill {refill:2d}: {ms:.3f} ms  {n / ms / 1e3:.0f} Mrays/s  equals batch kernel: {same}", flush=True)
        if out is not base:
            del out
