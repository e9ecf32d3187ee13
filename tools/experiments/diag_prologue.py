"""Reader for a one-off instrumented build of the trace kernel that stamps, per wave of the instrumented launch, the time of
its first instruction, of the moment its dispatch-order entry is known, of the arrival of its ray records and of its end
(the last two stamps replace the hardware ids in the wave-timeline record; the kernel patch is not kept in the tree).
Prints the mean duration of each stage, dragon 1080p."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import raytrace_amd as rt
W, H = 1920, 1080
mesh = rt.Mesh.load_obj(os.path.join(rt.MODELS_DIR, "dragon.obj"), "tolerant")
bvh = rt.BVH(mesh, "fixed")
scene = rt.Scene(mesh, bvh, device=0, width=4)
for k, v in {"width": 4, "image_width": W, "feedback": 1}.items():
    scene.set_param(k, v)
rays = rt.primary_rays(W, H, rt.frame_camera(bvh.root_box, W, H))
n = len(rays)
d = torch.from_numpy(rays).cuda()
out = torch.empty((n, 16), dtype=torch.float32, device="cuda")
for _ in range(40):
    scene.intersect_async(d.data_ptr(), n, out.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
cap = ((n + 63) // 64 + 1024) * 4 * 4
for _ in range(3):
    raw = np.zeros(cap, np.uint64)
    rt._check(rt.lib.rt_scene_wave_timeline(scene._h, rt._ptr(d.data_ptr()), n, rt._ptr(raw), cap))
tl = raw.reshape(-1, 4).astype(np.float64)
used = (tl[:, 1] != 0) & (tl[:, 3] != 0)
t0, t1, to, tr = tl[used, 0], tl[used, 1], tl[used, 2], tl[used, 3]
print("waves", int(used.sum()), "life mean %.2f us" % ((t1 - t0).mean() / 100))
print("start -> order entry known: mean %.3f p50 %.3f p90 %.3f us" % ((to - t0).mean() / 100, np.median(to - t0) / 100, np.quantile(to - t0, .9) / 100))
print("order entry -> rays arrived: mean %.3f p50 %.3f p90 %.3f us" % ((tr - to).mean() / 100, np.median(tr - to) / 100, np.quantile(tr - to, .9) / 100))
print("rays arrived -> end: mean %.3f us" % ((t1 - tr).mean() / 100))
start = (t0 - t0.min()) / 100
for a, b in ((0, 20), (20, 60), (60, 100), (100, 200)):
    s = (start >= a) & (start < b)
    if s.sum(): print(f"  waves starting in [{a},{b}) us: {int(s.sum())}  order {((to - t0)[s]).mean() / 100:.3f}  rays {((tr - to)[s]).mean() / 100:.3f} us")
