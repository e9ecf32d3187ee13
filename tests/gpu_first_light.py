"""First-light GPU check (run by hand: python tests/gpu_first_light.py).  Not collected by pytest."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytrace_amd as rt
import oracle

def check(model, W, H, stride, width, persistent):
    mesh, bvh, scene = rt.load_scene(model, width=width)
    scene.set_param("persistent", persistent)
    org = rt.frame_camera(bvh.root_box, W, H)
    rays = rt.primary_rays(W, H, org)
    t0 = time.time(); hits = scene.intersect(rays); dt = time.time() - t0
    ids, t, u, v = rt.hit_fields(hits)
    sel = np.arange(0, len(rays), stride)
    oid, ot, ou, ov = oracle.closest_hit(mesh.verts, mesh.faces, rays[sel])
    bad = int((ids[sel] != oid).sum())
    hitm = oid != rt.MISS_ID
    tb = int((t[sel][hitm].view(np.uint32) != ot[hitm].view(np.uint32)).sum())
    ub = int((u[sel][hitm].view(np.uint32) != ou[hitm].view(np.uint32)).sum())
    print(f"{model} {W}x{H} width={width} persistent={persistent}: rays={len(rays)} hit={int((ids!=rt.MISS_ID).sum())} "
          f"checked={len(sel)} id_mismatch={bad} t_bits_diff={tb} u_bits_diff={ub} host_call={dt*1e3:.1f} ms", flush=True)
    return bad == 0

ok = True
for width in (2, 4):
    for pers in (1, 0):
        ok &= check("cube.obj", 128, 128, 1, width, pers)
        ok &= check("teapot.obj", 256, 256, 7, width, pers)
        ok &= check("dragon.obj", 1920, 1080, 2003, width, pers)
print("OK" if ok else "FAIL")
sys.exit(0 if ok else 1)
