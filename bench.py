#!/usr/bin/env python3
"""bench.py -- closest-hit Mrays/s on MI355X.

A "step" is one pass of the hot path (ray_scene_intersection's device form, rt_intersect_async) over one
batch of rays that is already resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] ...

Workloads:
  dragon1080   dragon.obj 1920x1080 primary rays           (default with one GPU; the headline metric)
  tiles4k      dragon.obj 3840x2160, image tiles sharded over the ranks by measured traversal cost; every step
               ends with the compact framebuffer gathered to rank 0 over RCCL and scattered into the row-major
               image on the device (default with N > 1: strong scaling, the gather and scatter are inside the
               timed step and also reported on their own)
  bunny1080    bunny.obj 1920x1080 primary rays
  dragon4k     dragon.obj 3840x2160 primary rays on one GPU
  ao           bunny_big.obj incoherent AO-style rays (--rays per GPU, default 64 Mi)
  mirror5      cube_mirror_test.obj and teapot.obj 1920x1080: one step = for each scene the primary pass + four
               reflection passes of the full `trace` kernel (rt_trace_pass, records ping-ponged on the device);
               rays counted = those actually traced (a record whose previous pass missed is passed through)
  frames       with N > 1: every rank traces its own dragon1080 frame (weak scaling, no exchange)

The scene is built the way the reference builds it: tolerant OBJ load, core/CentroidBVH build with the
centroid-index repair (--bvh fixed).  --bvh sah swaps in the library's surface-area builder.

Before the W warm-up steps the card is brought to the clocks it holds under sustained load (--preheat-ms of dense
matrix products, nothing of the hot path; see preheat()): the line names the duration (`gpu_preheat_ms`) and carries the same
W + K steps timed on the idle card (`idle_gpu`).

Rank 0 prints ONE JSON line.  `roofline` is the byte roofline of the trace kernel: `achieved` = the bytes that have to
cross HBM per launch (the ray and hit streams) over the launch time measured live with HIP events, `peak` = the 8 TB/s
HBM3E peak, `traffic` = the HBM-side bytes the PMC counters saw per launch.  Beside it: `algorithmic_bytes` (streams +
every node and triangle record fetched, counted exactly by the instrumented kernel, held against the HBM and the L2
peak: most of it is served by L1 and L2) and `resources`, one fraction per resource (hbm as above; l2: L1-miss requests
against the request rate a gather microbenchmark reaches on this chip; valu: vector instructions x their 2-cycle minimum
issue time against all SIMD cycles), the busiest named in `busiest_resource`.  Counter-derived inputs come from committed
rocprofv3 passes (profiles/rNN/pmc_table.json), used only while the kernel sources still hash as they did then.
`cpu_baseline` times the CPU oracle (brute force over all triangles, the reference's cpu_raycast loop) on a
bounded sample.
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import sys
import threading
import time
import types

import numpy as np
import torch  # before raytrace_amd: both must bind the same HIP runtime

import raytrace_amd as rt
from raytrace_amd import tiles as tl

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak
L2_PEAK_GBS = 34500.0       # aggregate L2 bandwidth of the eight XCDs
SIMDS = 256 * 4             # vector units on the chip
CLOCK_HZ = 2.4e9            # peak engine clock: the denominator of the VALU fraction
ROOT = os.path.dirname(os.path.abspath(__file__))

WORKLOADS = {
    "dragon1080": ("dragon.obj", 1920, 1080),
    "frames": ("dragon.obj", 1920, 1080),
    "bunny1080": ("bunny.obj", 1920, 1080),
    "dragon4k": ("dragon.obj", 3840, 2160),
    "tiles4k": ("dragon.obj", 3840, 2160),
    "ao": ("bunny_big.obj", 0, 0),
    "mirror5": ("cube_mirror_test.obj", 1920, 1080),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="", choices=[""] + sorted(WORKLOADS),
                    help="default: dragon1080 on one GPU, tiles4k on several")
    ap.add_argument("--width", type=int, default=4, help="children per node record the traversal walks: 2 or 4")
    ap.add_argument("--rays", type=int, default=64 * 1024 * 1024, help="ray count per GPU for --workload ao")
    ap.add_argument("--bvh", default="fixed", choices=["fixed", "reference", "sah"],
                    help="tree handed to rt_scene_create: the reference's centroid build as written (reference), with its "
                         "centroid indexing repaired (fixed, the default), or the surface-area build in the same node format (sah)")
    ap.add_argument("--out", default="full", choices=["full", "compact"], help="64-byte hit records or 16-byte {id,t,u,v}")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--preheat-ms", type=float, default=600.0,
                    help="before the warm-up steps: this long of unrelated GPU load (matrix products), so that the steps are timed at the "
                         "clocks the card holds under sustained load; the same steps timed on the idle card are reported beside `value` "
                         "(`idle_gpu`); 0 = none")
    ap.add_argument("--no-compare", action="store_true",
                    help="skip the extra timings beside the headline (surface-area tree, two streams, host-pointer call): profiling runs")
    ap.add_argument("--cpu-stride", type=int, default=4, help="sub-grid stride (both axes) of the CPU baseline sample")
    ap.add_argument("--tile", type=int, default=64, help="tiles4k: side of the squares dealt to the ranks (multiple of 8)")
    ap.add_argument("--assign", default="cost", choices=["cost", "roundrobin"], help="tiles4k: how squares are dealt to ranks")
    ap.add_argument("--gather", default="bgra8", choices=["bgra8", "id4", "hits8", "hits16"],
                    help="tiles4k: what crosses the GPUs into rank 0's framebuffer: every pixel's shaded 8-bit BGRA value (the full trace "
                         "step with the pixel packed in the kernel, the default), or its closest hit as a 4-byte {id}, 8-byte {id,t} or "
                         "16-byte {id,t,u,v} record")
    ap.add_argument("--pipeline", type=int, default=-1, choices=[-1, 0, 1, 2, 3, 4],
                    help="tiles4k: frames in flight, each on a stream of its own (trace, exchange and scatter of a frame on its stream, so "
                         "the exchange of a frame runs beside the traces of the next ones); 0 = one frame at a time on the caller's stream; "
                         "-1 (default) = a few pipelines are timed briefly at start-up and the best is kept")
    ap.add_argument("--side-stream", type=int, default=1, choices=[0, 1],
                    help="tiles4k with a fixed --pipeline: 0 = the scenes build their launch orders on the frame's stream (\"side_stream\" 0)")
    ap.add_argument("--scene-per-stream", type=int, default=1,
                    help="tiles4k: 1 = every frame in flight has a scene object of its own (own launch-order tables and side stream)")
    ap.add_argument("--redeal", type=int, default=0, metavar="K",
                    help="tiles4k: deal the squares again every K frames inside the timed loop, from costs every rank measures on its own "
                         "squares (0 = only the deal at start-up, made the same way; one re-deal is always timed on its own and reported)")
    ap.add_argument("--root-head-start", type=float, default=-1.0,
                    help="tiles4k: share of the whole frame's cost rank 0 counts as carrying before the squares are dealt (its gather "
                         "and scatter); negative = measured at start-up (scatter of a whole frame / one-GPU frame time)")
    ap.add_argument("--image-hint", type=int, default=1, help="declare the ray array as a W-wide image (8x8 tiles per wave)")
    ap.add_argument("--raygen", default="table", choices=["table", "device"],
                    help="table: rays are a RayIntersection array resident in HBM (the reference's interface); device: "
                         "the same primary rays are generated inside the kernel (rt_trace_primary), no ray array is read")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the N > 1 flow on one GPU)")
    ap.add_argument("--extras-deadline", type=float, default=150.0, metavar="SECONDS",
                    help="tiles4k with N > 1: the extra measurements beside the headline (re-deal, one-GPU frame and framebuffer check, "
                         "the other gathered record, frame-parallel figure) get this long; then rank 0 prints the line with what it has")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--share-of", type=int, default=0, metavar="R",
                    help="rehearsal on one GPU (tiles4k, --gpus 1, with --no-cpu-baseline): trace, scatter and time only the "
                         "squares rank 0 of R ranks would get, i.e. the per-rank part of an R-GPU step without the exchange")
    ap.add_argument("--frames", type=int, default=4,
                    help="image workloads: distinct camera positions (ray tables, all resident in HBM) the steps cycle "
                         "through, so that no step repeats the previous step's rays")
    ap.add_argument("--feedback", type=int, default=1,
                    help="1: each launch starts its work units costliest-first, by the costs an earlier launch of the same shape measured")
    ap.add_argument("--param", action="append", default=[], metavar="KEY=VALUE",
                    help="extra rt_scene_set_param settings (repeatable), applied last")
    return ap.parse_args()


def newest_profile_table(name):
    """Contents and path of the newest profiles/r*/<name>, or ({}, None)."""
    root = os.path.join(ROOT, "profiles")
    for d in sorted((x for x in os.listdir(root) if x.startswith("r")), reverse=True) if os.path.isdir(root) else []:
        path = os.path.join(root, d, name)
        if os.path.exists(path):
            with open(path) as f:
                return json.load(f), f"profiles/{d}/{name}"
    return {}, None


def kernel_source_hash():
    """sha256 over the kernel sources (as tools/collect_profiles.py stamps the counter tables it writes)."""
    import hashlib
    h = hashlib.sha256()
    for name in ("trace_kernels.hip", "trace_math.h"):
        h.update(open(os.path.join(ROOT, "raytrace_amd", "csrc", name), "rb").read())
    return h.hexdigest()[:16]


def preheat(ms):
    """Brings the GPU to the clocks it holds under sustained load before the warm-up steps: dense matrix products, nothing of
    the hot path, for `ms` milliseconds.  The power management of the card takes about 0.3 s of load to get there (a 1080p
    launch times 0.123 ms in its first 25 launches on an idle card and 0.115 ms after 0.4 s or more of any load,
    tools/per_launch.py), far longer than a few warm-up launches last, and a render loop or a serving process runs at
    the sustained state.  The line says how long it ran (`gpu_preheat_ms`) and carries the same K steps timed on the idle
    card beside `value` (`idle_gpu`)."""
    if ms <= 0:
        return
    m = torch.randn(4096, 4096, device="cuda")
    try:
        _ = m @ m
        torch.cuda.synchronize()
        work = lambda: m @ m                      # noqa: E731
    except RuntimeError:                          # no usable BLAS on this box: any load will do
        work = lambda: m.mul_(1.0000001)          # noqa: E731
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(4):
            work()
        torch.cuda.synchronize()
    del m
    torch.cuda.synchronize()


def timed(dist, steps, fn, finish=None):
    """Runs fn() `steps` times (then finish(), which completes what the steps left in flight) between barriers; returns
    (wall seconds, device ms per step by events on the current stream)."""
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
        torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        fn()
    if finish:
        finish()
    ev1.record()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
        torch.cuda.synchronize()
    return time.perf_counter() - t0, ev0.elapsed_time(ev1) / max(1, steps)


def mirror_pose(rt, bvh):
    """Camera matrix that looks at cube_mirror_test.obj from the +z side, where the cube sits above the mirror slab and the
    reflected rays of pass 1 land on it (from -z, the framing of the other workloads, every reflected ray leaves the scene)."""
    f32 = np.float32
    c, s = f32(np.cos(2.6)), f32(np.sin(2.6))
    box = bvh.root_box
    centre = 0.5 * (box[0, :3] + box[1, :3])
    dist = f32(8.0) * f32(np.abs(box[1, :3] - box[0, :3]).max())
    cam = np.array([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1]], f32)
    cam[:3, 3] = centre - dist * np.array([s, 0, c], f32)
    return cam


def run_mirror5(a, local):
    """--workload mirror5: BASELINE config 5.  One step = for cube_mirror_test.obj AND teapot.obj a 1920x1080 primary pass plus
    four reflection passes of the full `trace` kernel (rt_trace_pass; records ping-ponged on the device, as
    render_client/img_main.cpp:447-498 does for its one bounce).  `value` counts the rays actually traced: a record whose
    previous pass missed is passed through, not traced (the record passes are reported beside it).  Every pass of both scenes is
    spot-checked against the CPU oracle (ray regeneration + brute-force closest hit) on a sub-grid, which is also the timed
    CPU baseline."""
    W, H = 1920, 1080
    n = W * H
    stream = torch.cuda.current_stream().cuda_stream
    jobs = []
    for model in ("cube_mirror_test.obj", "teapot.obj"):
        mesh = rt.Mesh.load_obj(os.path.join(rt.MODELS_DIR, model), "reference")
        bvh = rt.BVH(mesh, a.bvh)
        scene = rt.Scene(mesh, bvh, device=local, width=4)
        scene.set_param("width", 4)
        scene.set_param("image_width", W if a.image_hint else 0)
        scene.set_param("feedback", a.feedback)
        for kv in a.param:
            k, v = kv.split("=")
            scene.set_param(k, int(v))
        if model.startswith("cube_mirror"):
            cam, rays_h = mirror_pose(rt, bvh), rt.primary_rays(W, H, (0, 0, 0))
        else:
            cam, rays_h = np.eye(4, dtype=np.float32), rt.primary_rays(W, H, rt.frame_camera(bvh.root_box, W, H))
        jobs.append(types.SimpleNamespace(model=model, mesh=mesh, bvh=bvh, scene=scene, cam=cam, rays_h=rays_h,
                                          rays_d=torch.from_numpy(rays_h).cuda(), info=scene.info(),
                                          ping=torch.empty((n, 16), dtype=torch.float32, device="cuda"),
                                          pong=torch.empty((n, 16), dtype=torch.float32, device="cuda"),
                                          rgba=torch.empty((n, 4), dtype=torch.float32, device="cuda")))

    def chain(j, after_pass=None):
        j.scene.trace_pass(j.rays_d.data_ptr(), j.ping.data_ptr(), j.rgba.data_ptr(), n, j.cam, 0, stream)
        src, dst = j.ping, j.pong
        if after_pass:
            after_pass(0, j.rays_d, src)
        for k in range(1, 5):
            j.scene.trace_pass(src.data_ptr(), dst.data_ptr(), j.rgba.data_ptr(), n, j.cam, 1, stream)
            if after_pass:
                after_pass(k, src, dst)
            src, dst = dst, src

    def step():
        for j in jobs:
            chain(j)

    preheat(a.preheat_ms)
    for _ in range(a.warmup):
        step()
    wall, dev_ms = timed(None, a.steps, step)
    ms_per_step = wall * 1e3 / max(1, a.steps)

    # ---- one checked step: rays traced per pass, records fetched per pass, and the oracle on a sub-grid ----
    import oracle
    sgrid = max(1, a.cpu_stride)
    sel = (np.arange(0, H, sgrid)[:, None] * W + np.arange(0, W, sgrid)[None, :]).reshape(-1)
    sel_d = torch.from_numpy(sel).cuda()
    threads = oracle.default_threads()
    per_scene, traced_total, cpu_secs, cpu_rays, id_bad, t_bad = [], 0, 0.0, 0, 0, 0
    nodes_total = tris_total = 0
    for j in jobs:
        j.scene.set_param("feedback", 0)     # the counting launches below follow no order
        passes = []

        def check(k, src, dst, j=j, passes=passes):
            nonlocal traced_total, cpu_secs, cpu_rays, id_bad, t_bad, nodes_total, tris_total
            torch.cuda.synchronize()
            live = torch.ones(n, dtype=torch.bool, device="cuda") if k == 0 else (src[:, 14].view(torch.int32) != -1)
            traced = int(live.sum())
            hits = int((dst[:, 14].view(torch.int32) != -1).sum())
            # records fetched: the rays of this pass rebuilt from its input records (tracekernel.ck:227-265) and counted like any others
            if k == 0:
                o4 = torch.cat([src[:, 0:3], torch.ones((n, 1), device="cuda")], dim=1)
                d4 = torch.cat([src[:, 4:7], torch.zeros((n, 1), device="cuda")], dim=1)
                m = torch.from_numpy(j.cam).cuda()
                tab = torch.zeros((n, 16), dtype=torch.float32, device="cuda")
                tab[:, 0:3] = (o4 @ m.T)[:, 0:3]
                tab[:, 4:7] = (d4 @ m.T)[:, 0:3]
            else:
                d4, n4 = src[:, 4:8], src[:, 8:12]
                s2 = 2.0 * (d4 * n4).sum(dim=1)
                tab = torch.zeros((n, 16), dtype=torch.float32, device="cuda")
                tab[:, 0:3] = src[:, 0:3] + n4[:, 0:3] * 0.00001
                tab[:, 4:7] = d4[:, 0:3] - s2[:, None] * n4[:, 0:3]
                tab = tab[live].contiguous()
            tab[:, 3] = 1.0
            c = j.scene.count_fetches_device(tab.data_ptr(), len(tab)) if len(tab) else {"nodes": 0, "tris": 0}
            nodes_total += c["nodes"]
            tris_total += c["tris"]
            # the oracle on the sub-grid: regenerate the rays from the GPU's own input records, brute force over all triangles
            prev_s = src[sel_d].cpu().numpy()
            got = dst[sel_d].cpu().numpy()
            live_s = np.ones(len(sel), bool) if k == 0 else (prev_s[:, 14].view(np.uint32) != rt.MISS_ID)
            c0 = time.perf_counter()
            want_rays = oracle.trace_setup(prev_s[live_s], j.cam, 0 if k == 0 else 1)
            oid, ot, _, _ = oracle.closest_hit(j.mesh.verts, j.mesh.faces, want_rays, threads=threads) if live_s.any() else (np.zeros(0, np.uint32),) * 4
            cpu_secs += time.perf_counter() - c0
            cpu_rays += int(live_s.sum())
            gid, gt = got[live_s, 14].view(np.uint32), got[live_s, 15]
            hit = oid != rt.MISS_ID
            id_bad += int((gid != oid).sum()) + int((got[~live_s, 14].view(np.uint32) != rt.MISS_ID).sum())
            t_bad += int((gt[hit].view(np.uint32) != ot[hit].view(np.uint32)).sum())
            traced_total += traced
            passes.append({"pass": k, "rays_traced": traced, "rays_hit": hits, "node_records": c["nodes"], "tri_records": c["tris"]})

        chain(j, check)
        per_scene.append({"model": j.model, "triangles": len(j.mesh.faces), "nodes": j.info["nodes4"], "passes": passes})

    record_passes = len(jobs) * 5 * n
    secs = ms_per_step * 1e-3
    node_b, tri_b = jobs[0].info["node4_bytes"], jobs[0].info["tri_bytes"]
    stream_b = 64 + 64 + 16       # per record pass: record in, record out, RGBA32F pixel
    hbm = {"achieved": round(stream_b * record_passes / secs / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "what": "compulsory streams of every record pass (terminated records are read and passed through too): 64 B in + 64 B out + 16 B pixel"}
    hbm["frac"] = round(hbm["achieved"] / HBM_PEAK_GBS, 4)
    roofline = {"bound": "hbm", "achieved": hbm["achieved"], "peak": hbm["peak"], "unit": "GB/s", "frac": hbm["frac"], "traffic": None,
                "resources": {"hbm": hbm}, "kernel": "ClosestHitKernel<4, true, kModeTrace>", "kernel_ms": round(dev_ms / (len(jobs) * 5), 4),
                "algorithmic": {"stream_bytes_per_record_pass": stream_b,
                                "bytes_per_traced_ray": round((stream_b * record_passes + nodes_total * node_b + tris_total * tri_b) / max(1, traced_total), 1),
                                "node_records_per_traced_ray": round(nodes_total / max(1, traced_total), 3),
                                "tri_records_per_traced_ray": round(tris_total / max(1, traced_total), 3),
                                "node_bytes": node_b, "tri_bytes": tri_b}}
    cpu = {"value": round(cpu_rays / max(cpu_secs, 1e-9) / 1e6, 6), "unit": "Mrays/s", "cores": threads, "kind": "port",
           "sample": f"stride-{sgrid} sub-grid of every pass of both scenes ({cpu_rays} traced rays): ray regeneration from the GPU's own input "
                     "records + brute-force closest hit over all triangles",
           "seconds": round(cpu_secs, 2), "id_mismatches_vs_gpu": id_bad, "t_bit_mismatches_vs_gpu": t_bad}
    line = {
        "metric": "closest-hit Mrays/s", "value": round(traced_total / secs / 1e6, 1), "unit": "Mrays/s", "n_gpus": 1,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"cube_mirror_test.obj + teapot.obj {W}x{H}: primary pass + 4 reflection passes each (full trace kernel)",
                   "name": "mirror5", "rays_traced_per_step": traced_total, "record_passes_per_step": record_passes,
                   "record_passes_Mrecords_per_s": round(record_passes / secs / 1e6, 1), "loader": "reference",
                   "bvh": a.bvh, "bvh_width": 4, "hit_record": "64B RayIntersection + 16B RGBA32F pixel",
                   "camera": "cube_mirror_test: reference pinhole from the +z side (rotation 2.6 rad about y); teapot: framed from -z",
                   "scenes": per_scene},
        "roofline": roofline, "cpu_baseline": cpu, "gpu_preheat_ms": a.preheat_ms,
    }
    print(json.dumps(line), flush=True)
    for j in jobs:
        j.scene.close()


def run_tiles4k(a, rank, world, local, dist, red_dev):
    """--workload tiles4k (the default with N > 1): ONE dragon 3840x2160 frame per step, strong scaling.  The image is cut into
    squares and dealt to the ranks by measured traversal cost (rt_shard_plan); every rank traces its squares; the records
    cross into rank 0's framebuffer in the step (rt_gather_scatter: grouped RCCL point-to-point into rank 0, then the device
    scatter).  Several frames are in flight, each on a stream of its own, so the exchange of a frame runs beside the traces of
    the next ones.  Everything between the rays and the framebuffer goes through the library's C ABI."""
    model, W, H = WORKLOADS["tiles4k"]
    assert a.tile % 8 == 0
    mesh = rt.Mesh.load_obj(os.path.join(rt.MODELS_DIR, model), "tolerant")
    # The model ships no materials: every face names the all-zero sky material, and every pixel of the full trace step would be
    # black.  One diffuse grey instead (same arithmetic per pixel), so that a pixel depends on where its ray hits and the
    # comparison of the gathered framebuffer with the one-GPU frame can tell a misplaced square.
    mats = mesh.materials.copy()
    mats[:, 0:3] = 0.8
    mats[:, 9] = 1.0
    mesh = rt.Mesh.from_arrays(mesh.verts, mesh.faces, mesh.normals, mats)
    bvh = rt.BVH(mesh, a.bvh)
    def new_scene():
        sc = rt.Scene(mesh, bvh, device=local, width=4)
        sc.set_param("width", 4)
        sc.set_param("feedback", a.feedback)
        for kv in a.param:
            k, v = kv.split("=")
            sc.set_param(k, int(v))
        return sc
    scene = new_scene()
    scene_sets, stream_sets = {}, {}      # per (frames in flight, side stream): the scene objects and streams of that pipeline, made once
    info = scene.info()
    box = bvh.root_box
    origin = rt.frame_camera(box, W, H)
    null_stream = torch.cuda.current_stream().cuda_stream
    extra = {}
    R = a.share_of if (a.share_of > 1 and world == 1) else world
    me = rank if R == world else 0
    stand_for = [rank] if R == world else list(range(R))     # the ranks whose measurements this process makes
    if R != world:
        extra["rehearsal"] = f"rank 0's share of {a.share_of} ranks on one GPU, no exchange; value counts only those rays"
    full_h = rt.primary_rays(W, H, origin)
    full_d = torch.from_numpy(full_h).cuda()                 # the frame's ray table; a rank's rays are gathered from it on the device
    hint = a.tile if (a.image_hint and W % a.tile == 0) else 0
    from raytrace_amd import tiles as tl
    cam = np.eye(4, dtype=np.float32)
    comm = None
    if world > 1 and a.backend == "nccl":
        idt = torch.zeros(128, dtype=torch.uint8, device=red_dev)
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(rt.Comm.unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, 0)
        ok = torch.ones(1, dtype=torch.int32, device=red_dev)
        try:
            comm = rt.Comm(bytes(idt.cpu().numpy().tobytes()), world, rank, local)
        except rt.RtError as e:      # reported, and the job goes on through torch.distributed's own RCCL communicator
            extra["rt_comm_create_failed"] = str(e)
            ok.zero_()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0 and comm is not None:
            comm.close()
            comm = None
        if comm is not None:
            # every RCCL call the exchange uses, once, with the result looked at on the host (rt_comm_check; a collective,
            # so it runs only once every rank holds a communicator)
            try:
                comm.check()
            except rt.RtError as e:
                extra["rt_comm_check_failed"] = str(e)
                ok.zero_()
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                comm.close()
                comm = None

    def sum_over_ranks(values):
        """In place, float64 host array: the sum over the job's ranks (RCCL through the library; gloo in the rehearsal)."""
        if comm is not None:
            comm.sum_f64(values)
        elif dist:
            t = torch.from_numpy(values).to(red_dev)
            dist.all_reduce(t)
            values[...] = t.cpu().numpy()
        return values

    def share_rays(plan, r):
        idx = torch.from_numpy(plan.ray_indices(r).astype(np.int64)).cuda()
        return full_d.index_select(0, idx)

    def measured_square_costs(plan):
        """Per-square traversal cost of the current frame, measured where the squares are traced: every rank runs the
        instrumented launch over its OWN rays (its squares stacked, the image hint it traces them with), adds the work-unit
        costs up per square (rt_shard_square_costs), and the arrays are summed over the ranks."""
        sq = np.zeros(len(plan.tile_order), np.float64)
        scene.set_param("image_width", hint)
        for r in stand_for:
            rays_r = share_rays(plan, r)
            plan.square_costs_of_rank(r, scene.unit_costs(rays_r.data_ptr(), len(rays_r)), hint, sq)
            del rays_r
        return sum_over_ranks(sq)

    REC = {"hits16": 16, "hits8": 8, "id4": 4, "bgra8": 4}
    rec_default = REC[a.gather]

    def one_gpu_call(sc, rays_ptr, n, out_ptr, payload, q):
        if payload == "bgra8":   # the full trace step (identity camera), the pixel written as 8-bit BGRA by the kernel itself
            sc.trace_pass_bgra8(rays_ptr, None, out_ptr, n, cam, 0, q)
        else:                    # closest hits, only the record that crosses the GPUs
            sc.intersect_compact(rays_ptr, n, out_ptr, REC[payload], q)

    # What the gathering rank does besides tracing (the scatter of the whole frame), as a share of the time one GPU needs for
    # the frame: its head start in the deal.  Measured on rank 0, handed to the others by a sum.
    head = np.zeros(R, np.float64)
    one_ms = None
    if R > 1 and a.root_head_start < 0:
        hs = np.zeros(1, np.float64)
        if rank == 0:
            rr = rt.ShardPlan(W, H, a.tile, R)
            ids_h, first_h = rr.scatter_tables()
            ids_t, first_t = torch.from_numpy(ids_h.astype(np.int32)).cuda(), torch.from_numpy(first_h.astype(np.int32)).cuda()
            g = torch.zeros((W * H, rec_default // 4), dtype=torch.int32, device="cuda")
            fb = torch.zeros_like(g)
            whole = np.array([0, W * H], np.uint64)

            def scatter_whole():
                rt.gather_scatter(None, 0, 1, 0, local, g.data_ptr(), g.data_ptr(), whole, rec_default, ids_t.data_ptr(), first_t.data_ptr(),
                                  len(ids_h), a.tile, W, H, fb.data_ptr(), null_stream)
            probe = new_scene()      # a scene of its own: the job's scenes and their streams start untouched
            probe.set_param("image_width", W)

            def trace_whole():
                one_gpu_call(probe, full_d.data_ptr(), W * H, g.data_ptr(), a.gather, null_stream)
            t = []
            for fn, warm in ((scatter_whole, 3), (trace_whole, 12)):
                for _ in range(warm):
                    fn()
                torch.cuda.synchronize()
                c0 = time.perf_counter()
                for _ in range(20):
                    fn()
                torch.cuda.synchronize()
                t.append((time.perf_counter() - c0) / 20 * 1e3)
            one_ms = t[1]
            hs[0] = t[0] / t[1]
            extra["root_scatter_whole_frame_ms"] = round(t[0], 4)
            probe.close()
            del g, fb
        head[0] = float(sum_over_ranks(hs)[0]) if R == world else float(hs[0])
    elif R > 1:
        head[0] = a.root_head_start
    extra["root_head_start"] = round(float(head[0]), 4)

    # The deal: start round-robin, measure, deal by cost.  The same two calls re-deal the squares while the job runs.
    plan = rt.ShardPlan(W, H, a.tile, R)
    sq_costs = measured_square_costs(plan)
    if a.assign == "cost":
        plan = rt.ShardPlan(W, H, a.tile, R, None, sq_costs, head)
    extra["tile_assignment"] = {"squares": len(plan.tile_order), "side": a.tile, "by": a.assign, "planned_by": "rt_shard_square_costs + rt_shard_plan",
                                "modelled_max_over_mean_cost": round(tl.balance([t.astype(np.int64) for t in plan.tiles[1:]] if R > 1 else
                                                                                [t.astype(np.int64) for t in plan.tiles], sq_costs), 4),
                                "root_share_of_cost": round(float(sq_costs[plan.tiles[0]].sum() / max(1.0, sq_costs.sum())), 4),
                                "round_robin_would_be": round(tl.balance(tl.assign_tiles(len(plan.tile_order), R, None), sq_costs), 4)}
    rays_d = share_rays(plan, me)
    n_local = len(rays_d)
    scene.set_param("image_width", hint)     # a rank's rays are its squares stacked: an image one square wide, so waves keep 8x8 blocks

    WHAT = {"hits16": "16 B {id,t,u,v}", "hits8": "8 B {id,t}", "id4": "4 B {id}", "bgra8": "8-bit BGRA pixel (full trace step, rt_trace_pass_bgra8)"}

    def build(payload, frames_in_flight, side=1):
        """Buffers and the step of the sharded frame for one kind of gathered record and one pipeline: `frames_in_flight` frames,
        each on a stream of its own with a scene object of its own; side = 0: the scenes build their launch orders on the frame's
        stream instead of a side stream each.  Streams and scenes of a pipeline are made once and kept: a stream keeps the
        hardware queue it was given, so a pipeline that was timed is the pipeline that runs."""
        rec_b = REC[payload]
        NB = max(1, frames_in_flight)
        key = (frames_in_flight, side)
        if frames_in_flight >= 1 and key not in stream_sets:
            stream_sets[key] = [torch.cuda.Stream() for _ in range(NB)]
        streams = stream_sets[key] if frames_in_flight >= 1 else None

        def raw(b):
            return streams[b].cuda_stream if streams else null_stream

        def records(nrec):
            return torch.empty((nrec, rec_b // 4), dtype=torch.int32, device="cuda")

        scn = [scene] * NB
        if frames_in_flight >= 1 and (a.scene_per_stream or side == 0):
            if key not in scene_sets:
                scene_sets[key] = [new_scene() for _ in range(NB if a.scene_per_stream else 1)]
                for sc in scene_sets[key]:
                    sc.set_param("image_width", hint)
                    sc.set_param("side_stream", side)
                    # with several frames in flight the next frame's launch fills a launch's tail; splitting the costliest units of
                    # the full trace step then only adds waves (measured -5 %); one frame at a time keeps the library's default
                    sc.set_param("trace_split", 0 if frames_in_flight >= 2 else -1)
            scn = [scene_sets[key][b % len(scene_sets[key])] for b in range(NB)]
        root = me == 0
        if R == world:
            first_record = plan.first_record
            ids_h, first_h = plan.scatter_tables()
        else:     # rehearsal: only rank 0's squares exist
            first_record = np.array([0, n_local], np.uint64)
            ids_h = np.ascontiguousarray(plan.tiles[0])
            first_h = np.zeros(len(ids_h), np.uint32)
            rt._check(rt.lib.rt_shard_ray_indices(W, H, a.tile, rt._ptr(ids_h), len(ids_h), 0, None, rt._ptr(first_h)))
        total = int(first_record[-1])
        gather = [records(total) for _ in range(NB)] if root else [None] * NB
        fbs = [records(W * H) for _ in range(NB)] if root else [None] * NB
        ids_d = torch.from_numpy(ids_h.astype(np.int32)).cuda() if root else None
        first_d = torch.from_numpy(first_h.astype(np.int32)).cuda() if root else None
        lo = int(first_record[me])
        mine = [g[lo:lo + n_local] for g in gather] if root else [records(n_local) for _ in range(NB)]   # root traces into its slice
        host_in = [torch.empty((int(first_record[r + 1] - first_record[r]), rec_b // 4), dtype=torch.int32) for r in range(world)] \
            if (world > 1 and comm is None and root and a.backend == "gloo") else None
        state = types.SimpleNamespace(frame=0, last=None, fresh=True)

        def trace_into(b):
            one_gpu_call(scn[b], rays_d.data_ptr(), n_local, mine[b].data_ptr(), payload, raw(b))

        def exchange(b):
            q = raw(b)
            if world > 1 and comm is None:
                # without the library's communicator: through torch.distributed (gloo rehearsal: host memory, synchronously; nccl:
                # its own RCCL communicator, grouped point-to-point); the scatter is still the library's
                ctx = torch.cuda.stream(streams[b]) if streams else contextlib.nullcontext()
                with ctx:
                    if a.backend == "nccl":
                        ops = [dist.P2POp(dist.irecv, gather[b][int(first_record[r]):int(first_record[r + 1])], r) for r in range(1, world)] \
                            if root else [dist.P2POp(dist.isend, mine[b], 0)]
                        for w in dist.batch_isend_irecv(ops):
                            w.wait()       # makes the current stream wait, not the host
                    elif root:
                        for r in range(1, world):
                            dist.recv(host_in[r], src=r)
                            gather[b][int(first_record[r]):int(first_record[r + 1])].copy_(host_in[r])
                    else:
                        dist.send(mine[b].cpu(), dst=0)
                if root:
                    rt.gather_scatter(None, 0, 1, 0, local, gather[b].data_ptr(), gather[b].data_ptr(), np.array([0, total], np.uint64), rec_b,
                                      ids_d.data_ptr(), first_d.data_ptr(), len(ids_h), a.tile, W, H, fbs[b].data_ptr(), q)
            else:
                nr = world if R == world else 1
                rt.gather_scatter(comm, me if nr > 1 else 0, nr, 0, local, mine[b].data_ptr(), gather[b].data_ptr() if root else None, first_record, rec_b,
                                  ids_d.data_ptr() if root else None, first_d.data_ptr() if root else None, len(ids_h) if root else 0,
                                  a.tile, W, H, fbs[b].data_ptr() if root else None, q)
            if root:
                state.last = fbs[b]

        def step():
            b = state.frame % NB
            state.frame += 1
            trace_into(b)
            exchange(b)

        def flush():
            if streams:
                cur = torch.cuda.current_stream()
                for st in streams:
                    cur.wait_stream(st)

        def start():      # work enqueued on the current stream before the loop (warm-up, fills) comes first
            if streams:
                cur = torch.cuda.current_stream()
                for st in streams:
                    st.wait_stream(cur)
        return types.SimpleNamespace(payload=payload, rec_b=rec_b, NB=NB, nstreams=len(streams) if streams else 1, step=step, flush=flush, start=start,
                                     frames_in_flight=frames_in_flight, side=side,
                                     trace_into=trace_into, exchange=exchange, fbs=fbs, state=state, first_record=first_record)

    def run(sh, steps, warm):
        for _ in range(warm):
            sh.step()
        sh.flush()
        torch.cuda.synchronize()
        sh.start()
        w, _ = timed(dist, steps, sh.step, sh.flush)
        t = torch.tensor([w], dtype=torch.float64, device=red_dev)
        if dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) * 1e3 / max(1, steps)

    def redeal():
        """The squares dealt again from the costs of the frame as it is traced now (measured on every rank's own rays, summed
        over the ranks): what follows a moving camera.  Everything in flight is finished first; buffers are rebuilt."""
        nonlocal plan, rays_d, n_local, sh
        sh.flush()
        torch.cuda.synchronize()
        before = np.empty(len(plan.tile_order), np.int32)
        for r in range(R):
            before[plan.tiles[r]] = r
        sq = measured_square_costs(plan)
        plan = rt.ShardPlan(W, H, a.tile, R, None, sq, head) if a.assign == "cost" else plan
        after = np.empty(len(plan.tile_order), np.int32)
        for r in range(R):
            after[plan.tiles[r]] = r
        rays_d = share_rays(plan, me)
        n_local = len(rays_d)
        sh = build(a.gather, sh.frames_in_flight, sh.side)
        sh.start()
        return int((before != after).sum())

    def stepper():
        if a.redeal > 0 and sh.state.frame > 0 and sh.state.frame % a.redeal == 0 and not sh.state.fresh:
            redeal()
        sh.state.fresh = False
        sh.step()

    preheat(a.preheat_ms)      # the pipelines below are chosen, and the step timed, at the card's sustained clocks
    extra["gpu_preheat_ms"] = a.preheat_ms
    if a.pipeline >= 0:
        sh = build(a.gather, a.pipeline, a.side_stream)
    else:
        # Which pipeline: streams share the GPU's few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) in the order they were
        # first used, and a stream whose head waits -- a side stream for a launch's end, a frame stream for its records to arrive
        # -- holds up every stream mapped to the same queue.  How that falls out depends on the process (its other streams, the
        # communication library's), so a few pipelines are timed for a moment on every rank and the one with the best slowest
        # rank is kept, streams, scenes and all.
        cands = [(4, 1), (3, 1), (3, 0), (2, 0), (4, 0), (1, 0)]
        tried = {}
        for c in cands:
            cand = build(a.gather, c[0], c[1])
            tried[c] = (run(cand, 40, 24), cand)
        best = min(cands, key=lambda c: tried[c][0])
        sh = tried[best][1]
        extra["pipelines_tried"] = [{"frames_in_flight": c[0], "side_stream": c[1], "ms_per_step": round(tried[c][0], 4)} for c in cands]
        del tried, cand
    if a.redeal > 0:
        for _ in range(a.warmup):
            sh.step()
        sh.flush()
        torch.cuda.synchronize()
        sh.start()
        w_all, _ = timed(dist, a.steps, stepper, lambda: sh.flush())
        t_all = torch.tensor([w_all], dtype=torch.float64, device=red_dev)
        if dist:
            dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
        ms_per_step = float(t_all.item()) * 1e3 / max(1, a.steps)
        extra["squares_redealt_every_frames"] = a.redeal
    else:
        ms_per_step = run(sh, a.steps, a.warmup)
    n_total = n_local
    if dist:
        nn = torch.tensor([n_local], dtype=torch.float64, device=red_dev)
        dist.all_reduce(nn, op=dist.ReduceOp.SUM)
        n_total = int(nn.item())
    mrays = n_total / (ms_per_step * 1e-3) / 1e6

    # the parts of the step on their own: trace only (slowest rank), exchange only
    sub = max(5, a.steps // 4)
    solo = build(a.gather, 0)
    w_tr, _ = timed(dist, sub, lambda: solo.trace_into(0))
    w_ex, _ = timed(dist, sub, lambda: solo.exchange(0))
    parts = torch.tensor([w_tr / sub * 1e3, w_ex / sub * 1e3], dtype=torch.float64, device=red_dev)
    if dist:
        dist.all_reduce(parts, op=dist.ReduceOp.MAX)
    trace_ms = float(parts[0])
    extra["trace_only_ms"] = round(trace_ms, 4)
    extra["gather_and_scatter_ms"] = round(float(parts[1]), 4)
    extra["gather_bytes_into_rank0"] = int((int(sh.first_record[-1]) - int(plan.first_record[1] - plan.first_record[0])) * sh.rec_b) if R == world else 0
    extra["gathered_record"] = WHAT[a.gather]
    extra["frames_in_flight"] = sh.NB
    extra["streams"] = sh.nstreams
    extra["launch_orders_built_on"] = "a side stream per scene" if sh.side else "the frame's own stream"
    extra["exchange"] = ("rt_gather_scatter over an RCCL communicator made by rt_comm_create" if comm is not None else
                         ("rt_gather_scatter, one rank" if world == 1 else
                          ("gloo through host memory (rehearsal)" if a.backend == "gloo" else "torch.distributed batch_isend_irecv (the library's communicator "
                           "could not be made)") + " + rt_gather_scatter for the scatter"))
    # ---- everything the JSON line needs is measured; what follows are extras beside it.  On several GPUs they run under a
    # deadline: should one of them stall (they exercise paths no one-GPU run can), rank 0 still prints the line with the extras
    # it has and every rank leaves.
    base_line = {}
    emitted = threading.Lock()

    def emit(cut_short=None):
        if not emitted.acquire(blocking=False):
            return
        if rank == 0:
            out = dict(base_line)
            out.update(extra)
            if cut_short is not None:
                out["extras_cut_short_after_s"] = cut_short
            print(json.dumps(out), flush=True)

    def expired():
        emit(a.extras_deadline)
        os._exit(0)
    if rank == 0:
        n_count = min(n_local, 1 << 24)
        cnt = scene.count_fetches_device(rays_d.data_ptr(), n_count)
        n_node, n_tri = cnt["nodes"] / cnt["rays"], cnt["tris"] / cnt["rays"]
        out_b = 16
        stream_b = 64 + out_b
        trav_b = n_node * info["node4_bytes"] + n_tri * info["tri_bytes"]
        secs = trace_ms * 1e-3          # the roofline figures are the trace kernel's, not the exchange's
        hbm = {"achieved": round(stream_b * n_local / secs / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "what": f"compulsory streams of this rank's trace launch: 64 B ray in + {out_b} B out"}
        hbm["frac"] = round(hbm["achieved"] / HBM_PEAK_GBS, 4)
        roofline = {"bound": "hbm", "achieved": hbm["achieved"], "peak": hbm["peak"], "unit": hbm["unit"], "frac": hbm["frac"], "traffic": None,
                    "resources": {"hbm": hbm}, "kernel": "ClosestHitKernel", "kernel_ms": round(trace_ms, 4),
                    "algorithmic": {"bytes_per_ray": round(stream_b + trav_b, 1), "stream_bytes_per_ray": stream_b,
                                    "node_records_per_ray": round(n_node, 3), "tri_records_per_ray": round(n_tri, 3),
                                    "node_bytes": info["node4_bytes"], "tri_bytes": info["tri_bytes"],
                                    "hit_fraction": round(cnt["rays_hit"] / cnt["rays"], 4)}}
        cpu = cpu_bvh = None
        if world == 1 and R == 1 and not a.no_cpu_baseline and a.gather in ("hits8", "hits16"):
            import oracle  # the CPU checker, used here only as the timed baseline
            for fb in sh.fbs:
                fb.fill_(-1)            # wiped first: the check below must see THIS step's gather and scatter
            torch.cuda.synchronize()
            sh.start()
            sh.step()
            sh.flush()
            torch.cuda.synchronize()
            fb = sh.state.last.cpu().numpy()
            threads = oracle.default_threads()
            sgr = a.cpu_stride
            sel = (np.arange(0, H, sgr)[:, None] * W + np.arange(0, W, sgr)[None, :]).reshape(-1)
            c0 = time.perf_counter()
            oid, ot, _, _ = oracle.closest_hit(mesh.verts, mesh.faces, full_h[sel], threads=threads)
            cdt = time.perf_counter() - c0
            hit = oid != rt.MISS_ID
            gid, gt = fb[sel, 0].view(np.uint32), fb[sel, 1].view(np.float32)
            cpu = {"value": round(len(sel) / cdt / 1e6, 6), "unit": "Mrays/s", "cores": threads, "kind": "port",
                   "sample": f"stride-{sgr} sub-grid of the {W}x{H} ray set ({len(sel)} rays x {len(mesh.faces)} triangles, brute force)",
                   "seconds": round(cdt, 2), "id_mismatches_vs_gpu": int((gid != oid).sum()),
                   "t_bit_mismatches_vs_gpu": int((gt[hit].view(np.uint32) != ot[hit].view(np.uint32)).sum())}
            c0 = time.perf_counter()
            oid, ot, _, _ = oracle.closest_hit_bvh(mesh.verts, mesh.faces, bvh.nodes, full_h, threads=threads)
            cdt = time.perf_counter() - c0
            hit = oid != rt.MISS_ID
            gid, gt = fb[:, 0].view(np.uint32), fb[:, 1].view(np.float32)
            cpu_bvh = {"value": round(len(full_h) / cdt / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
                       "sample": f"all {len(full_h)} rays, closest hit through the same CentroidBVH tree (oracle tree walk), against the gathered framebuffer",
                       "seconds": round(cdt, 2), "id_mismatches_vs_gpu": int((gid != oid).sum()),
                       "t_bit_mismatches_vs_gpu": int((gt[hit].view(np.uint32) != ot[hit].view(np.uint32)).sum())}
        line = {
            "metric": "closest-hit Mrays/s", "value": round(mrays, 1), "unit": "Mrays/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{model} {W}x{H} primary rays, image squares sharded over {world} GPU(s), framebuffer gathered to rank 0",
                       "name": "tiles4k", "rays_per_gpu": n_local, "rays_total": n_total, "triangles": len(mesh.faces), "loader": "tolerant",
                       "bvh": a.bvh, "bvh_width": 4, "nodes": info["nodes4"], "hit_record": WHAT[a.gather],
                       "rays": "RayIntersection table resident in HBM",
                       "parallelism": f"{world} x one process per GPU, cost-balanced image squares (rt_shard_plan) + RCCL gather into rank 0 and "
                                      f"device scatter (rt_gather_scatter), {sh.NB} frame(s) in flight"},
            "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_bvh": cpu_bvh,
        }
        base_line.update(line)
    deadline = None
    if world > 1 and a.extras_deadline > 0:
        if dist:
            dist.barrier()
        deadline = threading.Timer(a.extras_deadline, expired)
        deadline.daemon = True
        deadline.start()
    # what the host spends enqueueing a step (nothing waited for): the floor of a step however fast the GPU is
    torch.cuda.synchronize()
    c0 = time.perf_counter()
    for _ in range(50):
        sh.step()
    extra["host_enqueue_us_per_step"] = round((time.perf_counter() - c0) / 50 * 1e6, 1)
    sh.flush()
    torch.cuda.synchronize()
    if sh.frames_in_flight >= 1:
        ms_1 = run(solo, sub, 2)
        extra["same_steps_one_frame_at_a_time"] = {"ms_per_step": round(ms_1, 4), "value": round(n_total / (ms_1 * 1e-3) / 1e6, 1), "unit": "Mrays/s"}
    del solo
    # one re-deal, untimed: what it costs, and (same camera) that it deals the same squares to the same ranks
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    c0 = time.perf_counter()
    moved = redeal()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    extra["redeal"] = {"ms": round((time.perf_counter() - c0) * 1e3, 2), "squares_that_changed_rank": moved,
                       "what": "every rank measures its own squares (rt_scene_unit_costs + rt_shard_square_costs), the costs are summed over the "
                               "ranks (rt_comm_sum_f64), rt_shard_plan deals again, ray tables and buffers are rebuilt"}

    if world > 1:
        # the whole frame on ONE GPU of this job (rank 0 alone, the others wait): the strong-scaling reference point
        if rank == 0:
            one_out = torch.empty((W * H, sh.rec_b // 4), dtype=torch.int32, device="cuda")
            scene.set_param("image_width", W)

            def one_gpu_frame():
                one_gpu_call(scene, full_d.data_ptr(), W * H, one_out.data_ptr(), a.gather, null_stream)
            for _ in range(10):
                one_gpu_frame()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(sub):
                one_gpu_frame()
            torch.cuda.synchronize()
            one_ms = (time.perf_counter() - t0) / sub * 1e3
            extra["one_gpu_same_frame"] = {"ms_per_step": round(one_ms, 4), "value": round(W * H / (one_ms * 1e-3) / 1e6, 1),
                                           "unit": "Mrays/s", "speedup_of_this_job": round(one_ms / ms_per_step, 3)}
            scene.set_param("image_width", hint)
            for fb in sh.fbs:
                fb.fill_(-1)             # every frame of the loop is the same image: only a wiped framebuffer proves this step's scatter
            torch.cuda.synchronize()
        sh.start()
        sh.step()
        sh.flush()
        torch.cuda.synchronize()
        if rank == 0:
            extra["framebuffer_equals_one_gpu"] = bool(torch.equal(sh.state.last, one_out))
            extra["framebuffer_distinct_records"] = int(torch.unique(one_out, dim=0).shape[0])   # the comparison is not of one colour
            del one_out
        dist.barrier()
        # the same sharded frame with the 16-byte hit as the gathered record (twice the bytes on the links)
        if a.gather != "hits16":
            alt = build("hits16", sh.frames_in_flight, sh.side)
            ms_a = run(alt, sub, max(2, a.warmup // 4))
            extra["same_frame_gathered_as"] = {"record": WHAT["hits16"], "ms_per_step": round(ms_a, 4), "value": round(W * H / (ms_a * 1e-3) / 1e6, 1), "unit": "Mrays/s"}
            del alt
        # weak scaling beside it: every rank traces its own 1920x1080 dragon frame (64-byte records), no exchange
        Wf, Hf = 1920, 1080
        of = rt.frame_camera(box, Wf, Hf)
        of[0] += rank * 0.01 * float(box[1, 0] - box[0, 0])
        rays_f = torch.from_numpy(rt.primary_rays(Wf, Hf, of)).cuda()
        hits_f = torch.empty((Wf * Hf, 16), dtype=torch.float32, device="cuda")
        scene.set_param("image_width", Wf)

        def step_f():
            scene.intersect_async(rays_f.data_ptr(), Wf * Hf, hits_f.data_ptr(), None, null_stream)
        for _ in range(a.warmup):
            step_f()
        w_f, _ = timed(dist, a.steps, step_f)
        tf = torch.tensor([w_f], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tf, op=dist.ReduceOp.MAX)
        ms_f = float(tf.item()) * 1e3 / max(1, a.steps)
        extra["frame_parallel_weak_scaling"] = {"workload": "dragon.obj 1920x1080 primary rays, one frame per rank, no exchange",
                                                "ms_per_step": round(ms_f, 4), "value": round(world * Wf * Hf / (ms_f * 1e-3) / 1e6, 1),
                                                "unit": "Mrays/s", "scaling": "weak"}
        scene.set_param("image_width", hint)
        del rays_f, hits_f

    if R == world and (world == 1 or comm is not None):
        # The same frame through the one-object-per-rank form of the step (rt_tiled_create / _redeal / _render): round-robin
        # deal, one re-deal by measured cost, then one frame at a time on one stream, pixels as the gathered record.  Last of
        # the extras: with N > 1 it is the first time this object meets more than one rank.
        try:
            tscene = new_scene()
            tframe = rt.TiledFrame(tscene, comm, rank, world, W, H, a.tile, full_h, local, float(head[0]) if world > 1 else 0.0)
            moved_t = tframe.redeal(cam)
            fb_t = torch.full((W * H, 1), -1, dtype=torch.int32, device="cuda") if rank == 0 else None

            def step_t():
                tframe.render(cam, fb_t.data_ptr() if rank == 0 else None, null_stream)
            for _ in range(max(3, a.warmup)):
                step_t()
            w_t, _ = timed(dist, sub, step_t)
            tt = torch.tensor([w_t], dtype=torch.float64, device=red_dev)
            if dist:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ms_t = float(tt.item()) * 1e3 / sub
            same = None
            if rank == 0:
                ref_t = torch.empty((W * H, 1), dtype=torch.int32, device="cuda")
                scene.set_param("image_width", W)
                scene.trace_pass_bgra8(full_d.data_ptr(), None, ref_t.data_ptr(), W * H, cam, 0, null_stream)
                torch.cuda.synchronize()
                scene.set_param("image_width", hint)
                same = bool(torch.equal(fb_t, ref_t))
                del ref_t
            extra["rt_tiled_object"] = {"ms_per_step": round(ms_t, 4), "value": round(W * H / (ms_t * 1e-3) / 1e6, 1), "unit": "Mrays/s",
                                        "frames_in_flight": 1, "squares_moved_by_its_redeal": moved_t, "rays_of_rank0": tframe.info()["rays"],
                                        "framebuffer_equals_one_gpu": same}
            tframe.close()
            tscene.close()
        except rt.RtError as e:
            extra["rt_tiled_object"] = {"failed": str(e)}
    if deadline is not None:
        deadline.cancel()
    emit()
    if dist:
        dist.barrier()
    if comm is not None:
        comm.close()
    for group in scene_sets.values():
        for sc in group:
            sc.close()
    scene.close()
    if dist:
        dist.destroy_process_group()


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    if a.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    red_dev = "cuda" if a.backend == "nccl" else "cpu"   # where the scalar reductions below live
    workload = a.workload or ("dragon1080" if world == 1 else "tiles4k")
    if workload == "mirror5":
        if world != 1:
            raise SystemExit("--workload mirror5 runs on one GPU")
        run_mirror5(a, local)
        return
    if workload == "tiles4k":
        run_tiles4k(a, rank, world, local, dist, red_dev)
        return

    model, W, H = WORKLOADS[workload]
    mesh = rt.Mesh.load_obj(os.path.join(rt.MODELS_DIR, model), "tolerant")
    bvh = rt.BVH(mesh, a.bvh)
    width = 4 if a.width == 4 else 2

    def make_scene(tree):
        sc = rt.Scene(mesh, tree, device=local, width=width)
        sc.set_param("width", width)
        sc.set_param("feedback", a.feedback)
        if workload == "ao" and width == 4:
            # incoherent rays: the kernel whose waves stream through the rays and refill their idle lanes (40 idle lanes trigger it)
            sc.set_param("refill", 40)
        for kv in a.param:
            k, v = kv.split("=")
            sc.set_param(k, int(v))
        return sc

    scene = make_scene(bvh)
    refill_on = workload == "ao" and width == 4 and not any(kv == "refill=0" for kv in a.param)
    info = scene.info()
    box = bvh.root_box
    stream = torch.cuda.current_stream().cuda_stream
    extra = {}

    # ---- synthetic input, generated on the host once, resident in HBM before timing ----
    all_lists = None
    if workload == "ao":
        n_local = a.rays
        # generated and uploaded in slices: the host table of 64 Mi records would be 4.3 GB
        rays_d = torch.empty((n_local, 16), dtype=torch.float32, device="cuda")
        piece = 4 * 1024 * 1024
        rays_h = None
        for off in range(0, n_local, piece):
            part = rt.ao_rays(mesh, box, 1234, rank * n_local + off, min(piece, n_local - off))
            if off == 0:
                rays_h = part            # the first slice stays on the host for the CPU sample and the fetch counter
            rays_d[off:off + len(part)].copy_(torch.from_numpy(part))
        frames, origins, tables = 1, [], [rays_d]
    else:
        origin = rt.frame_camera(box, W, H)
        origin[0] += rank * 0.01 * float(box[1, 0] - box[0, 0])  # each rank renders its own frame
        rays_h = rt.primary_rays(W, H, origin)
        if a.image_hint:
            scene.set_param("image_width", W)
        n_local = len(rays_h)
        rays_d = torch.from_numpy(rays_h).cuda()
        # A render loop never traces the same rays twice: frame k of the timed loop looks from origin_k, a camera
        # that drifts sideways by 0.2 % of the scene width per frame and returns after `frames` frames.
        frames = max(1, a.frames) if workload in ("dragon1080", "frames", "bunny1080", "dragon4k") else 1
        origins, tables = [], [rays_d]
        for k in range(frames):
            o = origin.copy()
            o[0] += k * 0.002 * float(box[1, 0] - box[0, 0])
            origins.append(o)
            if k > 0 and a.raygen == "table":
                tables.append(torch.from_numpy(rt.primary_rays(W, H, o)).cuda())
    step_no = [0]
    full_out = a.out == "full"
    out_b = 64 if full_out else 16
    hits_d = torch.empty((n_local, 16 if full_out else 4), dtype=torch.float32, device="cuda")

    def step():
        k = step_no[0] % frames
        step_no[0] += 1
        rays_k = tables[k] if k < len(tables) else rays_d
        if a.raygen == "device" and W:
            scene.trace_primary(W, H, origins[k], hits_d.data_ptr() if full_out else None,
                                None if full_out else hits_d.data_ptr(), stream)
        elif full_out:
            scene.intersect_async(rays_k.data_ptr(), n_local, hits_d.data_ptr(), None, stream)
        else:
            scene.intersect_async(rays_k.data_ptr(), n_local, None, hits_d.data_ptr(), stream)

    idle_gpu = None
    if a.preheat_ms > 0:
        if world == 1 and not a.no_compare:
            # the same W warm-up and K timed steps on the card as this process found it, with a scene object of their own (so the
            # steps timed for `value` below still start from a scene no launch has touched)
            scene_run, scene = scene, make_scene(bvh)
            if workload != "ao" and a.image_hint:
                scene.set_param("image_width", W)
            for _ in range(a.warmup):
                step()
            w_idle, _ = timed(None, a.steps, step)
            scene.close()
            scene = scene_run
            step_no[0] = 0
            idle_gpu = {"ms_per_step": round(w_idle * 1e3 / max(1, a.steps), 4), "value": round(n_local / (w_idle / max(1, a.steps)) / 1e6, 1),
                        "unit": "Mrays/s", "what": "the same warm-up and timed steps before the card was brought to its sustained clocks"}
        preheat(a.preheat_ms)
        extra["gpu_preheat_ms"] = a.preheat_ms
        extra["idle_gpu"] = idle_gpu
    for _ in range(a.warmup):
        step()
    wall, dev_ms = timed(dist, a.steps, step)
    if dist:
        tt = torch.tensor([wall], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = float(tt.item())
        nn = torch.tensor([n_local], dtype=torch.float64, device=red_dev)
        dist.all_reduce(nn, op=dist.ReduceOp.SUM)
        n_total = int(nn.item())
    else:
        n_total = n_local
    ms_per_step = wall * 1e3 / max(1, a.steps)
    mrays = n_total / (ms_per_step * 1e-3) / 1e6

    if rank == 0:
        # ---- algorithmic bytes per ray: records the traversal fetches, counted exactly by the instrumented kernel ----
        n_count = min(n_local, 1 << 24)
        cnt = scene.count_fetches_device(rays_d.data_ptr(), n_count)
        for more_rays in tables[1:]:   # records fetched: the mean over the frames the loop cycles through
            more = scene.count_fetches_device(more_rays.data_ptr(), n_count)
            cnt = {key: (max(cnt[key], more[key]) if key == "max_stack" else cnt[key] + more[key]) for key in cnt}
        stride = max(1, len(rays_h) // 8192)
        cnt_cpu = scene.count_fetches(rays_h, stride)
        node_b = info["node4_bytes"] if width == 4 else info["node2_bytes"]
        n_node, n_tri = cnt["nodes"] / cnt["rays"], cnt["tris"] / cnt["rays"]
        in_b = 0 if (a.raygen == "device" and W) else 64
        stream_b = in_b + out_b       # compulsory HBM bytes per ray
        trav_b = n_node * node_b + n_tri * info["tri_bytes"]
        secs = dev_ms * 1e-3
        key = f"{workload}/{a.bvh}/w{width}/out{out_b}/{a.raygen}" + ("/refill" if refill_on else "")
        pmc_table, pmc_src = newest_profile_table("pmc_table.json")
        pmc = pmc_table.get(key) if world == 1 else None
        # counters count for the kernels they were measured on: a table stamped with another source hash (or none) is named
        # but not used, and the line falls back to the byte roofline
        counters_head = None
        if pmc:
            stamp = pmc.get("code") or {}
            counters_head = {"git": stamp.get("git"), "kernel_source_sha256": stamp.get("kernel_source_sha256"),
                             "tree_kernel_source_sha256": kernel_source_hash()}
            counters_head["current"] = counters_head["kernel_source_sha256"] == counters_head["tree_kernel_source_sha256"]
            if not counters_head["current"]:
                pmc = None
        peaks, _ = newest_profile_table("l2_gather_peak.json")
        res = {"hbm": {"achieved": round(stream_b * n_local / secs / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "what": f"compulsory streams: {in_b} B ray in + {out_b} B hit out"}}
        res["hbm"]["frac"] = round(res["hbm"]["achieved"] / HBM_PEAK_GBS, 4)
        traffic = None
        if pmc:
            traffic = round((2 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024 / 1e9, 4)
            res["hbm"]["traffic_over_compulsory"] = round(traffic * 1e9 / (stream_b * n_local), 3)
            if peaks.get("l2_requests_per_ns_chip"):
                req = pmc["TCP_TCC_READ_REQ_sum"] / secs / 1e9
                res["l2"] = {"achieved": round(req, 2), "peak": peaks["l2_requests_per_ns_chip"], "unit": "read requests/ns",
                             "frac": round(req / peaks["l2_requests_per_ns_chip"], 4),
                             "what": "L1-miss read requests (TCP_TCC_READ_REQ) per launch / launch time, against the request "
                                     "rate of a divergent 16-byte gather over an L2-resident table on this chip",
                             "l1_hit_share": round(1.0 - pmc["TCP_TCC_READ_REQ_sum"] / pmc["TCP_TOTAL_CACHE_ACCESSES_sum"], 4)}
            # a SIMD issues one wave64 vector instruction per 2 cycles at best (32 lanes per cycle)
            issue = pmc["SQ_INSTS_VALU"] * 2.0 / secs
            res["valu"] = {"achieved": round(issue / 1e9, 1), "peak": round(SIMDS * CLOCK_HZ / 1e9, 1), "unit": "G SIMD issue cycles/s",
                           "frac": round(issue / (SIMDS * CLOCK_HZ), 4),
                           "what": "vector instructions per launch (SQ_INSTS_VALU) x the 2 cycles a wave64 instruction occupies a "
                                   "32-lane SIMD at best / launch time, against 1024 SIMDs x 2.4 GHz",
                           "instructions_per_wave": round(pmc["SQ_INSTS_VALU"] / pmc["SQ_WAVES"], 1),
                           "lane_utilisation": round(pmc["SQ_THREAD_CYCLES_VALU"] / (pmc["SQ_ACTIVE_INST_VALU"] * 64.0), 4),
                           # diagnostic, not a roofline: waves' VALU-active time overlaps within a SIMD, so this can pass 1
                           "busy_counter_over_simd_cycles": round(pmc["SQ_ACTIVE_INST_VALU"] * 4.0 / secs / (SIMDS * CLOCK_HZ), 4)}
        busiest = max(res, key=lambda k: res[k]["frac"])
        ops = n_node * width * 24 + n_tri * 52     # lane operations the algorithm needs: 24 per box test, 52 per triangle test
        alg_gbs = (stream_b + trav_b) * n_local / secs / 1e9
        roofline = {
            # the byte roofline: the bytes that have to cross HBM (the ray and hit streams) against the HBM3E peak; the
            # other resources' fractions are under "resources" and the busiest of them is named beside it
            "bound": "hbm", "achieved": res["hbm"]["achieved"], "peak": res["hbm"]["peak"], "unit": res["hbm"]["unit"],
            "frac": res["hbm"]["frac"], "traffic": traffic, "traffic_unit": "GB per launch, HBM side (2 x FETCH_SIZE + WRITE_SIZE)",
            "busiest_resource": busiest, "busiest_resource_frac": res[busiest]["frac"],
            # every algorithmic byte (streams + node and triangle records fetched) per launch over the launch time: most of it
            # is served by L1 and L2, so it is held against the L2 peak as well as the HBM peak
            "algorithmic_bytes": {"per_launch_GB": round((stream_b + trav_b) * n_local / 1e9, 4), "GBs": round(alg_gbs, 1),
                                  "over_hbm_peak": round(alg_gbs / HBM_PEAK_GBS, 4), "l2_peak_GBs": L2_PEAK_GBS,
                                  "over_l2_peak": round(alg_gbs / L2_PEAK_GBS, 4)},
            "resources": res, "counters_from": pmc_src if pmc else None, "counters_key": key, "counters_head": counters_head,
            "kernel": "RefillHitKernel" if refill_on else "ClosestHitKernel", "kernel_ms": round(dev_ms, 4),
            # the algorithmic traversal figure, never compared with the HBM peak: most of it is served by L1/L2
            "algorithmic": {"bytes_per_ray": round(stream_b + trav_b, 1), "stream_bytes_per_ray": stream_b,
                            "node_records_per_ray": round(n_node, 3), "tri_records_per_ray": round(n_tri, 3),
                            "node_bytes": node_b, "tri_bytes": info["tri_bytes"],
                            "node_records_per_ray_cpu_sample": round(cnt_cpu["nodes"] / max(1, cnt_cpu["rays"]), 3),
                            "tri_records_per_ray_cpu_sample": round(cnt_cpu["tris"] / max(1, cnt_cpu["rays"]), 3),
                            "traversal_GBs": round(trav_b * n_local / secs / 1e9, 1),
                            "hit_fraction": round(cnt["rays_hit"] / cnt["rays"], 4), "deepest_stack": cnt["max_stack"],
                            "valu_lane_ops_per_ray": round(ops, 1),
                            "valu_lane_ops_frac_of_peak": round(ops * n_local / secs / (SIMDS * 32 * CLOCK_HZ), 4)},
        }

        # the same launch on the library's surface-area tree (an extension: same node format, fewer records per ray)
        other = None
        if world == 1 and a.bvh == "fixed" and not a.no_compare and workload in ("dragon1080", "bunny1080", "dragon4k"):
            scene_s = make_scene(rt.BVH(mesh, "sah"))
            if a.image_hint:
                scene_s.set_param("image_width", W)
            out_args = (hits_d.data_ptr(), None) if full_out else (None, hits_d.data_ptr())
            k_no = [0]

            def step_s():
                k_no[0] += 1
                scene_s.intersect_async(tables[k_no[0] % len(tables)].data_ptr(), n_local, out_args[0], out_args[1], stream)
            for _ in range(a.warmup):
                step_s()
            w_s, _ = timed(None, a.steps, step_s)
            sms = w_s * 1e3 / max(1, a.steps)
            other = {"bvh": "sah", "ms_per_step": round(sms, 4), "value": round(n_local / (sms * 1e-3) / 1e6, 1), "unit": "Mrays/s"}
            scene_s.close()

        # the same frames with the rays generated in the kernel (rt_trace_primary: no 64-byte ray record is read)
        in_kernel = None
        if world == 1 and W and workload in ("dragon1080", "bunny1080", "dragon4k") and a.raygen == "table" and not a.no_compare:
            k_no = [0]

            def step_g():
                k_no[0] += 1
                scene.trace_primary(W, H, origins[k_no[0] % len(origins)], hits_d.data_ptr() if full_out else None,
                                    None if full_out else hits_d.data_ptr(), stream)
            for _ in range(a.warmup):
                step_g()
            w_g, _ = timed(None, a.steps, step_g)
            gms = w_g * 1e3 / max(1, a.steps)
            in_kernel = {"ms_per_step": round(gms, 4), "value": round(n_local / (gms * 1e-3) / 1e6, 1), "unit": "Mrays/s",
                         "what": "rt_trace_primary: the same camera rays computed in the kernel, bit-identical to the table, instead of read from HBM"}

        # the same steps issued alternately on two streams (two frames in flight, each with its own hit buffer): the tail of
        # one launch overlaps the head of the next.  Reported beside `value`, never as it: launches overlap, so there is no
        # per-launch duration to hold against a profile.
        overlapped = None
        if world == 1 and W and a.raygen == "table" and not a.no_compare:
            s2 = [torch.cuda.Stream(), torch.cuda.Stream()]
            hits2 = [hits_d, torch.empty_like(hits_d)]
            k_no = [0]

            def step_2():
                i = k_no[0] & 1
                k_no[0] += 1
                out = hits2[i].data_ptr()
                scene.intersect_async(tables[k_no[0] % len(tables)].data_ptr(), n_local, out if full_out else None,
                                      None if full_out else out, s2[i].cuda_stream)
            for _ in range(a.warmup):
                step_2()
            w_2, _ = timed(None, a.steps, step_2)
            ms2 = w_2 * 1e3 / max(1, a.steps)
            overlapped = {"streams": 2, "ms_per_step": round(ms2, 4), "value": round(n_local / (ms2 * 1e-3) / 1e6, 1), "unit": "Mrays/s",
                          "what": "the same launches issued alternately on two streams, two frames in flight"}
            del hits2

        # the drop-in call with HOST buffers (pageable memory both ways): PCIe-inclusive, reported beside `value`, never as it
        host_call = None
        if world == 1 and W and full_out and not a.no_compare:
            hits_h = np.zeros_like(rays_h)
            for _ in range(2):
                rt._check(rt.lib.ray_scene_intersection(rt._ptr(rays_h), n_local, scene._h, rt._ptr(hits_h)))
            c0 = time.perf_counter()
            for _ in range(5):
                rt._check(rt.lib.ray_scene_intersection(rt._ptr(rays_h), n_local, scene._h, rt._ptr(hits_h)))
            hms = (time.perf_counter() - c0) / 5 * 1e3
            host_call = {"ms_per_call": round(hms, 3), "value": round(n_local / (hms * 1e-3) / 1e6, 1), "unit": "Mrays/s",
                         "pcie_GBs": round(128 * n_local / (hms * 1e-3) / 1e9, 1),
                         "what": "ray_scene_intersection(host rays, n, scene, host hits): 64 B up + 64 B down per ray over PCIe included"}
            del hits_h

        step_no[0] = 0
        step()                      # leaves frame 0's hits in hits_d for the spot check below
        torch.cuda.synchronize()
        cpu = None
        if world == 1 and not a.no_cpu_baseline:
            import oracle  # the CPU checker, used here only as the timed baseline
            if workload == "ao":
                every = max(1, len(rays_h) // 16384)
                sel = np.arange(0, len(rays_h), every)
                what = f"{len(sel)} of the {n_local} rays (every {every}th of the first {len(rays_h)}) x {len(mesh.faces)} triangles, brute force"
            else:
                s = a.cpu_stride
                sel = (np.arange(0, H, s)[:, None] * W + np.arange(0, W, s)[None, :]).reshape(-1)
                what = f"stride-{s} sub-grid of the {W}x{H} ray set ({len(sel)} rays x {len(mesh.faces)} triangles, brute force)"
            sample = rays_h[sel]
            threads = oracle.default_threads()
            c0 = time.perf_counter()
            oid, ot, _, _ = oracle.closest_hit(mesh.verts, mesh.faces, sample, threads=threads)
            cdt = time.perf_counter() - c0
            # parity spot-check of the same sample against the GPU result of the last step
            got = hits_d[torch.from_numpy(sel).cuda()].cpu().numpy()
            gid = got[:, 14 if full_out else 0].view(np.uint32)
            gt = got[:, 15 if full_out else 1]
            hit = oid != rt.MISS_ID
            cpu = {"value": round(len(sample) / cdt / 1e6, 6), "unit": "Mrays/s", "cores": threads, "kind": "port",
                   "sample": what, "seconds": round(cdt, 2), "id_mismatches_vs_gpu": int((gid != oid).sum()),
                   "t_bit_mismatches_vs_gpu": int((gt[hit].view(np.uint32) != ot[hit].view(np.uint32)).sum())}

        # The accelerated CPU figure: the oracle's walk of the same CentroidBVH tree over EVERY ray of the step (the first 4 Mi of the
        # AO set), and with it a comparison of every hit of the last step with the CPU's.
        cpu_bvh = None
        if world == 1 and not a.no_cpu_baseline:
            import oracle
            all_rays = rays_h
            threads = oracle.default_threads()
            c0 = time.perf_counter()
            oid, ot, _, _ = oracle.closest_hit_bvh(mesh.verts, mesh.faces, bvh.nodes, all_rays, threads=threads)
            cdt = time.perf_counter() - c0
            hit = oid != rt.MISS_ID
            got = hits_d[:len(all_rays)].cpu().numpy()
            gid = got[:, 14 if full_out else 0].view(np.uint32)
            gt = got[:, 15 if full_out else 1]
            cpu_bvh = {"value": round(len(all_rays) / cdt / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
                       "sample": f"all {len(all_rays)} rays" + (f" of the first slice of the {n_local}" if len(all_rays) != n_local else "") +
                                 ", closest hit through the same CentroidBVH tree (oracle tree walk)",
                       "seconds": round(cdt, 2), "id_mismatches_vs_gpu": int((gid != oid).sum()),
                       "t_bit_mismatches_vs_gpu": int((gt[hit].view(np.uint32) != ot[hit].view(np.uint32)).sum())}
            del all_rays

        # The N = 1 point of the multi-GPU series: with N > 1 this script times ONE dragon 3840x2160 frame cut into squares
        # (--workload tiles4k, strong scaling), so the headline run carries the same step on this one GPU, run as a child process
        # once everything above is done.  Never `value`.
        if world == 1 and workload == "dragon1080" and not a.no_compare and not a.no_cpu_baseline:
            import subprocess
            try:
                child = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", "tiles4k", "--steps", "60", "--warmup", "10",
                                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=150)
                t4 = json.loads(child.stdout.strip().splitlines()[-1])
                extra["sharded_4k_frame_on_one_gpu"] = {
                    "workload": t4["config"]["workload"], "value": t4["value"], "unit": t4["unit"], "ms_per_step": t4["ms_per_step"],
                    "frames_in_flight": t4.get("frames_in_flight"), "gathered_record": t4.get("gathered_record"),
                    "rt_tiled_object": t4.get("rt_tiled_object"),
                    "what": "python bench.py --workload tiles4k on this GPU: the N = 1 point of what --gpus N > 1 measures"}
            except Exception as e:      # reported, never fatal to the headline line
                extra["sharded_4k_frame_on_one_gpu"] = {"failed": f"{type(e).__name__}: {e}"[:300]}

        if W:
            what = f"{model} {W}x{H} primary rays"
        else:
            what = f"{model} {n_local} AO rays per GPU"
        line = {
            "metric": "closest-hit Mrays/s", "value": round(mrays, 1), "unit": "Mrays/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": what, "name": workload, "rays_per_gpu": n_local, "rays_total": n_total, "triangles": len(mesh.faces),
                       "loader": "tolerant",
                       "bvh": {"fixed": "core/CentroidBVH build, centroid index repaired", "reference": "core/CentroidBVH build as written",
                               "sah": "library surface-area build"}[a.bvh],
                       "bvh_width": width, "nodes": info["nodes4" if width == 4 else "nodes2"],
                       "hit_record": "64B RayIntersection" if out_b == 64 else "16B {id,t,u,v}",
                       "frames": frames,
                       "rays": "RayIntersection table resident in HBM" if in_b else "generated in the kernel (rt_trace_primary)",
                       "dispatch_order": ("a fixed grid of waves streams through the rays in index order, lanes take new rays as they finish (refill)" if refill_on
                                          else "costliest work unit first, from the costs a launch 2 to 10 launches earlier measured (re-sorted every 8th launch)"
                                          if a.feedback else "image centre first"),
                       "parallelism": f"{world} x one process per GPU, frame-parallel"},
            "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_bvh": cpu_bvh, "same_launch_on_sah_tree": other, "rays_generated_in_kernel": in_kernel, "two_frames_in_flight": overlapped,
            "host_pointer_call": host_call,
        }
        line.update(extra)
        print(json.dumps(line), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
