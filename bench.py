#!/usr/bin/env python3
"""bench.py -- closest-hit Mrays/s on MI355X.

A "step" is one pass of the hot path (ray_scene_intersection's device form,
rt_intersect_async) over one batch of rays that is already resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] ...

Workloads:
  dragon1080   dragon.obj 1920x1080 primary rays           (default; the headline metric)
  bunny1080    bunny.obj 1920x1080 primary rays
  dragon4k     dragon.obj 3840x2160 primary rays
  ao           bunny_big.obj incoherent AO-style rays (--rays, default 16 Mi per GPU)
  mirror5      cube_mirror_test.obj 1920x1080: one step = the primary pass + four reflection passes of the
               full `trace` kernel (rt_trace_pass, records ping-ponged on the device); rays counted = 5 x W x H
  tiles4k      dragon.obj 3840x2160 split by interleaved 64x64 tiles over the ranks, with the
               compact framebuffer gathered to rank 0 over RCCL (strong scaling; the gather is
               timed inside the step and also reported on its own)

With N > 1 (one process per GPU under torch.distributed.run) every rank traces its own
frame of the workload (rank r's camera is shifted sideways by r steps), so per-GPU work is
fixed and the job's rays/s is the sum: weak scaling, no data-path collective.  `tiles4k` is
the sharded-image form with its one exchange step.

Rank 0 prints ONE JSON line.  `roofline.achieved` = algorithmic bytes per launch / mean
launch duration (HIP events on the launch stream).  `cpu_baseline` times the CPU oracle
(brute force over all triangles, the reference's cpu_raycast loop) on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch  # before raytrace_amd: both must bind the same HIP runtime

import raytrace_amd as rt
from raytrace_amd.tiles import tile_indices

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak

WORKLOADS = {
    "dragon1080": ("dragon.obj", 1920, 1080),
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
    ap.add_argument("--workload", default="dragon1080", choices=sorted(WORKLOADS))
    ap.add_argument("--width", type=int, default=int(os.environ.get("RT_WIDTH", "0")), help="2 or 4 (0 = library default)")
    ap.add_argument("--rays", type=int, default=16 * 1024 * 1024, help="ray count for --workload ao")
    ap.add_argument("--bvh", default="sah", choices=["fixed", "reference", "sah"],
                    help="tree handed to rt_scene_create: the reference's centroid build as written (reference), with its "
                         "centroid indexing repaired (fixed), or the surface-area build in the same node format (sah)")
    ap.add_argument("--out", default="full", choices=["full", "compact"], help="64-byte hit records or 16-byte {id,t,u,v}")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-compare", action="store_true", help="skip the extra timing on the centroid tree")
    ap.add_argument("--cpu-stride", type=int, default=4, help="sub-grid stride (both axes) of the CPU baseline sample")
    ap.add_argument("--tile", type=int, default=64)
    ap.add_argument("--extra-stack", type=int, default=0, help="diagnostic: extra LDS per lane to lower occupancy")
    ap.add_argument("--quant", type=int, default=-1, help="1: 16-bit quantized boxes, 0: float boxes (-1 = library default)")
    ap.add_argument("--block", type=int, default=0, help="workgroup threads: 64, 128 or 256 (0 = library default)")
    ap.add_argument("--image-hint", type=int, default=1, help="declare the ray array as a W-wide image (8x8 tiles per wave)")
    ap.add_argument("--waves-per-cu", type=int, default=-1, help="resident-wave cap per CU (-1 = library default)")
    ap.add_argument("--raygen", default="table", choices=["table", "device"],
                    help="table: rays are a RayIntersection array resident in HBM (the reference's interface); device: "
                         "the same primary rays are generated inside the kernel (rt_trace_primary), no ray array is read")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the N > 1 flow on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--frames", type=int, default=4,
                    help="image workloads: distinct camera positions (ray tables, all resident in HBM) the steps cycle "
                         "through, so that no step repeats the previous step's rays")
    ap.add_argument("--feedback-radius", type=int, default=-1, help="-1 = library default")
    ap.add_argument("--feedback-share", type=int, default=-1, help="-1 = library default")
    ap.add_argument("--param", action="append", default=[], metavar="KEY=VALUE",
                    help="extra rt_scene_set_param settings (repeatable), applied last")
    ap.add_argument("--feedback", type=int, default=1,
                    help="1: each launch starts its work units costliest-first, by the costs measured two launches earlier")
    return ap.parse_args()


def measured_traffic(workload, out_bytes, world):
    """(GB per launch, file) from the newest profiles/r*/hbm_traffic.json that has this workload, else (None, None)."""
    if world != 1:
        return None, None
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    key = f"{workload}/out{out_bytes}"
    for d in sorted((x for x in os.listdir(root) if x.startswith("r")), reverse=True) if os.path.isdir(root) else []:
        path = os.path.join(root, d, "hbm_traffic.json")
        if os.path.exists(path):
            with open(path) as f:
                table = json.load(f)
            if key in table:
                return table[key]["gb_per_launch"], f"profiles/{d}/hbm_traffic.json"
    return None, None


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
    red_dev = "cuda" if a.backend == "nccl" else "cpu"   # where the two scalar reductions below live

    model, W, H = WORKLOADS[a.workload]
    mesh = rt.Mesh.load_obj(os.path.join(rt.MODELS_DIR, model), "reference" if a.workload == "mirror5" else "tolerant")
    bvh = rt.BVH(mesh, a.bvh)
    scene = rt.Scene(mesh, bvh, device=local, width=4 if a.width == 4 else 2)
    if a.width in (2, 4):
        scene.set_param("width", a.width)
    if a.extra_stack > 0:
        scene.set_param("extra_stack", a.extra_stack)
    if a.quant >= 0:
        scene.set_param("quantized", a.quant)
    if a.block > 0:
        scene.set_param("block_threads", a.block)
    if W and a.workload != "tiles4k" and a.image_hint:
        scene.set_param("image_width", W)
    elif a.workload == "tiles4k" and a.image_hint and W % a.tile == 0:
        # a rank's rays are its tiles one after another, each row-major and a.tile pixels wide: to the library that
        # is an image a.tile wide and (tiles x rows) tall, so waves still take 8x8 pixel blocks
        scene.set_param("image_width", a.tile)
    scene.set_param("feedback", a.feedback)
    if a.feedback_radius >= 0:
        scene.set_param("feedback_radius", a.feedback_radius)
    if a.feedback_share >= 0:
        scene.set_param("feedback_share", a.feedback_share)
    if a.waves_per_cu >= 0:
        scene.set_param("waves_per_cu", a.waves_per_cu)
    for kv in a.param:
        k, v = kv.split("=")
        scene.set_param(k, int(v))
    info = scene.info()
    box = bvh.root_box

    # ---- synthetic input, generated on the host once, resident in HBM before timing ----
    if a.workload == "ao":
        n_local = a.rays
        rays_h = rt.ao_rays(mesh, box, 1234, rank * n_local, n_local)
        own = None
    else:
        origin = rt.frame_camera(box, W, H)
        if a.workload == "tiles4k":
            full = rt.primary_rays(W, H, origin)
            own = tile_indices(W, H, a.tile, rank, world)
            rays_h = full[own]
            del full
        else:
            origin[0] += rank * 0.01 * float(box[1, 0] - box[0, 0])  # each rank renders its own frame
            rays_h = rt.primary_rays(W, H, origin)
            own = None
        n_local = len(rays_h)
    rays_d = torch.from_numpy(rays_h).cuda()
    # A render loop never traces the same rays twice: frame k of the timed loop looks from origin_k, a camera
    # that drifts sideways by 0.2 % of the scene width per frame and returns after `frames` frames.
    frames = max(1, a.frames) if a.workload in ("dragon1080", "bunny1080", "dragon4k") else 1
    origins, tables = [], [rays_d]
    if a.workload != "ao":
        for k in range(frames):
            o = origin.copy()
            o[0] += k * 0.002 * float(box[1, 0] - box[0, 0])
            origins.append(o)
            if k > 0 and a.raygen == "table":
                tables.append(torch.from_numpy(rt.primary_rays(W, H, o)).cuda())
    step_no = [0]
    full_out = a.out == "full" or a.workload == "mirror5"
    hits_d = torch.empty((n_local, 16 if full_out else 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    gather_bufs = None
    if a.workload == "tiles4k" and world > 1:
        counts = [len(tile_indices(W, H, a.tile, r, world)) for r in range(world)]
        if rank == 0:
            gather_bufs = [torch.empty((c, 4), dtype=torch.float32, device="cuda") for c in counts]
        hits16_d = torch.empty((n_local, 4), dtype=torch.float32, device="cuda")

    passes = 5 if a.workload == "mirror5" else 1
    if a.workload == "mirror5":
        pong = torch.empty((n_local, 16), dtype=torch.float32, device="cuda")
        rgba_d = torch.empty((n_local, 4), dtype=torch.float32, device="cuda")
        cam = np.eye(4, dtype=np.float32)

    def step():
        k = step_no[0] % frames
        step_no[0] += 1
        rays_k = tables[k] if k < len(tables) else rays_d
        if a.workload == "mirror5":
            # pass 0 reads the primary table; passes 1..4 reflect the previous pass's records (ping-pong)
            scene.trace_pass(rays_d.data_ptr(), hits_d.data_ptr(), rgba_d.data_ptr(), n_local, cam, 0, stream)
            src, dst = hits_d, pong
            for _ in range(4):
                scene.trace_pass(src.data_ptr(), dst.data_ptr(), rgba_d.data_ptr(), n_local, cam, 1, stream)
                src, dst = dst, src
        elif a.workload == "tiles4k" and world > 1:
            scene.intersect_async(rays_d.data_ptr(), n_local, None, hits16_d.data_ptr(), stream)
            # one exchange: compact hit tiles to rank 0 (point-to-point sends, not a ring)
            if rank == 0:
                gather_bufs[0].copy_(hits16_d)
                reqs = [dist.irecv(gather_bufs[r], src=r) for r in range(1, world)]
                for q in reqs:
                    q.wait()
            else:
                dist.isend(hits16_d, dst=0).wait()
        elif a.raygen == "device" and W:
            scene.trace_primary(W, H, origins[k], hits_d.data_ptr() if full_out else None,
                                None if full_out else hits_d.data_ptr(), stream)
        elif full_out:
            scene.intersect_async(rays_k.data_ptr(), n_local, hits_d.data_ptr(), None, stream)
        else:
            scene.intersect_async(rays_k.data_ptr(), n_local, None, hits_d.data_ptr(), stream)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
        torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(a.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
        torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1) / max(1, a.steps)  # mean launch duration on the launch stream
    dev_ms /= passes
    if dist:
        tt = torch.tensor([wall], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = float(tt.item())
        nn = torch.tensor([n_local], dtype=torch.float64, device=red_dev)
        dist.all_reduce(nn, op=dist.ReduceOp.SUM)
        n_total = int(nn.item())
    else:
        n_total = n_local
    n_total *= passes
    ms_per_step = wall * 1e3 / max(1, a.steps)
    mrays = n_total / (ms_per_step * 1e-3) / 1e6

    if rank == 0:
        # ---- algorithmic bytes per ray: records the traversal fetches, counted exactly ----
        width = a.width if a.width in (2, 4) else 2
        if a.workload == "mirror5":
            scene.set_param("feedback", 0)   # the counting launch below is a primary pass only
        cnt = scene.count_fetches_device(rays_d.data_ptr(), n_local)
        for extra in tables[1:]:   # records fetched: the mean over the frames the loop cycles through
            more = scene.count_fetches_device(extra.data_ptr(), n_local)
            cnt = {key: cnt[key] + more[key] for key in cnt}
        stride = max(1, n_local // 8192)
        cnt_cpu = scene.count_fetches(rays_h, stride)
        node_b = info["node4_bytes"] if width == 4 else info["node2_bytes"]
        n_node, n_tri = cnt["nodes"] / cnt["rays"], cnt["tris"] / cnt["rays"]
        out_b = 64 if (full_out and not (a.workload == "tiles4k" and world > 1)) else 16
        in_b = 0 if (a.raygen == "device" and W and a.workload not in ("tiles4k", "mirror5")) else 64
        bytes_per_ray = in_b + out_b + n_node * node_b + n_tri * info["tri_bytes"]
        if a.workload == "mirror5":
            # reflection passes trace rays that exist only inside the kernel: their records are not counted, so
            # the figure below prices every pass at the streams alone (record in, record out, RGBA pixel)
            bytes_per_ray = 64 + 64 + 16
        achieved = bytes_per_ray * n_local / (dev_ms * 1e-3) / 1e9
        roofline = {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
            "kernel": "ClosestHitKernel", "kernel_ms": round(dev_ms, 4), "bytes_per_ray": round(bytes_per_ray, 1),
            "node_records_per_ray": round(n_node, 3), "tri_records_per_ray": round(n_tri, 3),
            "node_records_per_ray_cpu_sample": round(cnt_cpu["nodes"] / max(1, cnt_cpu["rays"]), 3),
            "tri_records_per_ray_cpu_sample": round(cnt_cpu["tris"] / max(1, cnt_cpu["rays"]), 3),
            "node_bytes": node_b, "tri_bytes": info["tri_bytes"], "ray_in_bytes": in_b, "hit_out_bytes": out_b,
            "hit_fraction": round(cnt["rays_hit"] / cnt["rays"], 4),
            # the two parts of the algorithmic figure on their own: the compulsory ray/hit streams (what HBM must
            # move) and the traversal records (served by L1/L2; the L2 sustains roughly 17-19 TB/s of gathers)
            "stream_GBs": round((in_b + out_b) * n_local / (dev_ms * 1e-3) / 1e9, 1),
            "traversal_GBs": round((bytes_per_ray - in_b - out_b) * n_local / (dev_ms * 1e-3) / 1e9, 1),
        }
        # The resource this kernel actually saturates is FP32 vector issue, not memory: algorithmic lane operations
        # (24 per box test, two boxes per node record; 52 per triangle test incl. the divide) against the rate at
        # which 256 CUs x 64 lanes retire one non-fused operation per clock at 2.4 GHz.  The issue and lane
        # utilisation measured by PMC counters are in the committed profile summary.
        ops_per_ray = n_node * (4 if width == 4 else 2) * 24 + n_tri * 52
        roofline["valu"] = {"algorithmic_ops_per_ray": round(ops_per_ray, 1),
                            "achieved_Tops": round(ops_per_ray * n_local / (dev_ms * 1e-3) / 1e12, 2),
                            "peak_Tops": round(256 * 64 * 2.4e9 / 1e12, 1),
                            "frac": round(ops_per_ray * n_local / (dev_ms * 1e-3) / (256 * 64 * 2.4e9), 4)}
        # HBM-side bytes per launch from PMC counters: measured in separate rocprofv3 passes (tools/profile_gpu.sh),
        # committed next to the kernel-trace summary, looked up here by workload and output record
        roofline["traffic"], roofline["traffic_source"] = measured_traffic(a.workload, out_b, world)
        roofline["traffic_unit"] = "GB per launch (2 x FETCH_SIZE + WRITE_SIZE)"

        # the same launch on the reference's own (repaired) centroid tree, for comparison
        centroid = None
        if world == 1 and a.bvh == "sah" and not a.no_compare and a.workload in ("dragon1080", "bunny1080", "dragon4k"):
            bvh_c = rt.BVH(mesh, "fixed")
            scene_c = rt.Scene(mesh, bvh_c, device=local, width=4 if a.width == 4 else 2)
            for k, v in (("quantized", a.quant), ("block_threads", a.block)):
                if v > 0 or (k == "quantized" and v == 0):
                    scene_c.set_param(k, v)
            if a.image_hint:
                scene_c.set_param("image_width", W)
            scene_c.set_param("feedback", a.feedback)
            for _ in range(a.warmup):
                scene_c.intersect_async(rays_d.data_ptr(), n_local, hits_d.data_ptr() if full_out else None,
                                        None if full_out else hits_d.data_ptr(), stream)
            torch.cuda.synchronize()
            c0 = time.perf_counter()
            for _ in range(a.steps):
                scene_c.intersect_async(rays_d.data_ptr(), n_local, hits_d.data_ptr() if full_out else None,
                                        None if full_out else hits_d.data_ptr(), stream)
            torch.cuda.synchronize()
            cms = (time.perf_counter() - c0) * 1e3 / max(1, a.steps)
            centroid = {"bvh": "fixed", "ms_per_step": round(cms, 4), "value": round(n_local / (cms * 1e-3) / 1e6, 1), "unit": "Mrays/s"}
            scene_c.close()

        step_no[0] = 0
        step()                      # leaves frame 0's hits in hits_d for the spot check below
        torch.cuda.synchronize()
        cpu = None
        if world == 1 and not a.no_cpu_baseline and a.workload != "mirror5":
            import oracle  # the CPU checker, used here only as the timed baseline
            if a.workload == "ao":
                sample = rays_h[:: max(1, n_local // 16384)]
                what = f"{len(sample)} of the {n_local} rays (every {max(1, n_local // 16384)}th)"
            else:
                s = a.cpu_stride
                grid = (np.arange(0, H, s)[:, None] * W + np.arange(0, W, s)[None, :]).reshape(-1)
                sample = rays_h[grid]
                what = f"stride-{s} sub-grid of the {W}x{H} ray set ({len(sample)} rays x {len(mesh.faces)} triangles, brute force)"
            threads = oracle.default_threads()
            c0 = time.perf_counter()
            oid, ot, _, _ = oracle.closest_hit(mesh.verts, mesh.faces, sample, threads=threads)
            cdt = time.perf_counter() - c0
            # parity spot-check of the same sample against the GPU result of the last step
            hits_h = hits_d.cpu().numpy()
            gid = hits_h[:, 14 if full_out else 0].view(np.uint32)
            sel = grid if a.workload != "ao" else np.arange(0, n_local, max(1, n_local // 16384))
            mism = int((gid[sel] != oid).sum()) if a.workload != "tiles4k" else None
            cpu = {"value": round(len(sample) / cdt / 1e6, 6), "unit": "Mrays/s", "cores": threads, "kind": "port",
                   "sample": what, "seconds": round(cdt, 2), "id_mismatches_vs_gpu": mism}

        line = {
            "metric": "closest-hit Mrays/s", "value": round(mrays, 1), "unit": "Mrays/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong" if a.workload == "tiles4k" else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{model} {W}x{H} primary pass + 4 reflection passes (full trace kernel)" if a.workload == "mirror5"
                       else f"{model} {W}x{H} primary rays" if W else f"{model} {n_local} AO rays per GPU",
                       "name": a.workload, "rays_per_gpu": n_local, "rays_total": n_total, "triangles": len(mesh.faces),
                       "loader": "tolerant", "bvh": a.bvh, "bvh_width": width, "nodes": info["nodes4" if width == 4 else "nodes2"],
                       "hit_record": "64B RayIntersection" if out_b == 64 else "16B {id,t,u,v}",
                       "frames": frames,
                       "rays": "RayIntersection table resident in HBM" if in_b else "generated in the kernel (rt_trace_primary)",
                       "dispatch_order": "costliest work unit first, from the costs measured two launches earlier" if a.feedback
                       else "image centre first",
                       "parallelism": f"{world} x frame-parallel" if a.workload != "tiles4k" else f"{world} x interleaved {a.tile}x{a.tile} tiles + RCCL gather"},
            "roofline": roofline, "cpu_baseline": cpu, "same_launch_on_centroid_tree": centroid,
        }
        print(json.dumps(line), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
