#!/usr/bin/env python3
"""Modelled per-rank cost balance (max / mean of the summed wave iteration counts) of the dragon 3840x2160 frame cut into
64x64 squares, for 2, 4 and 8 ranks: squares dealt by measured cost against squares dealt round-robin."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytrace_amd as rt  # noqa: E402
from raytrace_amd import tiles as tl  # noqa: E402

W, H, tile = 3840, 2160, 64
mesh = rt.Mesh.load_obj(os.path.join(rt.MODELS_DIR, "dragon.obj"), "tolerant")
bvh = rt.BVH(mesh, "fixed")
scene = rt.Scene(mesh, bvh, device=0, width=4)
costs = tl.tile_costs_from_units(scene.unit_costs_primary(W, H, rt.frame_camera(bvh.root_box, W, H)), W, H, tile)
tx, ty = tl.tile_grid(W, H, tile)
for n in (2, 4, 8):
    by_cost = tl.balance(tl.assign_tiles(tx * ty, n, costs), costs)
    rr = tl.balance(tl.assign_tiles(tx * ty, n, None), costs)
    print(f"{n} ranks: by cost {by_cost:.4f}  round-robin {rr:.4f}  ({tx * ty} squares)")
