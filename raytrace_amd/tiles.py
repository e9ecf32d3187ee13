"""Image-tile sharding for multi-GPU renders: which tiles a rank owns, which rays those are, and how
the gathered per-rank records go back into the row-major framebuffer.

The reference renders one image on one device; its only hint of tiling is a commented-out split of the
NDRange into four offsets (render_client/img_main.cpp:464-478).  Here a W x H image is cut into
tile x tile squares (edge squares smaller), numbered row-major.  Which rank traces which square follows
the squares' measured traversal cost (the per-8x8-tile wave iteration counts the kernel records for its
cost-feedback schedule, summed per square): squares are dealt costliest first, each to the rank with the
least cost so far, so the ranks finish together whatever the silhouette looks like.  Every rank derives
the same assignment from the same cost table; no cross-process queue is involved.  Without costs the
squares are dealt round-robin.
"""
from __future__ import annotations

import numpy as np


def tile_grid(width, height, tile):
    """(tiles_x, tiles_y) of a width x height image cut into tile x tile squares."""
    return (width + tile - 1) // tile, (height + tile - 1) // tile


def tile_shape(width, height, tile, k):
    """(x0, y0, w, h) of square k."""
    tx, _ = tile_grid(width, height, tile)
    x0, y0 = (k % tx) * tile, (k // tx) * tile
    return x0, y0, min(tile, width - x0), min(tile, height - y0)


def tile_costs_from_units(unit_costs, width, height, tile):
    """Sums the kernel's per-8x8-tile costs ((ceil(H/8), ceil(W/8)) array) over each tile x tile square.
    `tile` must be a multiple of 8.  Returns a float64 array of tiles_x * tiles_y costs, row-major."""
    assert tile % 8 == 0, "squares must be made of whole 8x8 work units"
    tx, ty = tile_grid(width, height, tile)
    u = np.asarray(unit_costs, dtype=np.float64)
    per = tile // 8
    out = np.zeros((ty, tx))
    for j in range(ty):
        for i in range(tx):
            out[j, i] = u[j * per:(j + 1) * per, i * per:(i + 1) * per].sum()
    return out.reshape(-1)


def assign_tiles(ntiles, world, costs=None, head_start=None):
    """Per-rank lists of square numbers.  With `costs` (one per square): longest-processing-time-first greedy --
    costliest square to the least-loaded rank (ties: lower rank), each rank's list in the order it was dealt
    (costliest first, which is also the order its waves should start in).  head_start[r] (optional) is the share
    of the whole frame's cost rank r counts as carrying before the deal.  Without costs: round-robin."""
    if costs is None:
        return [np.arange(r, ntiles, world, dtype=np.int64) for r in range(world)]
    costs = np.asarray(costs, dtype=np.float64)
    assert len(costs) == ntiles
    order = np.argsort(-costs, kind="stable")
    total = 0.0
    for k in range(ntiles):
        total += costs[k] + 1.0
    load = np.zeros(world) if head_start is None else np.asarray(head_start, dtype=np.float64) * total
    lists = [[] for _ in range(world)]
    for k in order:
        # every square carries a floor cost (its pixels have to be generated and stored even if nothing is hit)
        r = int(np.argmin(load))
        lists[r].append(int(k))
        load[r] += costs[k] + 1.0
    return [np.asarray(l, dtype=np.int64) for l in lists]


def balance(lists, costs):
    """max / mean of the per-rank cost sums of an assignment (1.0 = perfectly even)."""
    sums = np.array([float(np.asarray(costs, dtype=np.float64)[l].sum()) for l in lists])
    return float(sums.max() / sums.mean()) if sums.mean() > 0 else 1.0


def rays_of_tiles(width, height, tile, tile_ids):
    """Ray indices (row-major i + j*width) of the listed squares, square by square, row-major inside each;
    plus tile_first[i] = position of square i's first record in that order."""
    parts, first, pos = [], [], 0
    for k in tile_ids:
        x0, y0, w, h = tile_shape(width, height, tile, int(k))
        xs = np.arange(x0, x0 + w)
        ys = np.arange(y0, y0 + h)
        parts.append((ys[:, None] * width + xs[None, :]).reshape(-1))
        first.append(pos)
        pos += w * h
    idx = np.concatenate(parts) if parts else np.zeros(0, np.int64)
    return idx, np.asarray(first, dtype=np.uint32)


def tile_indices(width, height, tile, rank, world, costs=None):
    """Ray indices of the squares owned by `rank`."""
    tx, ty = tile_grid(width, height, tile)
    return rays_of_tiles(width, height, tile, assign_tiles(tx * ty, world, costs)[rank])[0]


def scatter_to_framebuffer(width, height, tile, per_rank_tiles, per_rank_records):
    """Host restatement of the device scatter (rt_scatter_tiles): per_rank_records[r] holds rank r's records in
    rays_of_tiles order for per_rank_tiles[r]."""
    first = per_rank_records[0]
    fb = np.zeros((width * height,) + tuple(first.shape[1:]), dtype=first.dtype)
    for tiles, rec in zip(per_rank_tiles, per_rank_records):
        idx, _ = rays_of_tiles(width, height, tile, tiles)
        assert len(idx) == len(rec)
        fb[idx] = rec
    return fb
