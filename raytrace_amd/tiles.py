"""Image-tile sharding for multi-GPU renders: which rays a rank owns and how gathered per-rank
results go back into the row-major framebuffer.  Tiles are dealt round-robin (tile k -> rank
k mod world), so neighbouring tiles -- which cost about the same -- land on different ranks and the
silhouette's expensive tiles spread evenly without any cross-process queue."""
from __future__ import annotations

import numpy as np


def tile_indices(width, height, tile, rank, world):
    """Ray indices (row-major i + j*width) of the tiles owned by `rank`, tile by tile."""
    tx, ty = (width + tile - 1) // tile, (height + tile - 1) // tile
    parts = []
    for k in range(rank, tx * ty, world):
        x0, y0 = (k % tx) * tile, (k // tx) * tile
        xs = np.arange(x0, min(x0 + tile, width))
        ys = np.arange(y0, min(y0 + tile, height))
        parts.append((ys[:, None] * width + xs[None, :]).reshape(-1))
    return np.concatenate(parts) if parts else np.zeros(0, np.int64)


def scatter_to_framebuffer(width, height, tile, world, per_rank):
    """Inverse of the partition: per_rank[r] holds rank r's records in tile_indices order."""
    first = per_rank[0]
    fb = np.zeros((width * height,) + tuple(first.shape[1:]), dtype=first.dtype)
    for r in range(world):
        idx = tile_indices(width, height, tile, r, world)
        assert len(idx) == len(per_rank[r])
        fb[idx] = per_rank[r]
    return fb
