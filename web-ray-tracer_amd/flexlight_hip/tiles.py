"""Reassembly of tile-sharded frames after the all-gather: which slot of the gathered buffers holds image
row y of frame i.  Used by bench.py (RCCL) and tests/test_tiles_gloo.py (gloo)."""
import numpy as np


def padded_rows(height, tile_rows, world):
    """rows of the rank that gets the most strips: every rank's gather buffer is padded to this"""
    strips = (height + tile_rows - 1) // tile_rows
    return ((strips + world - 1) // world) * tile_rows


def gather_index(rows_of, frames, rows_max, height):
    """rows_of[r] = image rows of rank r's packed rows (flx_tile_row_at).  Rank r packs a batch tight — float4[frames][len(rows_of[r])][W]
    — at the start of its frames * rows_max rows of the gathered buffer.  Returns perm (int64 [frames * height]):
    row y of frame i is row perm[i * height + y] of gathered.view(world * frames * rows_max, W, 4)."""
    perm = np.full(frames * height, -1, np.int64)
    for r, rows in enumerate(rows_of):
        rows = np.asarray(rows, np.int64)
        for i in range(frames):
            perm[i * height + rows] = r * frames * rows_max + i * len(rows) + np.arange(len(rows))
    if (perm < 0).any():
        raise ValueError("the ranks' strips do not cover the frame")
    return perm
