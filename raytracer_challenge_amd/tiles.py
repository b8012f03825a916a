"""Host-side index math of the interleaved tile partition used for multi-GPU rendering.

The canvas is cut into tiles of RTC_TILE_W x RTC_TILE_H pixels numbered row-major; rank r of
R renders the tiles t with t % R == r into a dense buffer in which local tile k (global tile
k*R + r) occupies the k-th block of RTC_TILE_W*RTC_TILE_H pixels, row-major inside the tile.
Edge tiles keep their full size in the buffer; the pixels that fall outside the canvas are
padding.  The device kernels (render_kernel, deinterleave_kernel) use the same mapping.
"""
import numpy as np

from .descs import RTC_TILE_H, RTC_TILE_W


def tile_grid(hsize, vsize):
    tiles_x = (hsize + RTC_TILE_W - 1) // RTC_TILE_W
    tiles_y = (vsize + RTC_TILE_H - 1) // RTC_TILE_H
    return tiles_x, tiles_y


def local_tile_count(hsize, vsize, rank, ranks):
    tx, ty = tile_grid(hsize, vsize)
    total = tx * ty
    return (total - rank + ranks - 1) // ranks if total > rank else 0


def local_pixel_indices(hsize, vsize, rank, ranks):
    """Canvas index (y*hsize + x) of every slot of rank's tile buffer, -1 for padding."""
    tx, _ = tile_grid(hsize, vsize)
    n_local = local_tile_count(hsize, vsize, rank, ranks)
    k = np.arange(n_local, dtype=np.int64)
    gt = k * ranks + rank
    within = np.arange(RTC_TILE_W * RTC_TILE_H, dtype=np.int64)
    px = (gt % tx)[:, None] * RTC_TILE_W + (within % RTC_TILE_W)[None, :]
    py = (gt // tx)[:, None] * RTC_TILE_H + (within // RTC_TILE_W)[None, :]
    idx = py * hsize + px
    idx[(px >= hsize) | (py >= vsize)] = -1
    return idx.reshape(-1)


def deinterleave(gathered, hsize, vsize, ranks):
    """numpy counterpart of rtc_deinterleave_device: gathered is (ranks, per_rank_pixels, 3)."""
    out = np.zeros((vsize * hsize, 3), dtype=gathered.dtype)
    for r in range(ranks):
        idx = local_pixel_indices(hsize, vsize, r, ranks)
        valid = idx >= 0
        out[idx[valid]] = gathered[r, : idx.size][valid]
    return out.reshape(vsize, hsize, 3)
