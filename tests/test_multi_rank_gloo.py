"""The N > 1 path on CPU: two ranks (gloo) each take their interleaved tiles, the tile
buffers are gathered to rank 0 and scattered back into the canvas.  The GPU render of a
rank's tiles is stood in for by the oracle rendering exactly the pixels of that rank's tile
buffer, so what is tested is the partition / gather / de-interleave logic the GPU path uses
(raytracer_challenge_amd.tiles mirrors the device kernels' index math)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from raytracer_challenge_amd import native, scenes, tiles


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world_size, port, hsize, vsize, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        world, cam = scenes.draw_world_scene(hsize, vsize)
        cd = cam.desc()
        idx = tiles.local_pixel_indices(hsize, vsize, rank, world_size)
        assert idx.size * 3 == native.tile_buffer_doubles(cd, rank, world_size)
        n_rank0 = native.tile_buffer_doubles(cd, 0, world_size)
        local = torch.zeros(n_rank0, dtype=torch.float64)
        valid = idx >= 0
        px = O.render_pixels(world, cam, idx[valid].astype(np.uint64), 5, O.STREAMING)
        buf = np.zeros((idx.size, 3))
        buf[valid] = px
        local[: idx.size * 3] = torch.from_numpy(buf.reshape(-1))
        gathered = [torch.zeros(n_rank0, dtype=torch.float64) for _ in range(world_size)] if rank == 0 else None
        dist.gather(local, gathered, dst=0)
        # the 8-bit variant: each rank quantises its tile buffer (ppm.zig:6-9) before the gather
        local_u8 = torch.from_numpy(O.quantize_u8(local.numpy()))
        gathered_u8 = [torch.zeros(n_rank0, dtype=torch.uint8) for _ in range(world_size)] if rank == 0 else None
        dist.gather(local_u8, gathered_u8, dst=0)
        if rank == 0:
            g = torch.stack(gathered).numpy().reshape(world_size, -1, 3)
            canvas = tiles.deinterleave(g, hsize, vsize, world_size)
            full = O.render(world, cam, 5, O.STREAMING)
            g8 = torch.stack(gathered_u8).numpy().reshape(world_size, -1, 3)
            canvas_u8 = tiles.deinterleave(g8, hsize, vsize, world_size)
            ok_u8 = canvas_u8.dtype == np.uint8 and np.array_equal(canvas_u8, O.quantize_u8(full.reshape(-1)).reshape(full.shape))
            q.put(bool(np.array_equal(canvas, full)) and bool(ok_u8))
    finally:
        dist.destroy_process_group()


def test_two_ranks_reassemble_the_canvas():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    hsize, vsize = 150, 70  # edge tiles in both directions, odd tile count
    procs = [ctx.Process(target=_worker, args=(r, 2, port, hsize, vsize, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
