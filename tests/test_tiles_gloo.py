"""Multi-process shape of the multi-GPU path on CPU (gloo, world_size 2): every rank takes the
row strips its tile policy assigns (flx_tile_row_count / flx_tile_row_at, the same C entry points
bench.py uses), the strips are all-gathered with padding and scattered back to image rows exactly
as bench.py does after the RCCL all-gather.  The per-rank "renderer" here is the CPU oracle (test
infrastructure), which honours the same tile policy; the point is the partition + gather + reassembly."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tile_rows, w, h, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import flx_oracle
    from flexlight_hip import capi
    from flexlight_hip.scene_io import Scene
    sc = Scene.golden("cornell")
    p = sc.frame_params(width=w, height=h, samples=1, max_reflections=2, use_filter=0, tile=(tile_rows, rank, world))
    rows = capi.Context.tile_rows(p)
    part, _, _ = flx_oracle.render(sc, p, threads=2)
    assert part.shape[0] == len(rows)
    strips = (h + tile_rows - 1) // tile_rows
    rows_max = ((strips + world - 1) // world) * tile_rows
    local = torch.zeros((rows_max, w, 4))
    local[:len(rows)] = torch.from_numpy(part)
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    frame = torch.full((h, w, 4), float("nan"))
    for r in range(world):
        pr = sc.frame_params(width=w, height=h, tile=(tile_rows, r, world))
        rr = capi.Context.tile_rows(pr)
        frame[rr] = gathered[r][:len(rr)]
    if rank == 0:
        np.save(os.path.join(out_dir, "frame.npy"), frame.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("tile_rows,h", [(8, 64), (8, 52), (5, 37)])
def test_two_ranks_reassemble_the_frame(tmp_path, tile_rows, h):
    sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import flx_oracle
    from flexlight_hip.scene_io import Scene
    flx_oracle.build()
    w = 48
    port = 29500 + (os.getpid() % 2000) + tile_rows
    mp.spawn(_worker, args=(2, port, tile_rows, w, h, str(tmp_path)), nprocs=2, join=True)
    frame = np.load(tmp_path / "frame.npy")
    sc = Scene.golden("cornell")
    full, _, _ = flx_oracle.render(sc, sc.frame_params(width=w, height=h, samples=1, max_reflections=2, use_filter=0), threads=2)
    assert not np.isnan(frame[..., 3]).any()
    assert np.array_equal(frame, full, equal_nan=True)
