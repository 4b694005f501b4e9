"""CPU suite: the N > 1 path (row-band split + ONE all-gather) with world_size 2 over gloo.

Each rank produces its band of the frame (here cut from the oracle's frame, since the HIP
kernels need a GPU) and ``all_gather_frame`` must assemble exactly the golden frame on every
rank.  The band arithmetic is the one bench.py and BandRenderer use on the GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import scenes
from conftest import load_golden
from py_numpy_renderer_amd.multigpu import all_gather_frame, row_band


def test_row_band_partition():
    assert [row_band(1080, r, 8) for r in (0, 7)] == [(0, 135), (945, 1080)]
    covered = sorted(sum(([*range(*row_band(240, r, 4))] for r in range(4)), []))
    assert covered == list(range(240))
    with pytest.raises(ValueError):
        row_band(1080, 0, 7)
    with pytest.raises(ValueError):
        row_band(10, 3, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, frame_path, result_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = np.load(frame_path)
        lo, hi = row_band(full.shape[0], rank, world)
        part = torch.from_numpy(np.ascontiguousarray(full[lo:hi]))
        frame = all_gather_frame(part)
        np.save(os.path.join(result_dir, f"rank{rank}.npy"), frame.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_bands_all_gather_into_the_golden_frame(api, oracle_mod, tmp_path, world):
    name = "diablo_floor_small"                      # 270 rows: splits evenly in 2 and 3
    g, _ = load_golden(name)
    frame = oracle_mod.render(scenes.build(api, name)).out
    assert np.abs(frame.astype(int) - g["out"].astype(int)).max() <= 1
    path = str(tmp_path / "frame.npy")
    np.save(path, frame)
    mp.spawn(_worker, args=(world, _free_port(), path, str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        assert np.array_equal(np.load(tmp_path / f"rank{rank}.npy"), frame)
