"""CPU suite: the N > 1 path (screen-tile split + ONE all-gather) with world_size 2 and 3 over gloo.

Every rank renders ONLY the rows it owns -- with the oracle, since the HIP kernels need a GPU; the
oracle takes the same ownership description the device gets (output-row band, or interleaved tile
rows) -- lays them out the way the device lays out its output buffer, all-gathers, and (stripes)
un-permutes.  What comes out on every rank must be the frame the reference rendered.  The band /
stripe arithmetic and the un-permute are the ones bench.py and BandRenderer use on the GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import scenes
from conftest import load_golden
from py_numpy_renderer_amd.multigpu import (all_gather_frame, row_band, stripe_rows, tile_row_costs, unband_index, unstripe,
                                             unstripe_index, weighted_bands)


def test_row_band_partition():
    assert [row_band(1080, r, 8) for r in (0, 7)] == [(0, 135), (945, 1080)]
    covered = sorted(sum(([*range(*row_band(240, r, 4))] for r in range(4)), []))
    assert covered == list(range(240))
    with pytest.raises(ValueError):
        row_band(1080, 0, 7)
    with pytest.raises(ValueError):
        row_band(10, 3, 2)


def test_weighted_bands_cut_on_tile_rows_and_balance_the_cost():
    rng = np.random.default_rng(7)
    for height in (1080, 2160, 270, 64, 17):
        n = -(-height // 16)
        top_rows = height - (n - 1) * 16
        for world in (1, 2, 3, 4, 8):
            if world > n:
                with pytest.raises(ValueError):
                    weighted_bands(np.ones(n), height, world)
                continue
            cost = rng.integers(1, 1000, n)
            bands = weighted_bands(cost, height, world)
            assert bands[0][0] == 0 and bands[-1][1] == height
            assert all(b[1] == c[0] for b, c in zip(bands, bands[1:])) and all(e > b for b, e in bands)
            assert all(b == 0 or (b - top_rows) % 16 == 0 for b, _ in bands)          # cuts sit on tile rows
            tile_of = lambda row: 0 if row == 0 else (row - top_rows) // 16 + 1       # from the top
            top = cost[::-1]
            worst = max(int(top[tile_of(b):(n if e == height else tile_of(e))].sum()) for b, e in bands)
            if world == 2:                                                            # against brute force
                assert worst == min(max(top[:k].sum(), top[k:].sum()) for k in range(1, n))
            assert worst <= max(int(top.sum()) // world + int(top.max()), int(top.max()))
            idx = unband_index(bands)
            assert len(idx) == height and len(set(idx.tolist())) == height
    # a frame whose bottom tile rows hold all the work: the top band takes everything else
    assert weighted_bands([100] * 10 + [1] * 58, 1080, 4) == [(0, 952), (952, 1000), (1000, 1048), (1048, 1080)]
    with pytest.raises(ValueError):
        weighted_bands(np.ones(5), 1080, 2)
    # with the rows' most expensive tiles: a band's cost is its sum plus a tail behind its heaviest tile (a fifth of that
    # tile's cost times the tiles the device holds at once), so the band under one very heavy tile gets fewer rows
    flat, spike = [100] * 68, [10] * 68
    spike[30] = 400
    even = weighted_bands(flat, 1080, 4, row_peak=[10] * 68, resident=100)
    tilted = weighted_bands(flat, 1080, 4, row_peak=spike, resident=100)
    rows = lambda bands: [e - b for b, e in bands]
    assert max(rows(even)) - min(rows(even)) <= 16 + 8                 # (the top tile row of 1080 is half a row)
    heavy = [i for i, (b, e) in enumerate(tilted) if b <= 1080 - 16 * 31 < e][0]      # the band that holds tile row 30 (from the bottom)
    assert rows(tilted)[heavy] < min(r for i, r in enumerate(rows(tilted)) if i != heavy)
    with pytest.raises(ValueError):
        weighted_bands(flat, 1080, 4, row_peak=[1] * 5)
    rec = np.zeros((6, 12), np.uint32)
    rec[:, 5], rec[:, 6], rec[:, 7] = [1, 0, 0, 0, 2, 0], [0, 1, 0, 0, 0, 0], [0, 0, 5, 0, 0, 0]
    assert tile_row_costs(rec, 3).tolist() == [20 + 2 + 20 + 30 + 20 + 15, 20 + 20 + 4 + 20]
    sums, peaks = tile_row_costs(rec, 3, peaks=True)
    assert sums.tolist() == [107, 64] and peaks.tolist() == [50, 24]


def stripe_pack(frame, rank, world):
    """A rank's output buffer in the striped layout (include/mi355rast.h, stripe_count), cut from a
    frame whose rows the rank owns are valid: tile rows g = rank, rank + world, ... counted from the
    bottom, highest first, rows inside a tile row top-down; unused rows stay zero."""
    height = frame.shape[0]
    per = stripe_rows(height, world)
    part = np.zeros((per,) + frame.shape[1:], frame.dtype)
    for py in range(height):
        g = py // 16
        if g % world != rank:
            continue
        local = g // world
        part[(per // 16 - 1 - local) * 16 + (16 * g + 15 - py)] = frame[height - 1 - py]
    return part


@pytest.mark.parametrize("height,world", [(1080, 8), (1080, 3), (270, 2), (17, 4), (16, 3), (1, 2)])
def test_unstripe_inverts_the_striped_layout(height, world):
    frame = np.arange(height * 5 * 3, dtype=np.int64).reshape(height, 5, 3) % 251
    frame = frame.astype(np.uint8)
    gathered = np.concatenate([stripe_pack(frame, r, world) for r in range(world)], axis=0)
    assert gathered.shape[0] == world * stripe_rows(height, world)
    idx = unstripe_index(height, world)
    assert sorted(set(idx.tolist())) == sorted(idx.tolist())          # every output row comes from its own source row
    assert np.array_equal(unstripe(torch.from_numpy(gathered), height, world).numpy(), frame)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, name, partition, result_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle
        from py_numpy_renderer_amd._pack import pack_scene
        scene = scenes.build(scenes.product_api(), name)
        height = int(scene.resolution[0])
        packed = pack_scene(scene, shadows=True)
        bands = None
        if partition == "weighted":
            # bands of equal cost: here the cost of a tile row is made up (the device takes it from its tile records),
            # all that matters is that every rank arrives at the same unequal cuts
            bands = weighted_bands(np.arange(-(-height // 16)) ** 2 + 1, height, world)
            assert len({e - b for b, e in bands}) > 1, "the test wants unequal bands"
        if partition in ("bands", "weighted"):
            band = bands[rank] if bands else row_band(height, rank, world)
            mine = oracle.render_packed(packed, own_rows=band, want_status=False, want_silhouette=False)
            part = np.ascontiguousarray(mine.out[band[0]:band[1]])
            if bands:                                # every rank sends the tallest band's rows, its own first
                pad = np.zeros((max(e - b for b, e in bands),) + part.shape[1:], part.dtype)
                pad[:part.shape[0]] = part
                part = pad
        else:
            mine = oracle.render_packed(packed, own_stripe=(rank, world), want_status=False, want_silhouette=False)
            part = stripe_pack(mine.out, rank, world)
        # rows the rank does not own were not rendered: they still hold the background
        other = np.ones(height, bool)
        if partition in ("bands", "weighted"):
            other[band[0]:band[1]] = False
        else:
            other[[height - 1 - py for py in range(height) if (py // 16) % world == rank]] = False
        assert (mine.winner[::-1][other] == -1).all(), "a rank rendered rows it does not own"
        gathered = all_gather_frame(torch.from_numpy(part))
        frame = (gathered if partition == "bands" else
                 gathered.index_select(0, unband_index(bands)) if bands else unstripe(gathered, height, world))
        np.save(os.path.join(result_dir, f"rank{rank}.npy"), frame.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("partition,world", [("bands", 2), ("bands", 3), ("stripes", 2), ("stripes", 3), ("weighted", 2),
                                             ("weighted", 3)])
def test_ranks_render_their_tiles_and_gather_the_golden_frame(api, oracle_mod, tmp_path, partition, world):
    name = "diablo_floor_small"                      # 270 rows: splits evenly in 2 and 3; 17 tile rows
    g, _ = load_golden(name)
    mp.spawn(_worker, args=(world, _free_port(), name, partition, str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        frame = np.load(tmp_path / f"rank{rank}.npy")
        assert frame.shape == g["out"].shape
        assert np.abs(frame.astype(int) - g["out"].astype(int)).max() <= 1, f"rank {rank}"
    assert np.array_equal(np.load(tmp_path / "rank0.npy"), np.load(tmp_path / f"rank{world - 1}.npy"))


def _overlay_worker(rank, world, port, name, partition, result_dir):
    """One rank of a split frame with upstream's default overlay, on the CPU: the oracle renders the rank's rows, the
    rank appends the state (z, float colour) of the touched pixels it owns, ONE all-gather carries rows and state,
    and the overlay is replayed on the assembled frame from the gathered state (what mr_overlay_apply does, restated
    in NumPy: frustums.replay_bids on slots of the list of touched pixels)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from types import SimpleNamespace
        from oracle import oracle
        from py_numpy_renderer_amd._pack import pack_scene
        from py_numpy_renderer_amd.frustums import OverlayOps, replay_bids
        scene = scenes.build(scenes.product_api(), name)
        height, width = (int(v) for v in scene.resolution)
        packed = pack_scene(scene, shadows=True)
        if partition == "bands":
            band = row_band(height, rank, world)
            mine = oracle.render_packed(packed, own_rows=band, want_status=False, want_silhouette=False)
            rows = np.ascontiguousarray(mine.out[band[0]:band[1]])
            owner = lambda py: (height - 1 - py) // (height // world)
        else:
            mine = oracle.render_packed(packed, own_stripe=(rank, world), want_status=False, want_silhouette=False)
            rows = stripe_pack(mine.out, rank, world)
            owner = lambda py: (py // 16) % world
        ops = OverlayOps(scene.camera, scene.debug_camera, scene.resolution, native=False)
        touched, inverse = np.unique(ops.target, return_inverse=True)
        state = np.zeros(len(touched), dtype=[("z", "<f8"), ("f", "<f4", 3), ("pad", "<u4")])
        own = owner(touched // width) == rank
        state["z"][own] = mine.z.reshape(-1)[touched[own]]
        state["f"][own] = mine.frame.reshape(-1, 3)[touched[own]]
        offset = -(-rows.size // 16) * 16
        part = np.zeros(offset + state.nbytes, np.uint8)
        part[:rows.size] = rows.reshape(-1)
        part[offset:] = state.view(np.uint8)
        gathered = all_gather_frame(torch.from_numpy(part)).numpy().reshape(world, -1)
        parts = torch.from_numpy(np.ascontiguousarray(gathered[:, :rows.size]).reshape((world * rows.shape[0],) + rows.shape[1:]))
        frame = (parts if partition == "bands" else unstripe(parts, height, world)).numpy().copy()
        states = np.ascontiguousarray(gathered[:, offset:]).view(state.dtype).reshape(world, len(touched))
        mine_of = states[owner(touched // width), np.arange(len(touched))]
        st_z, st_f = mine_of["z"].copy(), mine_of["f"].copy()
        slots = SimpleNamespace(seg_first=ops.seg_first, seg_count=ops.seg_count, z=ops.z, target=inverse.reshape(ops.target.shape))
        replay_bids(slots, st_f, st_z, int(scene.system))
        frame[height - 1 - touched // width, touched % width] = (st_f ** np.float32(0.8) * 255).astype(np.uint8)
        np.save(os.path.join(result_dir, f"rank{rank}.npy"), frame)
        np.save(os.path.join(result_dir, f"z{rank}.npy"), np.stack([touched.astype(np.float64), st_z]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("partition,world", [("bands", 2), ("stripes", 3)])
def test_split_frame_with_overlay_gathers_the_golden_overlay_frame(api, oracle_mod, tmp_path, partition, world):
    """The overlay of a split frame (one all-gather of rows + touched-pixel state, replay on the assembled frame)
    against the reference's capture with the overlay left on: frame +-1, post-overlay z at the touched pixels bit-exact."""
    name = "diablo_small_overlay"                    # 240 rows
    g, _ = load_golden(name)
    mp.spawn(_overlay_worker, args=(world, _free_port(), name[:-len("_overlay")], partition, str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        frame = np.load(tmp_path / f"rank{rank}.npy")
        assert np.abs(frame.astype(int) - g["out"].astype(int)).max() <= 1, f"rank {rank}"
        touched, z = np.load(tmp_path / f"z{rank}.npy")
        assert np.array_equal(z.view(np.uint64), g["z_overlay"].reshape(-1)[touched.astype(np.int64)].view(np.uint64)), f"rank {rank}"
    assert np.array_equal(np.load(tmp_path / "rank0.npy"), np.load(tmp_path / f"rank{world - 1}.npy"))
