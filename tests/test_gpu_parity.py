"""GPU parity: the HIP path (through the C ABI) against the committed reference captures and
against the C oracle on the same scenes.

Bars (BASELINE.json north_star): z-buffer, winner map and stencil bit-exact; uint8 frame
within +-1 per channel.  The float frame is additionally held to 2e-6 absolute (values are in
[0.05, 1]; only libm-vs-ocml pow/sqrt rounding separates the two paths).
"""
import numpy as np
import pytest

import scenes
from conftest import load_golden

pytestmark = pytest.mark.gpu

SMALL = list(scenes.SMALL) + ["diablo_small_noshadow"]


def _render(api, name):
    scene = scenes.build(api, name)
    shadows = name not in scenes.NO_SHADOW
    backend = scene._backend()
    out = backend.render(scene, shadows=shadows, keep_float=True)
    return scene, backend, out, shadows


def _assert_buffers(backend, out, want_z, want_winner, want_stencil, want_frame, want_out, label):
    z = backend.read_z()
    bad_z = int((z.view(np.uint64) != want_z.view(np.uint64)).sum())
    assert bad_z == 0, f"{label}: {bad_z} z-buffer entries not bit-exact"
    winner = backend.read_winner()
    assert int((winner != want_winner).sum()) == 0, f"{label}: winner map differs"
    stencil = backend.read_stencil()
    assert int((stencil != want_stencil).sum()) == 0, f"{label}: stencil differs"
    if want_frame is not None:
        frame = backend.read_frame_f32()
        err = np.abs(frame.astype(np.float64) - want_frame.astype(np.float64))
        assert err.max() <= 2e-6, f"{label}: float frame off by {err.max():.3g} at {np.argwhere(err == err.max())[0]}"
    d = np.abs(out.astype(np.int16) - want_out.astype(np.int16))
    assert d.max() <= 1, f"{label}: uint8 frame off by {d.max()} ({int((d > 1).sum())} values > 1)"


@pytest.mark.parametrize("name", SMALL)
def test_small_scene_matches_reference_capture(api, name):
    """Committed golden vectors (captured from the reference itself)."""
    g, meta = load_golden(name)
    scene, backend, out, _ = _render(api, name)
    _assert_buffers(backend, out, g["z"], g["winner"], g["stencil"], g["frame"], g["out"], f"{name} vs reference")
    st = backend.last_stats
    assert st["frag_tri"] == meta["counts"]["frag_tri_pass1"]
    assert st["frag_quad"] == meta["counts"]["frag_quad"]
    assert st["n_quads"] == meta["counts"]["n_quads"]
    sil = set(map(tuple, backend.read_silhouette().tolist()))
    assert sil == set(map(tuple, g["silhouette"].tolist()))
    scene.close()


@pytest.mark.parametrize("name", SMALL)
def test_small_scene_matches_oracle(api, oracle_mod, name):
    """Same scenes against the C oracle run on this host (checks the oracle travels intact)."""
    scene, backend, out, shadows = _render(api, name)
    want = oracle_mod.render(scene, shadows=shadows)
    _assert_buffers(backend, out, want.z, want.winner, want.stencil, want.frame, want.out, f"{name} vs oracle")
    scene.close()


FRAME_ONLY = SMALL + ["c3_diablo_floor_1080p", "c4_torus200k_1080p"]


@pytest.mark.parametrize("name", FRAME_ONLY)
def test_frame_only_mode_renders_the_same_frame(api, name):
    """Without MR_FRAME_COUNTERS (how Scene.render() and bench.py render) the library skips
    shadow quads that cannot pass the depth test anywhere in a strip of pixels.  The frame,
    the z-buffer, the winners and -- at every pixel a triangle covers -- the stencil must be
    exactly those of the counted frame; the counters are reported as "not counted" (-1)."""
    scene = scenes.build(api, name)
    shadows = name not in scenes.NO_SHADOW
    backend = scene._backend()
    exact = backend.render(scene, shadows=shadows, keep_float=True, counters=True)
    z, winner, stencil, frame = backend.read_z(), backend.read_winner(), backend.read_stencil(), backend.read_frame_f32()
    assert backend.last_stats["frag_tri"] > 0
    fast = backend.render(scene, shadows=shadows, keep_float=True, counters=False, keep_buffers=True)
    assert np.array_equal(fast, exact)
    assert np.array_equal(backend.read_z().view(np.uint64), z.view(np.uint64))
    assert np.array_equal(backend.read_winner(), winner)
    assert np.array_equal(backend.read_frame_f32().view(np.uint32), frame.view(np.uint32))
    covered = winner >= 0
    assert np.array_equal(backend.read_stencil()[covered], stencil[covered])
    assert backend.last_stats["frag_tri"] == -1 and backend.last_stats["frag_quad"] == -1
    assert backend.last_stats["n_quads"] >= 0
    assert np.array_equal(scene.render(shadows=shadows), exact)
    scene.close()


def test_render_is_repeatable_and_matches_scene_render(api):
    """Scene.render() (the drop-in call) returns the same frame on every call -- unlike the
    reference, whose silhouette set toggles between calls (obj/core.py:251,605)."""
    scene = scenes.build(api, "diablo_small")
    a = scene.render()
    b = scene.render()
    assert a.dtype == np.uint8 and a.shape == (240, 320, 3)
    assert np.array_equal(a, b)
    g, _ = load_golden("diablo_small")
    assert np.abs(a.astype(np.int16) - g["out"].astype(np.int16)).max() <= 1
    scene.close()


@pytest.mark.parametrize("name,shadows", [("c3_diablo_floor_1080p", True), ("diablo_small", True), ("kat_house", False)])
def test_heaviest_first_tile_order_is_a_permutation(api, name, shadows):
    """The tile kernel takes its tiles heaviest first, by what the previous frame on the slot learnt: the
    first frame's order is row-major, later ones are permutations sorted by cost class, and the frame does
    not depend on the order (kernels_tile.h tile_class, kernels_bin.h order_tiles_block)."""
    scene = scenes.build(api, name)
    backend = scene._backend()
    first = backend.render(scene, shadows=shadows).copy()
    n = len(backend.read_tile_order())       # (row-major, unless growing a work list made this a second attempt)
    rec = backend.read_tile_records().astype(np.int64)
    cost = 20 + 2 * rec[:, 5] + 30 * rec[:, 6] + 3 * rec[:, 7]                           # kernels_tile.h tile_cost
    cls = np.digitize(-cost, [-500, -350, -250, -170, -110, -70, -40], right=True)        # 0 = heaviest
    # a frame rendered in order shares out the tiles of its first class: "quads worth sharing" (kernels_tile.h
    # HEAVY_SPLIT), with a lower bar on a device that owns few tiles
    bar = (350, 32) if n <= 2048 else (400, 48)
    cls = np.where((cost >= bar[0]) & (rec[:, 7] >= bar[1]), 0, np.maximum(cls, 1))
    for _ in range(2):
        again = backend.render(scene, shadows=shadows)
        order = backend.read_tile_order().astype(np.int64)
        assert np.array_equal(np.sort(order), np.arange(n))
        assert (np.diff(cls[order]) >= 0).all()
        assert np.array_equal(again, first)
        assert np.array_equal(backend.read_tile_records().astype(np.int64)[:, 5:8], rec[:, 5:8])
    # a new tile grid starts in row-major order again
    half = backend.render(scene, shadows=shadows, row_band=(0, first.shape[0] // 32 * 16))
    assert np.array_equal(backend.read_tile_order(), np.arange(len(backend.read_tile_order()), dtype=np.uint32))
    assert np.array_equal(half, first[:half.shape[0]])
    assert np.array_equal(backend.render(scene, shadows=shadows, row_band=(0, first.shape[0] // 32 * 16)), half)
    scene.close()


def test_frames_from_several_streams_keep_row_major_order(api):
    """The tile order is for a frame that has the device to itself; a scene rendered from several streams at
    once (frames in flight) keeps row-major order (mi355rast.hip, enqueue_frame) -- and the same frames."""
    from py_numpy_renderer_amd.multigpu import BandRenderer
    scene = scenes.build(api, "diablo_floor_small")
    want = scene.render()
    br = BandRenderer(scene, frames_in_flight=3, timing_every=0)
    for _ in range(9):
        frame = br.step()
    assert br.verify()
    order = scene._backend().read_tile_order()
    assert np.array_equal(order, np.arange(len(order), dtype=np.uint32))
    assert np.array_equal(frame.cpu().numpy(), want)
    scene.close()


@pytest.mark.parametrize("bands", [2, 3, 8])
def test_row_bands_tile_the_frame(api, bands):
    """Screen-tile split: rendering disjoint row bands and stacking them gives the whole frame
    (what the multi-GPU path all-gathers)."""
    scene = scenes.build(api, "diablo_floor_small")
    full = scene.render()
    h = full.shape[0]
    edges = [round(i * h / bands) for i in range(bands + 1)]
    parts = [scene.render(row_band=(edges[i], edges[i + 1])) for i in range(bands)]
    assert np.array_equal(np.concatenate(parts, axis=0), full)
    scene.close()


@pytest.mark.parametrize("name,world", [("diablo_floor_small", 2), ("diablo_floor_small", 3), ("diablo_floor_small", 8),
                                        ("tetra_ortho", 5), ("c4_torus200k_1080p", 8), ("c3_diablo_floor_1080p", 4)])
def test_tile_row_stripes_tile_the_frame(api, name, world):
    """Screen-tile split, interleaved: rendering every rank's tile rows (t mod world == rank) into the
    striped layout, concatenating them the way the all-gather does and un-permuting gives the whole
    frame (what the multi-GPU path does with partition="stripes")."""
    import torch
    from py_numpy_renderer_amd.multigpu import stripe_rows, unstripe
    scene = scenes.build(api, name)
    backend = scene._backend()
    full = scene.render()
    h = full.shape[0]
    parts = [backend.render(scene, counters=False, stripe=(r, world)) for r in range(world)]
    assert all(p.shape[0] == stripe_rows(h, world) for p in parts)
    frame = unstripe(torch.from_numpy(np.concatenate(parts, axis=0)), h, world).numpy()
    assert np.array_equal(frame, full)
    scene.close()


@pytest.mark.parametrize("name,world", [("diablo_floor_small", 3), ("c4_torus200k_1080p", 4), ("c4_torus200k_1080p", 8),
                                        ("c3_diablo_floor_1080p", 2)])
def test_cost_weighted_bands_tile_the_frame(api, name, world):
    """Screen-tile split, bands of equal COST (partition="weighted"): the cuts come from the tile records of a whole
    frame, every rank's band is rendered into a buffer as tall as the tallest band, and the row gather after the
    all-gather gives the whole frame; the cuts even the modelled cost out better than equal bands do."""
    import torch
    from py_numpy_renderer_amd.multigpu import row_band, tile_row_costs, unband_index, weighted_bands
    scene = scenes.build(api, name)
    backend = scene._backend()
    full = scene.render()
    h, w = full.shape[:2]
    costs = tile_row_costs(backend.read_tile_records(), -(-w // 16))
    assert len(costs) == -(-h // 16)
    bands = weighted_bands(costs, h, world)
    per = max(e - b for b, e in bands)
    parts = []
    for b, e in bands:
        part = np.zeros((per, w, 3), np.uint8)
        part[:e - b] = backend.render(scene, counters=False, row_band=(b, e))
        parts.append(part)
    frame = torch.from_numpy(np.concatenate(parts, axis=0)).index_select(0, unband_index(bands)).numpy()
    assert np.array_equal(frame, full)
    top = costs[::-1]
    top_rows = h - (len(costs) - 1) * 16
    tile_of = lambda row: 0 if row == 0 else (row - top_rows) // 16 + 1
    worst = lambda bb: max(int(top[tile_of(b):(len(top) if e == h else tile_of(e))].sum()) for b, e in bb)
    if h % world == 0 and (h // world) % 16 == 0:
        assert worst(bands) <= worst([row_band(h, r, world) for r in range(world)])
    scene.close()


@pytest.mark.parametrize("name,world,rank", [("c4_torus200k_1080p", 8, 3), ("c3_diablo_floor_1080p", 8, 2)])
def test_split_heavy_tiles_render_the_same_rows(api, name, world, rank):
    """A device that owns few tiles (a rank of a multi-GPU split) shares the shadow quads of its heaviest tiles
    out over four workgroups once the previous frame has told it which they are (kernels_tile.h HEAVY_SPLIT):
    the first frame of a tile grid (row-major, one workgroup per tile) and the later ones (ordered, split)
    must be the same rows, and some tile must actually qualify for the split."""
    scene = scenes.build(api, name)
    backend = scene._backend()
    first = backend.render(scene, counters=False, stripe=(rank, world)).copy()
    rec = backend.read_tile_records().astype(np.int64)
    assert len(rec) <= 2048
    cost = 20 + 2 * rec[:, 5] + 30 * rec[:, 6] + 3 * rec[:, 7]          # kernels_tile.h tile_cost
    heavy = (cost >= 350) & (rec[:, 7] >= 32)
    assert heavy.any(), "no tile qualifies: the test would not exercise the split"
    for _ in range(3):
        again = backend.render(scene, counters=False, stripe=(rank, world))
        order = backend.read_tile_order().astype(np.int64)
        assert np.array_equal(np.sort(order), np.arange(len(rec)))
        assert heavy[order[:int(heavy.sum())]].all()              # the qualifying tiles lead the order
        assert np.array_equal(again, first)
    scene.close()


def test_work_lists_grow_on_overflow(api):
    """Per-tile lists start at a fixed capacity and grow when a tile overflows: with capacities forced
    far too small, mr_render (Scene.render) retries by itself and BandRenderer (frames enqueued without
    host synchronisation) notices while priming; both end up with the frame of an unconstrained render."""
    import torch
    from py_numpy_renderer_amd.multigpu import BandRenderer
    scene = scenes.build(api, "diablo_floor_small")
    want = scene.render()
    scene.close()

    scene = scenes.build(api, "diablo_floor_small")
    scene._backend().set_list_capacities(small_pairs=4, big_pairs=2, quads=3, work=16)
    assert np.array_equal(scene.render(), want)
    scene.close()

    scene = scenes.build(api, "diablo_floor_small")
    scene._backend().set_list_capacities(small_pairs=4, big_pairs=2, quads=3, work=16)
    br = BandRenderer(scene, 0, 1, shadows=True, light_timing=True, frames_in_flight=2)
    frames = [br.step() for _ in range(4)]
    assert br.verify()
    torch.cuda.synchronize()
    for frame in frames[-2:]:
        assert np.array_equal(frame.cpu().numpy(), want)
    scene.close()


def test_overflow_in_an_earlier_frame_is_not_forgotten(api):
    """Frames enqueued without host synchronisation: a work list that overflows in ONE frame must still be
    reported after later frames of the same stream that overflow nothing (the frame counters are
    double-buffered and recycled; the verdict has to survive that), and once the lists have been grown and
    the host has acted, a clean frame must read as clean again."""
    import torch
    scene = scenes.build(api, "diablo_floor_small")
    want = scene.render()
    backend = scene._backend()
    h, w = (int(v) for v in scene.resolution)
    out = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.Stream()
    looking_at = (scene.camera, scene.debug_camera)
    kw = dict(fovy=60, near=0.1, far=20, backface_culling=True)
    looking_away = (api.Camera((0.5, 1, 2), (1, 2, 4), **kw), api.Camera((0.5, 1, 2), (1, 2, 4), **kw))

    def enqueue(cameras, n):
        scene.camera, scene.debug_camera = cameras
        for _ in range(n):
            backend.render_device(scene, out.data_ptr(), stream.cuda_stream, shadows=True, no_timing=True)

    enqueue(looking_at, 1)
    stream.synchronize()
    assert not backend.overflowed()
    backend.set_list_capacities(small_pairs=4, big_pairs=2, quads=3)
    enqueue(looking_at, 1)                       # overflows its tile lists
    enqueue(looking_away, 5)                     # nothing on screen: these overflow nothing
    stream.synchronize()
    assert backend.overflowed(), "the overflow of an earlier frame of the stream was forgotten"
    for _ in range(8):                           # lists were grown; the frame may need to grow them once or twice more
        enqueue(looking_at, 1)
        enqueue(looking_away, 2)
        stream.synchronize()
        if not backend.overflowed():
            break
    else:
        raise AssertionError("work lists kept overflowing")
    enqueue(looking_at, 1)
    stream.synchronize()
    assert not backend.overflowed()
    assert np.array_equal(out.cpu().numpy(), want)
    scene.close()


def test_frame_constant_cache_holds_what_it_is_keyed_on(api):
    """The per-frame constants and the overlay lists are cached under the ids of the cameras and their MVPs:
    the cache keeps those objects alive, so a camera created later cannot be mistaken for one of them, and a
    loop that replaces the cameras every frame renders every view."""
    scene = scenes.build(api, "cube_outward")
    scene.draw_debug_frustum = True
    backend = scene._backend()
    kw = dict(fovy=60, near=0.1, far=20, backface_culling=True)
    frames = []
    for k in range(6):
        pos = (0.5 + 0.3 * (k % 3), 1, 2)
        scene.camera, scene.debug_camera = api.Camera(pos, (0, 0, 0), **kw), api.Camera(pos, (0, 0, 0), **kw)
        frames.append(scene.render())
        assert backend._packed_refs[0] is scene.camera and backend._packed_refs[2] is scene.camera.MVP
        assert backend._overlay_refs[1] is scene.debug_camera
    for k in range(3):
        assert np.array_equal(frames[k], frames[k + 3])
        assert not np.array_equal(frames[k], frames[(k + 1) % 3])
    scene.close()


def test_errors_are_loud(api):
    from py_numpy_renderer_amd import _native
    lib = _native.load_library()
    assert lib.mr_device_available() == 1
    scene = scenes.build(api, "cube_small")
    scene.models[0]._faces = scene.models[0]._faces[:, :, :3]          # corners without a material column
    with pytest.raises(ValueError, match="_faces"):
        scene.render()
    scene.close()
    scene = scenes.build(api, "cube_small")
    backend = scene._backend()
    with pytest.raises(RuntimeError, match="whole frame"):      # a part of a split frame: mr_render_device + mr_overlay_apply
        backend.render(scene, row_band=(0, 60), overlay=True)
    scene.close()


FULL = ["c1_diablo_800x600", "c2_diablo_1080p", "c3_diablo_floor_1080p", "c4_torus200k_1080p", "c5_torus1m_4k_skybox"]


@pytest.mark.parametrize("name", FULL)
def test_full_size_config_matches_reference_capture(api, oracle_mod, name):
    """BASELINE.json configs 1-5 (800x600, 1920x1080; c5: 1M triangles + skybox at 3840x2160) against what the reference itself rendered
    (uint8 frame, winner map, stencil and per-row sums of the z-buffer bit patterns are
    committed; the full float buffers are too large to commit, so they are checked against the
    oracle, which the CPU suite pins to the same captures)."""
    g, meta = load_golden(name)
    scene = scenes.build(api, name)
    shadows = name not in scenes.NO_SHADOW
    backend = scene._backend()
    out = backend.render(scene, shadows=shadows, keep_float=True)
    z = backend.read_z()
    assert np.array_equal(z.view(np.uint64).sum(axis=1, dtype=np.uint64), g["z_row_sums"]), "z row sums differ"
    assert np.array_equal(backend.read_winner(), g["winner"])
    assert np.array_equal(backend.read_stencil(), g["stencil"])
    d = np.abs(out.astype(np.int16) - g["out"].astype(np.int16))
    assert d.max() <= 1, f"uint8 frame off by {d.max()}"
    st = backend.last_stats
    assert st["frag_tri"] == meta["counts"]["frag_tri_pass1"]
    assert st["frag_quad"] == meta["counts"]["frag_quad"]
    assert st["n_quads"] == meta["counts"]["n_quads"]
    want = oracle_mod.render(scene, shadows=shadows, want_status=False, want_silhouette=False)
    assert np.array_equal(z.view(np.uint64), want.z.view(np.uint64))
    err = np.abs(backend.read_frame_f32().astype(np.float64) - want.frame.astype(np.float64))
    assert err.max() <= 2e-6, f"float frame off by {err.max():.3g}"
    scene.close()


def test_large_frame_properties(api):
    """Size-independent properties at the benchmark size: the frame is identical when rendered
    whole or as 8 bands, every covered pixel has a finite z, uncovered pixels keep +inf and the
    background colour, and stencil is zero wherever no shadow quad was drawn."""
    scene = scenes.build(api, "c4_torus200k_1080p")
    backend = scene._backend()
    full = backend.render(scene, shadows=True)
    z, winner = backend.read_z(), backend.read_winner()
    assert np.isfinite(z[winner >= 0]).all() and np.isinf(z[winner < 0]).all()
    bg = (np.float32([64 / 255, 0.5, 198 / 255]) ** np.float32(0.8) * 255).astype(np.uint8)
    assert (np.abs(full[::-1][winner < 0].astype(int) - bg.astype(int)) <= 1).all()
    parts = [backend.render(scene, shadows=True, row_band=(i * 135, (i + 1) * 135)) for i in range(8)]
    assert np.array_equal(np.concatenate(parts, axis=0), full)
    no_shadow = backend.render(scene, shadows=False)
    assert (backend.read_stencil() == 0).all()
    assert (no_shadow.astype(int) >= full.astype(int) - 1).all()      # shadows only darken
    scene.close()


@pytest.mark.parametrize("name", SMALL + FULL)
def test_face_status_and_stdout_match_reference(api, capsys, name):
    """Per-face codes of the lit pass (upstream's Errors flags) and the three lines upstream
    prints per model (obj/core.py:634-636), captured from the reference run."""
    g, meta = load_golden(name)
    scene = scenes.build(api, name)
    shadows = name not in scenes.NO_SHADOW
    backend = scene._backend()
    backend.render(scene, shadows=shadows, face_status=True)
    assert np.array_equal(backend.read_face_status(), g["face_status"])
    scene.verbose = True
    capsys.readouterr()
    scene.render(shadows=shadows)
    assert capsys.readouterr().out == meta["stdout"]
    scene.close()


def _against_oracle(api, oracle_mod, scene, shadows=True, label=""):
    backend = scene._backend()
    out = backend.render(scene, shadows=shadows, keep_float=True)
    want = oracle_mod.render(scene, shadows=shadows)
    _assert_buffers(backend, out, want.z, want.winner, want.stencil, want.frame, want.out, label)
    return out, want


def test_edge_cases_against_oracle(api, oracle_mod):
    """Empty scene, everything off screen, 1x1 and odd-sized frames, several models, faces cut by
    the screen border, a face exactly filling one pixel box."""
    cam, dbg = scenes._std_cameras(api)
    empty = scenes._scene(api, cam, dbg, scenes._std_light(api), (9, 13), [])
    out, want = _against_oracle(api, oracle_mod, empty, label="empty scene")
    assert (want.winner == -1).all()
    empty.close()

    cam, dbg = scenes._std_cameras(api)
    far_away = scenes._floor(api, textured=False) @ api.translation((100.0, 0.0, 0.0))
    off = scenes._scene(api, cam, dbg, scenes._std_light(api), (33, 47), [far_away])
    _against_oracle(api, oracle_mod, off, label="off screen")
    off.close()

    for res in ((1, 1), (7, 5), (17, 31), (250, 333)):
        cam, dbg = scenes._std_cameras(api)
        cube = api.Model.load_model(__import__("os").path.join(scenes.ASSETS, "cube", "cube.obj"))
        cube.normals = -cube.normals
        sc = scenes._scene(api, cam, dbg, scenes._std_light(api), res, [cube @ api.scale(0.7), scenes._floor(api)])
        _against_oracle(api, oracle_mod, sc, label=f"resolution {res}")
        sc.close()

    # three models, one of them the same mesh twice (coincident surfaces: every z ties, later face wins)
    cam, dbg = scenes._std_cameras(api)
    a, b = scenes._floor(api), scenes._floor(api, textured=False)
    tet = api.Model.load_model(scenes.bare_tetra_obj())
    sc = scenes._scene(api, cam, dbg, scenes._std_light(api), (96, 128), [a, tet, b])
    out, want = _against_oracle(api, oracle_mod, sc, label="coincident models")
    n_first = len(a._faces) + len(tet._faces)
    assert (want.winner[(want.winner >= 0) & (want.winner != 2) & (want.winner != 3) & (want.winner != 4) & (want.winner != 5)] >= n_first).all()
    sc.close()


def test_spot_and_directional_lights_full_pipeline(api, oracle_mod):
    for kind in (api.Lightning.SPOT_LIGHTNING, api.Lightning.DIRECTIONAL_LIGHTNING):
        cam, dbg = scenes._std_cameras(api)
        light = api.Light((2, 3, 4), light_type=kind, ambient_strength=0.1, specular_strength=0.3)
        sc = scenes._scene(api, cam, dbg, light, (150, 200), [scenes._torus(api, 24, 16), scenes._floor(api)])
        _against_oracle(api, oracle_mod, sc, label=str(kind))
        sc.close()


def test_scene_changes_are_picked_up(api, oracle_mod):
    """Adding a model, moving one (Model @ M) and registering a texture all re-upload the scene."""
    cam, dbg = scenes._std_cameras(api)
    floor = scenes._floor(api, textured=False)
    sc = scenes._scene(api, cam, dbg, scenes._std_light(api), (90, 120), [floor])
    first = sc.render()
    sc.add_model(api.Model.load_model(scenes.bare_tetra_obj()))
    second = sc.render()
    assert not np.array_equal(first, second)
    _against_oracle(api, oracle_mod, sc, label="after add_model")
    sc.models[1] = sc.models[1] @ api.translation((0.3, 0.2, 0.0))
    _against_oracle(api, oracle_mod, sc, label="after Model @ M")
    floor.textures.register("diffuse", __import__("os").path.join(scenes.ASSETS, "floor_diffuse.tga"), normalize=False)
    _against_oracle(api, oracle_mod, sc, label="after texture register")
    sc.close()


def test_in_place_edits_are_picked_up(api, oracle_mod):
    """Editing a model's arrays in place, or a material's fields the way parse_mtl assigns them, between
    two renders re-uploads the scene (the change detector fingerprints content, not object identity)."""
    cam, dbg = scenes._std_cameras(api)
    tet = api.Model.load_model(scenes.bare_tetra_obj())
    sc = scenes._scene(api, cam, dbg, scenes._std_light(api), (90, 120), [tet, scenes._floor(api, textured=False)])
    first = sc.render()
    tet.vertices[:, :3] *= np.float32(0.7)                      # in place: same array object
    second = sc.render()
    assert not np.array_equal(first, second)
    _against_oracle(api, oracle_mod, sc, label="after in-place vertex edit")
    tet.materials["default"].Kd = ["0.9", "0.2", "0.1"]           # the way parse_mtl assigns (obj/core.py:346)
    third = sc.render()
    assert not np.array_equal(second, third)
    _against_oracle(api, oracle_mod, sc, label="after material edit")
    tet.vertices[0, 1] += np.float32(0.2)                       # a single element: needs invalidate()
    tet.invalidate()
    _against_oracle(api, oracle_mod, sc, label="after invalidate()")
    sc.close()


@pytest.mark.parametrize("in_flight", [1, 4])
def test_frames_in_flight_all_match(api, in_flight):
    """bench.py's mode: successive frames enqueued on several HIP streams (mr_render_device), each
    with its own work buffers inside the library.  Every frame that comes out must be the frame
    a plain Scene.render() returns, whichever stream and slot produced it."""
    import torch
    from py_numpy_renderer_amd.multigpu import BandRenderer
    scene = scenes.build(api, "diablo_floor_small")
    want = scene.render()
    br = BandRenderer(scene, 0, 1, shadows=True, light_timing=True, frames_in_flight=in_flight)
    frames = [br.step() for _ in range(3 * in_flight)]
    br.synchronize()
    torch.cuda.synchronize()
    for i, frame in enumerate(frames[-in_flight:]):
        assert np.array_equal(frame.cpu().numpy(), want), f"frame of lane {i} differs"
    scene._backend().stats()            # raises if a work list overflowed on any lane
    scene.close()


def test_random_views_against_oracle(api, oracle_mod):
    """Twenty seeded random cameras / lights / handedness over three small meshes (a camera inside
    the mesh, behind it, grazing views, a light below the floor ...).  The fixed scenes pin the
    oracle to the reference; this holds the HIP path to the oracle away from them: z, winners and
    stencil bit-exact, float frame within 2e-6, in both handedness conventions and with the
    frame-only mode rendering the same frame."""
    rng = np.random.default_rng(20261005)
    recipes = (lambda: [scenes._torus(api, 24, 16), scenes._floor(api)],
               lambda: [api.Model.load_model(scenes.bare_tetra_obj()), scenes._floor(api, textured=False)],
               lambda: [scenes._diablo(api) @ api.scale(0.8), scenes._floor(api)])
    kinds = (api.Lightning.POINT_LIGHTNING, api.Lightning.SPOT_LIGHTNING, api.Lightning.DIRECTIONAL_LIGHTNING)
    for i in range(20):
        eye = rng.uniform(-2.5, 2.5, 3)
        eye[1] = rng.uniform(-0.5, 3.0)
        if i % 5 == 0:
            eye *= 0.15                                  # inside / right next to the mesh: faces cross every plane
        target = rng.uniform(-0.4, 0.4, 3)
        kw = dict(fovy=float(rng.uniform(35, 95)), near=float(rng.uniform(0.05, 0.5)), far=float(rng.uniform(4, 30)),
                  backface_culling=bool(i % 3))
        cam, dbg = api.Camera(tuple(eye), tuple(target), **kw), api.Camera(tuple(eye), tuple(target), **kw)
        light = api.Light(tuple(rng.uniform(-4, 4, 3)), light_type=kinds[i % 3], ambient_strength=0.1,
                          specular_strength=float(rng.uniform(0, 0.6)))
        res = (int(rng.integers(40, 200)), int(rng.integers(40, 260)))
        over = {} if i % 4 else dict(system=api.SYSTEM.LH, subsystem=api.SUBSYSTEM.OPENGL)
        sc = scenes._scene(api, cam, dbg, light, res, recipes[i % 3](), **over)
        out, _ = _against_oracle(api, oracle_mod, sc, label=f"random view {i}")
        assert np.array_equal(sc.render(), out), f"random view {i}: frame-only mode differs"
        sc.close()


def test_pipelined_frames_equal_synchronous_ones(api):
    """Scene.render_frames / render_async (mr_render_async + mr_render_wait: two to four frames in flight on the
    library's lanes, the host copy of one frame running beside the kernels of the next): sixteen frames of a moving
    camera, overlay on as by default, must be the sixteen frames render() returns one at a time -- also when the
    work lists start far too small and every lane has to grow them."""
    scene = scenes.build(api, "diablo_floor_small")
    scene.draw_debug_frustum = True
    kw = dict(fovy=60, near=0.1, far=20, backface_culling=True)

    def views():
        for k in range(16):
            pos = (0.5 + 0.02 * k, 1.0, 2.0 - 0.015 * k)
            yield api.Camera(pos, (0, 0, 0), **kw), api.Camera(pos, (0, 0, 0), **kw)

    want = []
    for cam, dbg in views():
        scene.camera, scene.debug_camera = cam, dbg
        want.append(scene.render().copy())
    assert any(not np.array_equal(want[0], w) for w in want[1:])
    for depth in (2, 4):
        got = [f.copy() for f in scene.render_frames(views(), depth=depth)]
        assert len(got) == 16
        for k in range(16):
            assert np.array_equal(got[k], want[k]), f"depth {depth}: frame {k} differs"
    pending = [scene.render_async() for _ in range(4)]
    with pytest.raises(RuntimeError):
        scene.render_async()
    last = [p.result() for p in pending]
    assert all(np.array_equal(f, want[-1]) for f in last)
    scene.close()

    scene = scenes.build(api, "diablo_floor_small")
    scene._backend().set_list_capacities(small_pairs=4, big_pairs=2, quads=3, work=16)
    got = [f.copy() for f in scene.render_frames(views(), depth=3)]
    scene.draw_debug_frustum = False
    for k, (cam, dbg) in enumerate(views()):
        scene.camera, scene.debug_camera = cam, dbg
        assert np.array_equal(got[k], scene.render()), f"small lists: frame {k} differs"
    scene.close()


@pytest.mark.parametrize("name", ["c4_torus200k_1080p", "diablo_floor_small", "torus_skybox_small", "tetra_ortho", "diablo_closeup_noclip"])
def test_cluster_culling_changes_nothing_but_the_work(api, name, monkeypatch):
    """k_setup drops whole 64-face clusters that are off the screen, off the device's rows or turned away from the
    camera before reading a face of them (cluster_culled): the frame, the z-buffer, the winners and the set-up counts
    must be those of the per-face tests, on whole frames and on one rank's rows, and on a closed mesh a good part of
    the clusters must actually go."""
    scene = scenes.build(api, name)
    backend = scene._backend()
    shadows = name not in scenes.NO_SHADOW
    results = {}
    for mode in ("0", "count", "box"):
        monkeypatch.setenv("MR_CLUSTER_CULL", mode)
        full = backend.render(scene, shadows=shadows, counters=False, keep_buffers=True).copy()
        culled = backend.clusters_culled()
        taps = (backend.read_z().copy(), backend.read_winner().copy(), backend.read_stencil().copy())
        stats = {k: backend.last_stats[k] for k in ("n_faces_setup", "n_quads", "n_quads_drawn", "tri_bin_entries", "quad_bin_entries")}
        h = full.shape[0]
        band = backend.render(scene, shadows=shadows, counters=False, row_band=(h // 4, h // 2)).copy()
        band_culled = backend.clusters_culled()
        stripe = backend.render(scene, shadows=shadows, counters=False, stripe=(1, 3)).copy()
        results[mode] = (full, taps, stats, band, stripe, culled, band_culled)
    ref = results["0"]
    assert ref[5] == 0 and ref[6] == 0
    monkeypatch.delenv("MR_CLUSTER_CULL")           # the default: on for part of a frame, off for a whole one
    assert np.array_equal(backend.render(scene, shadows=shadows, counters=False, row_band=(h // 4, h // 2)), ref[3])
    assert np.array_equal(backend.render(scene, shadows=shadows, counters=False), ref[0])
    for mode in ("count", "box"):
        got = results[mode]
        assert np.array_equal(got[0], ref[0]), mode
        for a, b in zip(got[1], ref[1]):
            assert np.array_equal(a, b, equal_nan=True), mode
        assert got[2] == ref[2], mode
        assert np.array_equal(got[3], ref[3]) and np.array_equal(got[4], ref[4]), mode
    n_clusters = -(-sum(len(m._faces) for m in scene.models) // 64)
    if name == "c4_torus200k_1080p":
        assert results["count"][5] > 0.25 * n_clusters, (results["count"][5], n_clusters)      # back faces of the torus
        assert results["count"][6] > results["count"][5]                                     # and what is not on the band's rows
    scene.close()


def test_the_frames_bench_times_are_the_oracles_frames(api, oracle_mod):
    """bench.py's timed region renders eight views of c4 (a 0.05-degree camera swing); only one of them is a reference
    capture.  Every one of the eight against the C oracle at full size: z bit patterns, winners, stencil, the uint8
    frame within 1 -- and the frame-only mode bench.py times against the counted one."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    scene = scenes.build(api, "c4_torus200k_1080p")
    backend = scene._backend()
    for k, (cam, dbg) in enumerate(bench.swing_cameras(api, scene, bench.N_VIEWS)):
        scene.camera, scene.debug_camera = cam, dbg
        out = backend.render(scene, shadows=True)
        z, winner, stencil = backend.read_z(), backend.read_winner(), backend.read_stencil()
        want = oracle_mod.render(scene, shadows=True, want_status=False, want_silhouette=False)
        assert np.array_equal(z.view(np.uint64), want.z.view(np.uint64)), f"view {k}: z"
        assert np.array_equal(winner, want.winner), f"view {k}: winners"
        assert np.array_equal(stencil, want.stencil), f"view {k}: stencil"
        assert np.abs(out.astype(np.int16) - want.out.astype(np.int16)).max() <= 1, f"view {k}: frame"
        assert np.array_equal(backend.render(scene, shadows=True, counters=False), out), f"view {k}: frame-only mode"
    scene.close()
