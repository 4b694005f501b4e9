"""Debug-frustum overlay (obj/core.py:638): host-side restatement against the reference's
post-overlay frame and z-buffer.  CPU test: applied to the oracle's buffers; GPU test: the
drop-in ``Scene.render()`` with ``draw_debug_frustum`` on."""
import numpy as np
import pytest

import scenes
from conftest import load_golden
from py_numpy_renderer_amd.frustums import bresenham_line, draw_view_frustum


def test_dda_line():
    a, b = np.array([5.0, 1.0, 0.5, 1.0]), np.array([1.0, 2.0, 0.1, 1.0])
    pts = bresenham_line(a, b)
    assert pts.shape == (4, 4) and np.allclose(pts[0], a) and np.allclose(pts[1], a + (b - a) / 4)
    assert np.array_equal(bresenham_line(b, a), pts)                 # always walked towards decreasing x
    assert bresenham_line(a, a).shape == (1, 4)


@pytest.mark.parametrize("name", scenes.OVERLAY)
def test_overlay_on_oracle_buffers_matches_reference(api, oracle_mod, name):
    g, _ = load_golden(name)
    base = name[:-len("_overlay")]
    scene = scenes.build(api, base)
    r = oracle_mod.render(scene, shadows=True)
    frame, z = r.frame.copy(), r.z.copy()
    draw_view_frustum(frame, scene.camera, scene.debug_camera, z, scene.system)
    assert np.array_equal(z.view(np.uint64), g["z_overlay"].view(np.uint64))
    assert np.abs(frame.astype(np.float64) - g["frame_overlay"].astype(np.float64)).max() <= 1e-6
    out = (frame[::-1] ** 0.8 * 255).astype(np.uint8)
    assert np.abs(out.astype(np.int16) - g["out"].astype(np.int16)).max() <= 1
    assert (out != oracle_mod.finalise(r.frame)).any(), "the overlay drew nothing"


def test_statement_lists_are_well_formed(api):
    """What mr_scene_set_overlay validates: links point forward inside their own segment, targets are pixels
    of the frame, `touched` is exactly the set of targets."""
    from py_numpy_renderer_amd.frustums import OverlayOps
    scene = scenes.build(api, "diablo_floor_lh_gl")
    ops = OverlayOps(scene.camera, scene.debug_camera, scene.resolution)
    h, w = scene.resolution
    assert ops.n_points > 100 and ops.target.shape == (5, ops.n_points) == ops.next.shape
    assert ops.target.min() >= 0 and ops.target.max() < h * w
    seg = np.repeat(np.arange(len(ops.seg_first)), ops.seg_count)
    for k in range(5):
        linked = np.nonzero(ops.next[k] >= 0)[0]
        assert (ops.next[k][linked] > linked).all() and (seg[ops.next[k][linked]] == seg[linked]).all()
        assert (ops.target[k][ops.next[k][linked]] == ops.target[k][linked]).all()
    assert np.array_equal(ops.touched, np.unique(ops.target))


@pytest.mark.gpu
@pytest.mark.parametrize("name", scenes.OVERLAY)
def test_scene_render_with_overlay_matches_reference(api, name):
    """The drop-in call with upstream's default (overlay on): the device replays the statement lists on its
    own z-buffer and float frame -- nothing is read back but the uint8 frame.  The z-buffer it leaves must
    be the reference's post-overlay z-buffer bit for bit, the float frame within 1e-6, the frame +-1."""
    g, _ = load_golden(name)
    scene = scenes.build(api, name[:-len("_overlay")])
    scene.draw_debug_frustum = True
    out = scene.render()
    d = np.abs(out.astype(np.int16) - g["out"].astype(np.int16))
    assert d.max() <= 1, f"{int((d > 1).sum())} values differ by more than 1"
    backend = scene._backend()
    backend.render(scene, counters=True, keep_float=True, overlay=True)
    assert np.array_equal(backend.read_z().view(np.uint64), g["z_overlay"].view(np.uint64))
    assert np.abs(backend.read_frame_f32().astype(np.float64) - g["frame_overlay"].astype(np.float64)).max() <= 2e-6
    plain = backend.render(scene, counters=False)
    assert (plain != out).any(), "the overlay drew nothing"
    # the explicit-lists entry point (mr_scene_set_overlay, fed by the NumPy walk) draws the same overlay
    from py_numpy_renderer_amd.frustums import OverlayOps
    backend.set_overlay_lists(OverlayOps(scene.camera, scene.debug_camera, scene.resolution, native=False))
    assert np.array_equal(backend.render(scene, counters=False, keep_buffers=False, overlay=True), out)
    backend.set_overlay_lists(OverlayOps(scene.camera, scene.debug_camera, scene.resolution, native=False), pin=False)
    assert np.array_equal(scene.render(), out)
    scene.close()


@pytest.mark.gpu
def test_overlay_is_on_by_default_like_upstream(api):
    """An unmodified caller gets upstream's frame: Scene() draws the debug camera's frustum unless told not to."""
    g, _ = load_golden("cube_outward_overlay")
    scene = scenes.cube_outward(api)
    assert api.Scene(scene.camera, scene.light, debug_camera=scene.debug_camera).draw_debug_frustum is True
    default = api.Scene(scene.camera, scene.light, debug_camera=scene.debug_camera, resolution=scene.resolution)
    for m in scene.models:
        default.add_model(m)
    out = default.render()
    assert np.abs(out.astype(np.int16) - g["out"].astype(np.int16)).max() <= 1
    default.close()


def test_native_list_builder_equals_the_numpy_walk():
    """``mr_host_overlay_build`` (the overlay's lists in host C++, csrc/host_overlay.h) against the NumPy walk
    that states what they are (frustums.OverlayOps, native=False): every array bit for bit, over random pairs of
    viewing / debug cameras -- both handednesses, the viewing camera inside and outside the debug frustum
    (dashed edges), frustums cut by the viewing planes, odd resolutions, points that wrap around the frame."""
    import py_numpy_renderer_amd as pkg
    from py_numpy_renderer_amd._native import load_library
    from py_numpy_renderer_amd.frustums import OverlayOps
    load_library()                                   # the helper is plain host code: no GPU needed
    rng = np.random.default_rng(20)
    fields = ("seg_first", "seg_count", "target", "next", "touched")
    n_points = dashed = 0
    for trial in range(160):
        lh = trial % 3 == 0
        system, subsystem = (pkg.SYSTEM.LH, pkg.SUBSYSTEM.OPENGL) if lh else (pkg.SYSTEM.RH, pkg.SUBSYSTEM.DIRECTX)
        res = (int(rng.integers(40, 200)), int(rng.integers(40, 260)))
        near = float(rng.choice([0.05, 0.1, 0.5, 1.0]))
        kw = dict(fovy=float(rng.choice([30, 60, 90])), near=near, far=near + float(rng.choice([2.0, 20.0, 400.0])),
                  backface_culling=True)
        eye = rng.standard_normal(3) * 2 + np.array([0.5, 1.0, 2.0])
        cam = pkg.Camera(tuple(eye), (0, 0, 0), **kw)
        if trial % 4 == 0:
            dbg = pkg.Camera(tuple(eye), (0, 0, 0), **kw)                      # the same frustum: border + diagonal
        else:
            kd = dict(kw, fovy=float(rng.choice([20, 45, 80])), near=float(rng.choice([0.2, 1.0])), far=float(rng.choice([3.0, 8.0])))
            kd["far"] += kd["near"]
            dbg = pkg.Camera(tuple(rng.standard_normal(3) * 1.5 + np.array([0.0, 1.0, 0.5])), tuple(rng.standard_normal(3) * 0.3), **kd)
        scene = pkg.Scene(cam, pkg.Light((2, 3, 4)), debug_camera=dbg, resolution=res, system=system, subsystem=subsystem)
        a = OverlayOps(scene.camera, scene.debug_camera, res, native=False)
        b = OverlayOps(scene.camera, scene.debug_camera, res, native=True)
        for f in fields:
            assert np.array_equal(getattr(a, f), getattr(b, f)), (trial, f)
        assert np.array_equal(a.z.view(np.uint64), b.z.view(np.uint64)), (trial, "z")
        n_points += a.n_points
        dashed += int(len(a.seg_count) > 0 and a.seg_count.min() < 30)
    assert n_points > 20000 and dashed > 10          # the trials did draw something, short (dashed / clipped) segments included


def test_device_scheme_replays_like_the_statement_lists():
    """What the device does with the lists (per segment: keep flags, bids, the winning bidder applies the segment to
    its pixel -- csrc/kernels_overlay.h, restated in NumPy by frustums.replay_bids) against the statement-by-statement
    replay that restates upstream (OverlayOps.replay), on random z-buffers and frames: z bit for bit, the float
    frame bit for bit -- 60 random camera pairs incl. the border-and-diagonal case (debug camera == camera)."""
    import py_numpy_renderer_amd as pkg
    from py_numpy_renderer_amd.frustums import OverlayOps, replay_bids
    rng = np.random.default_rng(7)
    points = 0
    for trial in range(60):
        lh = trial % 3 == 0
        system, subsystem = (pkg.SYSTEM.LH, pkg.SUBSYSTEM.OPENGL) if lh else (pkg.SYSTEM.RH, pkg.SUBSYSTEM.DIRECTX)
        res = (int(rng.integers(40, 160)), int(rng.integers(40, 220)))
        kw = dict(fovy=float(rng.choice([30, 60, 90])), near=0.1, far=float(rng.choice([5.0, 20.0])), backface_culling=True)
        eye = rng.standard_normal(3) * 2 + np.array([0.5, 1.0, 2.0])
        cam = pkg.Camera(tuple(eye), (0, 0, 0), **kw)
        if trial % 4 == 0:
            dbg = pkg.Camera(tuple(eye), (0, 0, 0), **kw)
        else:
            kd = dict(kw, fovy=float(rng.choice([20, 45, 80])), near=float(rng.choice([0.2, 1.0])), far=float(rng.choice([3.0, 8.0])) + 1.0)
            dbg = pkg.Camera(tuple(rng.standard_normal(3) * 1.5 + np.array([0.0, 1.0, 0.5])), tuple(rng.standard_normal(3) * 0.3), **kd)
        pkg.Scene(cam, pkg.Light((2, 3, 4)), debug_camera=dbg, resolution=res, system=system, subsystem=subsystem)
        ops = OverlayOps(cam, dbg, res, native=False)
        if ops.n_points == 0:
            continue
        sign = -1 if lh else 1
        # a z-buffer the lines partly pass and partly fail against: around the lines' own depths
        z0 = rng.choice(ops.z, size=res) * rng.uniform(0.7, 1.4, size=res)
        f0 = rng.uniform(0.05, 1.0, size=res + (3,)).astype(np.float32)
        za, fa, zb, fb = z0.copy(), f0.copy(), z0.copy(), f0.copy()
        ops.replay(fa, za, sign)
        replay_bids(ops, fb, zb, sign)
        assert np.array_equal(za.view(np.uint64), zb.view(np.uint64)), trial
        assert np.array_equal(fa.view(np.uint32), fb.view(np.uint32)), trial
        assert not np.array_equal(za, z0)
        points += ops.n_points
    assert points > 5000


@pytest.mark.gpu
@pytest.mark.parametrize("partition,world", [("bands", 2), ("bands", 3), ("stripes", 2), ("stripes", 3), ("stripes", 5)])
@pytest.mark.parametrize("name", scenes.OVERLAY)
def test_split_frame_with_overlay_equals_the_single_device_frame(api, name, partition, world):
    """A frame split over N devices with upstream's default overlay: every rank appends the state (z, float colour) of
    the touched pixels it owns to its rows, the parts are gathered, and the overlay is replayed on the assembled frame
    (mr_overlay_apply) -- the lines test z at pixels other ranks own, so no rank could draw its share alone.  Here one
    GPU plays every rank in turn; the assembled frame must be, byte for byte, the frame one device renders with the
    overlay on (which the reference captures pin)."""
    import torch
    from py_numpy_renderer_amd.multigpu import row_band, stripe_rows, unstripe
    scene = scenes.build(api, name[:-len("_overlay")])
    scene.draw_debug_frustum = True
    want = scene.render().copy()
    h, w = (int(v) for v in scene.resolution)
    if partition == "bands" and h % world:
        pytest.skip("rows do not split evenly")
    backend = scene._backend()
    striped = partition == "stripes"
    rows = stripe_rows(h, world) if striped else h // world
    rows_bytes = rows * w * 3
    offset = -(-rows_bytes // 16) * 16
    state = -(-backend.overlay_state_bytes() // 16) * 16
    assert state > 0
    part_bytes = offset + state
    gathered = torch.zeros(world * part_bytes, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for r in range(world):
        backend.render_device(scene, gathered.data_ptr() + r * part_bytes, 0, shadows=True, no_timing=True, overlay=True,
                              row_band=None if striped else row_band(h, r, world), stripe=(r, world) if striped else None)
    assert not backend.overflowed()                       # (synchronises)
    parts = gathered.view(world, part_bytes)[:, :rows_bytes].contiguous().view(world * rows, w, 3)
    frame = unstripe(parts, h, world) if striped else parts.clone()
    plain = frame.cpu().numpy().copy()
    torch.cuda.synchronize()
    backend.overlay_apply(gathered.data_ptr(), part_bytes, offset, world, striped, int(scene.system), frame.data_ptr(), 0)
    backend.overflowed()
    got = frame.cpu().numpy()
    assert (got != plain).any(), "the overlay drew nothing"
    assert np.array_equal(got, want)
    scene.close()
