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
