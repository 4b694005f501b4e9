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


@pytest.mark.gpu
@pytest.mark.parametrize("name", scenes.OVERLAY)
def test_scene_render_with_overlay_matches_reference(api, name):
    g, _ = load_golden(name)
    scene = scenes.build(api, name[:-len("_overlay")])
    scene.draw_debug_frustum = True
    out = scene.render()
    d = np.abs(out.astype(np.int16) - g["out"].astype(np.int16))
    assert d.max() <= 1, f"{int((d > 1).sum())} values differ by more than 1"
    scene.close()
