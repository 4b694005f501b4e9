"""``MR_VERTEX_PATH=mfma``: the vertex transform as its own launch on the matrix cores (``k_vertex_mfma``, two
``v_mfma_f64_16x16x4_f64`` per 16 vertices) with ``k_setup<true>`` reading its output -- the path north_star asks
to be built and measured, not the default (MFMA).  The variable is read once per process, so
the captures are re-checked in a child process that has it set; the child also checks that the launch took place."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CHILD = r'''
import os, sys
ROOT = sys.argv[1]
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import scenes
from conftest import load_golden
api = scenes.product_api()
for name in sys.argv[2:]:
    g, meta = load_golden(name)
    scene = scenes.build(api, name)
    shadows = name not in scenes.NO_SHADOW
    backend = scene._backend()
    out = backend.render(scene, shadows=shadows, keep_float=True)
    z, want = backend.read_z(), g.get("z")
    if want is not None and want.shape == z.shape:
        assert np.array_equal(z.view(np.uint64), want.view(np.uint64)), name + ": z"
        err = np.abs(backend.read_frame_f32().astype(np.float64) - g["frame"].astype(np.float64)).max()
        assert err <= 2e-6, (name, err)
    assert np.array_equal(backend.read_winner(), g["winner"]), name + ": winner"
    assert np.array_equal(backend.read_stencil(), g["stencil"]), name + ": stencil"
    assert np.abs(out.astype(np.int16) - g["out"].astype(np.int16)).max() <= 1, name + ": uint8"
    kt, _ = backend.kernel_times(1)
    assert kt["vertex_mfma"] > 0, (name, kt)
    print("ok", name, "vertex_mfma ms", round(kt["vertex_mfma"], 5))
    scene.close()
'''


def test_mfma_vertex_path_matches_reference_captures(tmp_path):
    names = ["cube_small", "diablo_small", "diablo_floor_lh_gl", "torus_spot", "tetra_ortho", "kat_house",
             "gizmos_small", "c3_diablo_floor_1080p"]
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    env = dict(os.environ, MR_VERTEX_PATH="mfma")
    run = subprocess.run([sys.executable, str(script), ROOT] + names, env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-3000:]
    assert run.stdout.count("ok ") == len(names), run.stdout
