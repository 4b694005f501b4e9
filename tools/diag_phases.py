"""GPU box, diagnostic build MR_ABLATE=30 (tools/make_ablate.py): where a tile's time goes, phase by phase."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import scenes
api = scenes.product_api()
name = sys.argv[1] if len(sys.argv) > 1 else "c4_torus200k_1080p"
sc = scenes.build(api, name)
be = sc._backend()
for _ in range(5):
    be.render(sc, shadows=name not in scenes.NO_SHADOW, counters=False)
r = be.read_tile_records().astype(np.int64)
start, end, t_lists, t_big, t_s0, t_raster, t_quads = r[:, 8], r[:, 9], r[:, 0], r[:, 1], r[:, 2], r[:, 10], r[:, 11]
small, big, quads = r[:, 5], r[:, 6], r[:, 7]
live = (small > 0) | (big > 0)
d = lambda a, b: ((b - a) & 0xffffffff) * 10e-3
ph = {"lists known": d(start, t_lists), "big pairs": d(t_lists, t_big), "sweep 0": d(t_big, t_s0), "sweep 1": d(t_s0, t_raster),
      "quads": d(t_raster, t_quads), "shade+end": d(t_quads, end)}
for sel, label in ((live & (small == 0), "big-only tiles"), (small > 0, "mesh tiles"), ((small > 64), "mesh tiles > 64 pairs")):
    print(label, int(sel.sum()), "mean us:", {k: round(float(v[sel].mean()), 2) for k, v in ph.items()},
          "total", round(float(d(start, end)[sel].mean()), 2), "| mean small", round(float(small[sel].mean()), 1), "quads", round(float(quads[sel].mean()), 1))
