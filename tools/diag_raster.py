"""Diagnostic (GPU box): per-kernel times of the c4 scene under ablations."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import scenes

api = scenes.product_api()
name = sys.argv[1] if len(sys.argv) > 1 else "c4_torus200k_1080p"
for label, shadows, clip in (("full", True, True), ("no-shadow", False, True), ("no-clip", True, False), ("neither", False, False)):
    sc = scenes.build(api, name)
    for m in sc.models:
        m.clip = clip
    be = sc._backend()
    for _ in range(12):
        be.render(sc, shadows=shadows)
    kt, n = be.kernel_times(10)
    st = be.last_stats
    print(f"{label:10s}", {k: round(v * 1e3, 1) for k, v in kt.items()}, "entries", st["tri_bin_entries"], st["quad_bin_entries"],
          "frags", st["frag_tri"], st["frag_quad"], "quads", st["n_quads_drawn"], flush=True)
    sc.close()
