"""Diagnostic (GPU box): distribution of the per-tile list sizes (small pairs, big pairs, shadow quads)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import scenes
api = scenes.product_api()
for name in sys.argv[1:]:
    sc = scenes.build(api, name)
    be = sc._backend()
    shadows = name not in scenes.NO_SHADOW
    be.render(sc, shadows=shadows)
    st = dict(be.last_stats)
    r = be.read_tile_records().astype(np.int64)
    small, big, quads = r[:, 7] & 0xfff, (r[:, 7] >> 12) & 0x3ff, (r[:, 7] >> 22) & 0x3ff
    print(name, "tiles", len(r), {k: st[k] for k in ("n_faces_setup", "n_quads", "n_quads_drawn", "tri_bin_entries", "quad_bin_entries")})
    for label, v in (("small", small), ("big", big), ("quads", quads)):
        nz = v[v > 0]
        print(f"  {label}: nonzero tiles {len(nz)} sum {v.sum()} max {v.max()} pct50/90/99 of nonzero",
              np.percentile(nz, [50, 90, 99]).round(0) if len(nz) else None)
    for thr in (16, 32, 48, 64, 96, 128, 256):
        m = quads > thr
        print(f"  quads > {thr}: {m.sum()} tiles holding {quads[m].sum()} of {quads.sum()} pairs")
    sc.close()
