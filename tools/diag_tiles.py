"""Diagnostic (GPU box): timeline of the visibility kernel's workgroups."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import scenes
api = scenes.product_api()
name = sys.argv[1] if len(sys.argv) > 1 else "c4_torus200k_1080p"
shadows = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
sc = scenes.build(api, name)
be = sc._backend()
for _ in range(5):
    be.render(sc, shadows=shadows)
r = be.read_tile_records().astype(np.int64)
start, end = r[:, 5], r[:, 6]
t0 = start.min()
dur = (end - start) * 10e-3          # us
small, big, quads = r[:, 7] & 0xfff, (r[:, 7] >> 12) & 0x3ff, (r[:, 7] >> 22) & 0x3ff
print("tiles", len(r), "kernel span us", (end.max() - t0) * 10e-3, "sum dur us", dur.sum(), "mean", dur.mean(), "max", dur.max())
print("list sizes: small max", small.max(), "sum", small.sum(), "| big max", big.max(), "sum", big.sum(), "| quads max", quads.max(), "sum", quads.sum())
order = np.argsort(-dur)[:8]
for i in order:
    print(f"  tile {i}: dur {dur[i]:.1f} us start +{(start[i]-t0)*10e-3:.1f} small {small[i]} big {big[i]} quads {quads[i]}")
for lo, hi in ((0, 1), (1, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 1000)):
    m = (dur >= lo) & (dur < hi)
    print(f"  dur [{lo},{hi}) us: {m.sum()} tiles, mean small {small[m].mean() if m.any() else 0:.1f} quads {quads[m].mean() if m.any() else 0:.1f}")
st = (start - t0) * 10e-3
print("start-time percentiles us:", np.percentile(st, [10, 50, 90, 99, 100]).round(1))
