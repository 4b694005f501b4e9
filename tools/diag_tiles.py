"""Diagnostic (GPU box): timeline of the tile kernel's workgroups."""
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
    be.render(sc, shadows=shadows, counters=False)
r = be.read_tile_records().astype(np.int64)
start, end = r[:, 8], r[:, 9]
t0 = start.min()
dur = (end - start) * 10e-3          # us
small, big, quads = r[:, 5], r[:, 6], r[:, 7]
print("tiles", len(r), "kernel span us", (end.max() - t0) * 10e-3, "sum dur us", dur.sum(), "mean", dur.mean(), "max", dur.max())
print("list sizes: small max", small.max(), "sum", small.sum(), "| big max", big.max(), "sum", big.sum(), "| quads max", quads.max(), "sum", quads.sum())
order = np.argsort(-dur)[:8]
tr, tq = (r[:, 10] - start) * 10e-3, (r[:, 11] - r[:, 10]) * 10e-3
ts = (end - r[:, 11]) * 10e-3
for i in order:
    print(f"     wave1: survivors {r[i,1]} allpass {r[i,2]} hits {r[i,4]}")
    print(f"  tile {i}: dur {dur[i]:.1f} us start +{(start[i]-t0)*10e-3:.1f} small {small[i]} big {big[i]} quads {quads[i]} | raster {tr[i]:.1f} quads {tq[i]:.1f} shade {ts[i]:.1f}")
print("phase sums (ms): raster", tr.sum() / 1e3, "quads", tq.sum() / 1e3, "shade+epilogue", ts.sum() / 1e3)
for lo, hi in ((0, 1), (1, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 1000)):
    m = (dur >= lo) & (dur < hi)
    print(f"  dur [{lo},{hi}) us: {m.sum()} tiles, sum {dur[m].sum():.0f} us, mean small {small[m].mean() if m.any() else 0:.1f} big {big[m].mean() if m.any() else 0:.1f} quads {quads[m].mean() if m.any() else 0:.1f}")
st = (start - t0) * 10e-3
print("start-time percentiles us:", np.percentile(st, [10, 50, 90, 99, 100]).round(1))
en = (end - t0) * 10e-3
print("end-time percentiles us:", np.percentile(en, [10, 50, 90, 99, 100]).round(1))
# classes of tiles
empty = (small == 0) & (big == 0) & (quads == 0)
print("empty tiles", empty.sum(), "mean dur", dur[empty].mean() if empty.any() else 0)
only_big = (small == 0) & (big > 0) & (quads == 0)
print("big-only tiles", only_big.sum(), "mean dur", dur[only_big].mean() if only_big.any() else 0)
mesh = small > 0
print("mesh tiles", mesh.sum(), "mean dur", dur[mesh].mean() if mesh.any() else 0)
q = quads > 0
print("quad tiles", q.sum(), "mean dur", dur[q].mean() if q.any() else 0)

# least-squares fit of the tile time (0.1 us units) on the list lengths: what tile_cost should be
A = np.stack([np.ones(len(r)), small, big, quads], 1).astype(np.float64)
coef, *_ = np.linalg.lstsq(A, dur * 10.0, rcond=None)
print("fit dur[0.1us] = %.1f + %.2f*small + %.2f*big + %.2f*quads" % tuple(coef))
cost = 20 + 2 * small + 30 * big + 3 * quads                        # kernels_tile.h tile_cost
for c0, q0 in ((400, 48), (350, 32), (600, 64), (800, 64)):
    m = (cost >= c0) & (quads >= q0)
    print(f"tiles that qualify for the quad split at cost >= {c0}, quads >= {q0}: {m.sum()} (HEAVY0_MAX 128); the slowest tile qualifies: {bool(m[np.argmax(dur)])}")
