"""GPU box: one frame at a time on one stream (the latency regime): wall time per frame and the tile kernel's
timeline summary.   usage: tools/solo.py [scene]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
import scenes
from py_numpy_renderer_amd.multigpu import BandRenderer
api = scenes.product_api()
name = sys.argv[1] if len(sys.argv) > 1 else "c4_torus200k_1080p"
sc = scenes.build(api, name)
br = BandRenderer(sc, 0, 1, shadows=name not in scenes.NO_SHADOW, frames_in_flight=1, timing_every=0)
for _ in range(50):
    br.step()
br.synchronize()
t0 = time.perf_counter()
for _ in range(400):
    br.step()
br.synchronize()
wall = (time.perf_counter() - t0) / 400 * 1e6
r = sc._backend().read_tile_records().astype(np.int64)
start, end = r[:, 8], r[:, 9]
dur = (end - start) * 10e-3
t0 = start.min()
print(name, "wall us/frame", round(wall, 1), "| k_tile span", round((end.max() - t0) * 10e-3, 1), "slowest tile", round(dur.max(), 1),
      "sum dur ms", round(dur.sum() / 1e3, 2), "| tiles over 40 us", int((dur > 40).sum()), "over 50", int((dur > 50).sum()),
      os.environ.get("MR_SPLIT_COST"), os.environ.get("MR_SPLIT_QUADS"))
