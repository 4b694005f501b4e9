import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, scenes
from py_numpy_renderer_amd.multigpu import BandRenderer
api = scenes.product_api()
for name in ("c2_diablo_1080p", "c4_torus200k_1080p"):
    sc = scenes.build(api, name)
    br = BandRenderer(sc, shadows=name not in scenes.NO_SHADOW, frames_in_flight=3, timing_every=0)
    for _ in range(200): br.step()
    br.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n): br.step()
    t1 = time.perf_counter()
    br.synchronize()
    t2 = time.perf_counter()
    print(name, "host enqueue us/frame", round((t1 - t0) / n * 1e6, 2), "| until drained us/frame", round((t2 - t0) / n * 1e6, 2))
