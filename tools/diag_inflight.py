import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import scenes
from py_numpy_renderer_amd.multigpu import BandRenderer
api = scenes.product_api()
scene = scenes.build(api, "diablo_floor_small")
want = scene.render()
br = BandRenderer(scene, 0, 1, shadows=True, light_timing=True, frames_in_flight=1)
br.synchronize(); torch.cuda.synchronize()
got = br.frame.cpu().numpy()
print("after priming: differing values", (got != want).sum())
for k in range(3):
    f = br.step(); br.synchronize(); torch.cuda.synchronize()
    got = f.cpu().numpy()
    d = (got != want).any(axis=2)
    ys, xs = np.nonzero(d)
    print("step", k, "differing px", d.sum(), "rows", (ys.min(), ys.max()) if len(ys) else None, "cols", (xs.min(), xs.max()) if len(xs) else None)
print(scene._backend().stats())
