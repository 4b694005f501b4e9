"""GPU box: host time to enqueue a frame vs device time per frame."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, scenes
from py_numpy_renderer_amd.multigpu import BandRenderer
api = scenes.product_api()
sc = scenes.torus_floor(api, resolution=(1080, 1920), nu=500, nv=200)
for fif in (1, 2, 4):
    br = BandRenderer(sc, 0, 1, shadows=True, light_timing=True, frames_in_flight=fif)
    for _ in range(20): br.step()
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n): br.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"fif {fif}: host enqueue {1e6*(t1-t0)/n:.1f} us/frame, total {1e6*(t2-t0)/n:.1f} us/frame", flush=True)
