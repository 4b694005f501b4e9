"""GPU box: wall time per frame when this device renders only the share of one rank of an N-GPU split
(what that rank does per frame, minus the all-gather), for both partitions: contiguous row bands and
interleaved tile-row stripes.  Frames are enqueued back to back on one stream without event marks and
timed with the host clock over many frames.   usage: tools/time_band.py <scene> <N> [N ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import scenes
from py_numpy_renderer_amd.multigpu import BandRenderer
api = scenes.product_api()
name = sys.argv[1]
sc = scenes.build(api, name)
shadows = name not in scenes.NO_SHADOW
h = sc.resolution[0]


IN_FLIGHT = int(os.environ.get("FRAMES_IN_FLIGHT", "1"))       # 1: one stream, frames back to back; 3: bench.py's default


STREAMS = [torch.cuda.Stream() for _ in range(IN_FLIGHT)]


def per_frame(world, rank, partition, frames=300):
    br = BandRenderer(sc, rank, world, shadows=shadows, frames_in_flight=IN_FLIGHT, partition=partition, timing_every=0,
                      streams=STREAMS)
    br.world = 1                                    # this device only: no collective
    br.index = None
    for _ in range(30):
        br.step()
    br.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        br.step()
    br.synchronize()
    return (time.perf_counter() - t0) / frames * 1e6


print(name, "whole frame us", round(per_frame(1, 0, "bands"), 1), flush=True)
for n in (int(a) for a in sys.argv[2:]):
    for partition in os.environ.get("PARTITIONS", "bands stripes weighted").split():
        if partition == "bands" and h % n:
            continue
        t = [per_frame(n, r, partition) for r in range(n)]
        print(f"{name} {n} ranks, {partition}: per-rank us {[round(x, 1) for x in t]} -> slowest {max(t):.1f}", flush=True)
