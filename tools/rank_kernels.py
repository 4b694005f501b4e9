"""GPU box: HIP-event spans of the frame's kernels when this device renders one rank's share (every stage marked, one
frame at a time).   usage: tools/rank_kernels.py <scene> <world> <partition> [rank ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import scenes
from py_numpy_renderer_amd.multigpu import BandRenderer
api = scenes.product_api()
name, world, partition = sys.argv[1], int(sys.argv[2]), sys.argv[3]
ranks = [int(a) for a in sys.argv[4:]] or list(range(world))
sc = scenes.build(api, name)
shadows = name not in scenes.NO_SHADOW
streams = [torch.cuda.Stream()]
for rank in ranks:
    br = BandRenderer(sc, rank, world, shadows=shadows, frames_in_flight=1, partition=partition, timing_every=1, streams=streams)
    br.world, br.index = 1, None
    for _ in range(80):
        br.step()
    br.synchronize()
    t, n = br.kernel_times(64)
    print(name, f"{world} ranks, {partition}, rank {rank} band {br.band}:", {k: round(v * 1e3, 1) for k, v in t.items()}, f"({n} frames)", flush=True)
