"""bench.py -- Mfrag/s and frames/s of the rasterisation hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[3], the scene the metric is quoted on; it fits one GPU):
synthetic 200 000-triangle torus + floor, 1920x1080, point light, shadow volumes -- the scene
recipe of SURVEY.md section 8(d) / BASELINE.md row c4, generated deterministically (no RNG).

A step is one frame: vertex transform, triangle set-up, silhouette + shadow-quad set-up,
binning, tile visibility (coverage, z, stencil), deferred shading and finalise to uint8, with
the scene already resident in HBM and the frame left in HBM (``mr_render_device``).  With N > 1
every rank renders a band of H/N output rows and ONE RCCL all-gather assembles the frame on
every rank (fixed total work -> "strong" scaling).

value = reference-equivalent fragments per frame (2 x triangle fragments + shadow-quad
fragments: the reference rasterises every triangle in two passes) / time.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

# Successive frames are rendered on separate HIP streams so that their short, latency-bound
# kernels overlap on the device.  ROCm multiplexes a process's streams onto 4 hardware queues by
# default; with 4 frames in flight plus the library's own stream two frames then share a queue
# and run back to back.  Must be set before the HIP runtime starts (i.e. before importing torch).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E peak 8 TB/s (spec)


def algorithmic_bytes(st, npx, n_faces, textured_spec=False):
    """Algorithmic HBM bytes of one frame in the reference's buffer formats:
    z float64, stencil int16, colour float32x3, texel float32x3.  Returned per kernel family
    and in total.  ``covered``/``lit`` pixel counts stand in for the reference's order-dependent
    "fragments that passed z" counts (they are lower bounds, so the figure is conservative)."""
    f_tri, f_quad = st["frag_tri"], st["frag_quad"]
    covered, lit, upd = st["covered_px"], st["lit_px"], st["stencil_updates"]
    visibility_tris = (8 * f_tri + 8 * covered       # pass 1: z read per fragment, z write per pass
                       + 10 * f_tri                  # pass 2: z + stencil read per fragment
                       + 10 * npx)                   # clear of z (8) and stencil (2)
    visibility_quads = 8 * f_quad + 4 * upd          # quads: z read, stencil read-modify-write
    visibility = visibility_tris + visibility_quads
    shading = ((12 + 12) * covered                   # pass 1: Kd texel + colour write
               + (12 + 12 + 12 + (4 if textured_spec else 0)) * lit   # pass 2: Kd + normal texel + write
               + (12 + 15) * npx)                    # frame clear (12) + finalise read 12 / write 3
    primitives = 2 * (120 + 48) * n_faces + 32 * st["n_quads"]        # attributes + indices per pass
    return dict(visibility=visibility, visibility_tris=visibility_tris, visibility_quads=visibility_quads,
                shading=shading, primitives=primitives,
                total=visibility + shading + primitives)


def cpu_baseline(scene, frags_per_frame, budget_s=12.0):
    """The C oracle (oracle/, kind "port") on this host, one thread, same scene, bounded sample."""
    from oracle import oracle
    from py_numpy_renderer_amd._pack import pack_scene
    packed = pack_scene(scene, shadows=True)
    oracle.render_packed(packed, want_status=False, want_silhouette=False)      # warm
    n, t0 = 0, time.perf_counter()
    while True:
        oracle.render_packed(packed, want_status=False, want_silhouette=False)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or n >= 40:
            break
    return {"value": round(frags_per_frame * n / dt / 1e6, 3), "unit": "Mfrag/s", "cores": 1, "kind": "port",
            "sample": f"{n} frames of the same 200k-triangle 1080p scene in {dt:.1f} s on 1 of "
                      f"{os.cpu_count()} host cores (oracle/raster_oracle.c; the reference's own NumPy loop "
                      "measured 0.0239 Mfrag/s on this scene in the build container, BASELINE.md)",
            "frames_per_s": round(n / dt, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--all-marks", action="store_true", help="time every stage (9 event marks per frame instead of 5)")
    ap.add_argument("--frames-in-flight", type=int, default=4,
                    help="successive frames rendered on this many HIP streams (1 = one frame at a time)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
        args.gpus = world

    import torch
    import torch.distributed as dist
    import scenes
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    H, W = 1080, 1920
    api = scenes.product_api()
    scene = scenes.torus_floor(api, resolution=(H, W), nu=500, nv=200)
    scene.device = local_rank
    backend = scene._backend()
    n_faces = sum(len(m._faces) for m in scene.models)

    from py_numpy_renderer_amd.multigpu import BandRenderer

    # reference-equivalent fragment count of the WHOLE frame (one full render on this rank)
    backend.render(scene, shadows=True)
    full = dict(backend.last_stats)
    frags_per_frame = 2 * full["frag_tri"] + full["frag_quad"]
    frags_unique = full["frag_tri"] + full["frag_quad"]

    br = BandRenderer(scene, rank, world, shadows=True, light_timing=not args.all_marks,
                      frames_in_flight=args.frames_in_flight)
    rows = br.band[1] - br.band[0]
    step = br.step
    # The counters of this rank's band, from one counted frame (MR_FRAME_COUNTERS).  The timed
    # frames are rendered the way Scene.render() renders them, without the counters: like the
    # reference, which counts nothing, they produce the frame only; it is checked below against
    # a counted single-device frame.
    backend.render(scene, shadows=True, row_band=br.band)
    band_stats = dict(backend.last_stats)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    backend.stats()                      # raises if a work list overflowed during warm-up (it is then grown)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    backend.stats()                      # raises if a work list overflowed in the timed frames
    ktimes, n_avg = backend.kernel_times(min(args.steps, 128))

    # The same frames again, one at a time and with every stage marked (outside the timed
    # region): with several frames in flight a kernel shares the device with the other frames'
    # kernels, so its duration above says how long it was resident, not how fast it runs.
    solo = BandRenderer(scene, rank, world, shadows=True, light_timing=False, frames_in_flight=1)
    for _ in range(40):
        solo.step()
    solo.synchronize()
    ktimes_solo, n_solo = backend.kernel_times(32)

    if rank == 0:
        # the frame every rank now holds must be the frame a single device renders
        import numpy as np
        got = br.frame.cpu().numpy()
        want = backend.render(scene, shadows=True)
        assert np.array_equal(got, want), "assembled frame differs from the single-device frame"

        ms = elapsed / args.steps * 1e3
        alg = algorithmic_bytes(band_stats, W * rows, n_faces)
        kernel_alg = {"tile_raster": alg["visibility_tris"], "tile_quads": alg["visibility_quads"], "shade": alg["shading"]}
        dominant = max(kernel_alg, key=lambda k: ktimes[k])
        k_ms = ktimes[dominant]
        achieved = kernel_alg[dominant] / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as fh:
                traffic = json.load(fh).get(dominant)
        line = {
            "metric": "Mfrag/s (1920x1080, 200k tris + shadow volumes)",
            "value": round(frags_per_frame / (elapsed / args.steps) / 1e6, 2),
            "unit": "Mfrag/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4),
            "frames_per_s": round(args.steps / elapsed, 2),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "torus 500x200 (200k tris) + floor, 1920x1080, point light, shadow volumes "
                                   "(BASELINE.json configs[3] / BASELINE.md c4)",
                       "fragments_per_frame": frags_per_frame, "unique_fragments_per_frame": frags_unique,
                       "faces": n_faces, "frames_in_flight": args.frames_in_flight,
                       "parallelism": f"screen row bands x{world}"
                                                         + (" + 1 RCCL all-gather" if world > 1 else "")},
            "mfrag_unique_per_s": round(frags_unique / (elapsed / args.steps) / 1e6, 2),
            "gpu_ms_per_kernel": {k: round(v, 4) for k, v in ktimes.items()},
            "frame_algorithmic_gb": round(alg["total"] / 1e9, 4),
            "frame_hbm_frac": round(alg["total"] / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 5),
            "frame_latency_ms": round(ktimes["frame"], 4),
            "roofline": {"bound": "hbm", "kernel": "k_" + dominant, "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": traffic, "algorithmic_bytes_per_launch": int(kernel_alg[dominant]),
                         "avg_launch_ms": round(k_ms, 4), "launches_averaged": n_avg,
                         "frames_in_flight": args.frames_in_flight,
                         "note": "avg_launch_ms is the event span of the kernel in the timed region, where it shares the "
                                 "device with the kernels of the other frames in flight; solo_* is the same kernel on the "
                                 "same frames rendered one at a time right after the timed region",
                         "solo_launch_ms": round(ktimes_solo[dominant], 4),
                         "solo_achieved": round(kernel_alg[dominant] / (ktimes_solo[dominant] * 1e-3) / 1e9, 2),
                         "solo_frac": round(kernel_alg[dominant] / (ktimes_solo[dominant] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
            "gpu_ms_per_kernel_solo": {k: round(v, 4) for k, v in ktimes_solo.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(scene, frags_per_frame)
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
