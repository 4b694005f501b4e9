"""bench.py -- Mfrag/s and frames/s of the rasterisation hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4|c5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (default c4 = BASELINE.json configs[3], the scene the metric is quoted on; it fits one
GPU): synthetic 200 000-triangle torus + floor, 1920x1080, point light, shadow volumes -- the
scene recipe of SURVEY.md section 8(d) / BASELINE.md row c4, generated deterministically (no RNG).
``--config c5`` is BASELINE.json configs[4] (1M triangles + cubemap skybox at 3840x2160), c2 / c3
the diablo configs.

A frame is the whole hot path: per-frame constants in, vertex transform, face set-up, silhouette
+ shadow-quad set-up, binning, tile visibility (coverage, z, stencil), deferred shading and
finalise to uint8, with the scene already resident in HBM and the frame left in HBM
(``mr_render_device``).  A STEP is a batch of frames -- at least ``--frames-per-step`` (default 256), and as
many as it takes for the K timed steps to last 3 s at the rate of a short calibration run (``frames_per_step``
in the line says how many) -- so the timed region does not depend on --steps; successive frames use DIFFERENT per-frame
constants (the camera swings through 8 slightly different views), and ``--frames-in-flight``
(default 3) of them are in flight on separate HIP streams.  With N > 1 every rank renders its share
of the screen tiles (contiguous row bands of equal modelled cost by default, ``--partition stripes`` for
interleaved tile rows, ``--partition bands`` for equal row bands) and ONE RCCL all-gather assembles the frame on every rank (fixed total work -> "strong").

value = reference-equivalent fragments per frame (2 x triangle fragments + shadow-quad fragments:
the reference rasterises every triangle in two passes) / time, summed over the views.
The same line also carries the other regimes: one frame at a time (``latency_ms_single``,
``value_single_frame``), with the fragment counters on (``value_counters_on``: no depth cull of
shadow quads) and the host-visible ``Scene.render()`` (``scene_render_ms_host``: packing, per-frame
upload, render, device->host copy into a NumPy array; the camera is a NEW object on every call, as when it moves,
so nothing is served from the host's caches -- ``*_same_camera`` is the same call with the cameras left alone).
"""
import argparse
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
N_VIEWS = 8
TIMED_SECONDS = 3.0          # the K timed steps together last at least this long
N_SIMDS = 1024               # 256 CUs x 4 SIMDs


def valu_roofline(config, launch_ms, solo_ms):
    """The tile kernel's VECTOR-ISSUE bound: its vector instructions by class (rocprofv3 SQ_INSTS_VALU_* of this
    config, profiles/r03_<config>_valu_mix.json) priced with the SIMD time a wave-instruction of that class takes
    on MI355X (tools/micro/valu_f64_rate.hip, measured on the same pool: profiles/r03_valu_rate.txt, the column for
    six wavefronts per SIMD), spread over the chip's 1 024 SIMDs."""
    mix_path = os.path.join(ROOT, "profiles", f"r03_{config}_valu_mix.json")
    rate_path = os.path.join(ROOT, "profiles", "r03_valu_rate.txt")
    if not (os.path.exists(mix_path) and os.path.exists(rate_path)):
        return None
    with open(mix_path) as fh:
        mix = json.load(fh)["per_launch"]["k_tile"]
    ns = {}
    with open(rate_path) as fh:
        for line in fh:
            if "6/SIMD" in line and "chain" not in line:
                name = line.split("  ")[0].strip()
                ns[name] = float(line.split("6/SIMD")[1].split("t")[1].split("ns")[0])
    cost = {"SQ_INSTS_VALU_FMA_F64": ns["v_fma_f64"], "SQ_INSTS_VALU_MUL_F64": ns["v_mul_f64"], "SQ_INSTS_VALU_ADD_F64": ns["v_add_f64"],
            "SQ_INSTS_VALU_TRANS_F64": ns["v_rcp_f64"], "SQ_INSTS_VALU_CVT": ns["v_cvt_f64_f32"]}
    classified = sum(mix.get(k, 0) for k in cost)
    other = mix["SQ_INSTS_VALU"] - classified          # 32-bit arithmetic, compares, selects, moves: the v_add_u32 / v_fma_f32 rate
    simd_ns = sum(mix.get(k, 0) * c for k, c in cost.items()) + other * max(ns["v_add_u32"], ns["v_fma_f32"])
    issue_us = simd_ns / N_SIMDS / 1e3
    return {"bound": "valu", "kernel": "k_tile", "instructions": int(mix["SQ_INSTS_VALU"]),
            "float64_instructions": int(sum(mix.get(k, 0) for k in list(cost)[:4])),
            "ns_per_wave_instruction": {k.replace("SQ_INSTS_VALU_", "").lower(): c for k, c in cost.items()} | {"other": max(ns["v_add_u32"], ns["v_fma_f32"])},
            "issue_us_per_launch": round(issue_us, 2), "unit": "us of SIMD time per launch, spread over 1 024 SIMDs",
            "frac": round(issue_us / (launch_ms * 1e3), 4) if launch_ms > 0 else None,
            "solo_frac": round(issue_us / (solo_ms * 1e3), 4) if solo_ms > 0 else None,
            "source": "profiles/r03_%s_valu_mix.json x profiles/r03_valu_rate.txt" % config}

CONFIGS = {
    "c2": ("c2_diablo_1080p", "diablo3_pose (5 022 tris), 1920x1080, Phong + normal map, z-buffer only (BASELINE.json configs[1] / BASELINE.md c2)"),
    "c3": ("c3_diablo_floor_1080p", "diablo3_pose + floor, 1920x1080, shadow volumes (BASELINE.json configs[2] / BASELINE.md c3)"),
    "c4": ("c4_torus200k_1080p", "torus 500x200 (200k tris) + floor, 1920x1080, point light, shadow volumes (BASELINE.json configs[3] / BASELINE.md c4)"),
    "c5": ("c5_torus1m_4k_skybox", "torus 1000x500 (1M tris) + floor + cubemap skybox, 3840x2160, shadow volumes (BASELINE.json configs[4] / BASELINE.md c5)"),
}
REFERENCE_MFRAGS = {"c2": 0.118, "c3": 0.609, "c4": 0.0239, "c5": 0.0224}     # BASELINE.md, reference NumPy loop, build container


def algorithmic_bytes(st, npx, n_faces, skybox=False, textured_spec=False):
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
               + (12 + 15) * npx                     # frame clear (12) + finalise read 12 / write 3
               + ((24 + 12) * npx if skybox else 0))  # skybox fill: float64 texel + colour write per pixel
    primitives = 2 * (120 + 48) * n_faces + 32 * st["n_quads"]        # attributes + indices per pass
    return dict(visibility=visibility, visibility_tris=visibility_tris, visibility_quads=visibility_quads,
                shading=shading, primitives=primitives,
                total=visibility + shading + primitives)


def cpu_baseline(scene, frags_per_frame, label, budget_s=12.0):
    """The C oracle (oracle/, kind "port") on this host, one thread, same scene, bounded sample."""
    from oracle import oracle
    from py_numpy_renderer_amd._pack import pack_scene
    packed = pack_scene(scene, shadows=True)
    sky = getattr(scene.skybox, "texels", None)
    kw = dict(want_status=False, want_silhouette=False, sky_texels=sky)
    t0 = time.perf_counter()
    oracle.render_packed(packed, **kw)      # warm
    one = time.perf_counter() - t0
    n, t0 = 0, time.perf_counter()
    while True:
        oracle.render_packed(packed, **kw)
        n += 1
        dt = time.perf_counter() - t0
        if dt + one > budget_s or n >= 40:
            break
    return {"value": round(frags_per_frame * n / dt / 1e6, 3), "unit": "Mfrag/s", "cores": 1, "kind": "port",
            "sample": f"{n} frames of the same scene ({label}) in {dt:.1f} s on 1 of "
                      f"{os.cpu_count()} host cores (oracle/raster_oracle.c; the reference's own NumPy loop "
                      "measured the Mfrag/s in reference_numpy_mfrag_s on this scene in the build container, BASELINE.md)",
            "frames_per_s": round(n / dt, 3)}


def swing_cameras(api, scene, n_views, offset=0):
    """*n_views* camera pairs on a short arc around the scene's camera (0.05 degrees apart about the
    y axis): frames of a sequence differ in their per-frame constants, like these."""
    import numpy as np
    cam = scene.camera
    kw = dict(fovy=cam.fovy, near=cam.near, far=cam.far, backface_culling=cam.backface_culling, up=cam.up,
              projection_type=cam.projection_type)
    base = np.asarray(cam.position, dtype=np.float64)
    pairs = []
    for k in range(n_views):
        a = np.deg2rad((k + offset - n_views // 2) * 0.05)
        pos = (base[0] * np.cos(a) + base[2] * np.sin(a), base[1], -base[0] * np.sin(a) + base[2] * np.cos(a))
        pairs.append((api.Camera(pos, cam.center, **kw), api.Camera(pos, cam.center, **kw)))
    return pairs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c4")
    ap.add_argument("--frames-per-step", type=int, default=256, help="frames in one step (a step is a batch of frames)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--all-marks", action="store_true", help="time every stage (5 event marks per frame instead of 3)")
    ap.add_argument("--frames-in-flight", type=int, default=3,
                    help="successive frames rendered on this many HIP streams (1 = one frame at a time)")
    ap.add_argument("--partition", choices=("stripes", "bands", "weighted"), default="weighted",
                    help="screen-tile split for --gpus > 1: interleaved tile rows, contiguous row bands, or bands of equal cost")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
        args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist
    import scenes
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # MR_BENCH_REHEARSE=1: the N-rank path on ONE GPU (every rank on device 0, gloo instead of RCCL): checks the
    # partition, the collective's layout and the assembled frame where no second GPU exists; its rates mean nothing
    rehearse = world > 1 and os.environ.get("MR_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1 and rehearse:
        dist.init_process_group("gloo")
    elif world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    scene_name, label = CONFIGS[args.config]
    shadows = scene_name not in scenes.NO_SHADOW
    api = scenes.product_api()
    scene = scenes.build(api, scene_name)
    H, W = (int(v) for v in scene.resolution)
    scene.device = local_rank
    backend = scene._backend()
    n_faces = sum(len(m._faces) for m in scene.models)
    skybox = hasattr(scene.skybox, "texels")
    base_cameras = (scene.camera, scene.debug_camera)

    from py_numpy_renderer_amd._native import fill_frame_desc
    from py_numpy_renderer_amd._pack import pack_frame
    from py_numpy_renderer_amd.multigpu import BandRenderer

    partition = args.partition if world > 1 else "bands"
    if partition == "bands" and H % world:
        partition = "stripes"

    def renderer(in_flight, light):
        # one frame in 13 carries HIP event marks (frame + tile kernel, or every stage with --all-marks):
        # a mark costs ~5 us between two kernels; 13 is coprime to the number of views and streams
        scene.camera, scene.debug_camera = base_cameras
        return BandRenderer(scene, rank, world, shadows=shadows, light_timing=light,
                            frames_in_flight=in_flight, partition=partition, timing_every=13)

    br = renderer(args.frames_in_flight, not args.all_marks)

    # the views of the timed sequence: descriptors for this rank's tiles (timed frames: the frame only,
    # like Scene.render()) and the reference-equivalent fragment counts of each whole frame
    def descriptors(counters, light):
        out = []
        for cam, dbg in views:
            scene.camera, scene.debug_camera = cam, dbg
            out.append(fill_frame_desc(pack_frame(scene, shadows), br.band, light_timing=light, counters=counters,
                                       stripe=br.stripe))
        scene.camera, scene.debug_camera = base_cameras
        return out

    views = swing_cameras(api, scene, N_VIEWS)
    frags_view, unique_view, stats_view = [], [], []
    for cam, dbg in views:
        scene.camera, scene.debug_camera = cam, dbg
        backend.render(scene, shadows=shadows)                       # whole frame, counted
        st = dict(backend.last_stats)
        stats_view.append(st)
        frags_view.append(2 * st["frag_tri"] + st["frag_quad"])
        unique_view.append(st["frag_tri"] + st["frag_quad"])
    scene.camera, scene.debug_camera = base_cameras
    backend.render(scene, shadows=shadows)
    full = dict(backend.last_stats)                                  # the base view = BASELINE.md's frame
    frags_base = 2 * full["frag_tri"] + full["frag_quad"]
    # this rank's share of the base view, counted (for the roofline's algorithmic bytes)
    backend.render(scene, shadows=shadows, row_band=br.band if br.stripe is None else None, stripe=br.stripe)
    part_stats = dict(backend.last_stats)
    part_px = W * (br.band[1] - br.band[0]) if br.stripe is None else sum(
        W * min(16, H - 16 * g) for g in range(-(-H // 16)) if g % world == rank)

    br.set_descriptors(descriptors(False, not args.all_marks))
    fps = max(1, args.frames_per_step)

    def run(brx, n_frames):
        for _ in range(n_frames):
            brx.step()

    run(br, args.warmup * fps)
    if not br.verify():                  # a work list overflowed during warm-up: it has been grown, warm up again
        run(br, args.warmup * fps)
        assert br.verify(), "work lists kept overflowing"
    # A step is a batch of frames, sized so that the K timed steps last at least ~3 s whatever the config and
    # however fast a frame has become (long enough for an outside observer sampling the device once a second to
    # land inside the timed region): the rate of a short calibration run (untimed as far as the result goes)
    # decides, the same on every rank.
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(br, 256)
    torch.cuda.synchronize()
    est = torch.tensor([(time.perf_counter() - t0) / 256], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(est, op=dist.ReduceOp.MAX)
    need = int(TIMED_SECONDS / (max(args.steps, 1) * max(float(est.item()), 1e-6))) + 1
    fps = max(fps, -(-need // 8) * 8)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(br, args.steps * fps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_frames = args.steps * fps
    assert br.verify(), "a work list overflowed in the timed frames"
    ktimes, n_avg = br.kernel_times(2048)            # every marked frame still in the streams' event rings (~40 each)
    last_frame = br.frame

    # the frame every rank now holds must be the frame a single device renders for that view
    view_of_last = (br.count - 1) % N_VIEWS
    if rank == 0:
        scene.camera, scene.debug_camera = views[view_of_last]
        want = backend.render(scene, shadows=shadows)
        scene.camera, scene.debug_camera = base_cameras
        assert np.array_equal(last_frame.cpu().numpy(), want), "assembled frame differs from the single-device frame"
    total_frags = sum(frags_view[i % N_VIEWS] for i in range(n_frames))
    total_unique = sum(unique_view[i % N_VIEWS] for i in range(n_frames))

    # ---- the other regimes, outside the timed region (fewer frames each)
    def timed(brx, n):
        run(brx, max(8, n // 8))
        brx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        run(brx, n)
        brx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        return (time.perf_counter() - t0) / n

    n_side = max(64, min(n_frames // 4, 1024))
    solo = renderer(1, False)                                        # one frame at a time, every stage marked
    solo.set_descriptors(descriptors(False, False))
    per_frame_solo = timed(solo, n_side)
    ktimes_solo, _ = solo.kernel_times(64)
    counted = renderer(args.frames_in_flight, True)                  # fragment counters on: no depth cull of quads
    counted.set_descriptors(descriptors(True, True))
    per_frame_counted = timed(counted, n_side)

    host_ms = host_overlay_ms = host_same_ms = host_overlay_same_ms = pipe_ms = pipe_overlay_ms = None
    protocol_ms = protocol_copy_ms = None
    if rank == 0 and world == 1:
        # the drop-in call as a user makes it: Scene.render() returning the uint8 ndarray (packing of the
        # per-frame constants, upload, the three kernels, device->host copy into page-locked memory), first
        # the way the hot path is defined (debug-frustum overlay off, like the parity captures), then with
        # upstream's default (overlay on: obj/core.py:638)
        scene.camera, scene.debug_camera = base_cameras

        def host_median(moving, n=15):
            # moving: new Camera objects for every call, as in a sequence whose camera moves -- nothing the host
            # derives from the cameras (matrices, planes, the overlay's line lists) can come from a cache
            def step(k):
                if moving:
                    scene.camera, scene.debug_camera = swing_cameras(api, scene_base, 1, offset=k)[0]
                return scene.render(shadows=shadows)
            for k in range(3):
                step(1000 + k)
            samples = []
            for k in range(n):
                if moving:
                    scene.camera, scene.debug_camera = base_cameras
                t0 = time.perf_counter()
                step(k + 1)
                samples.append(time.perf_counter() - t0)
            scene.camera, scene.debug_camera = base_cameras
            return sorted(samples)[len(samples) // 2] * 1e3

        def pipelined(n=48, depth=3):
            # the same frames as a sequence: Scene.render_frames keeps `depth` frames in flight (mr_render_async), so the
            # host copy of one frame runs beside the kernels of the next and the host's preparation of the one after
            cams = lambda off: (swing_cameras(api, scene_base, 1, offset=off + k)[0] for k in range(n))
            for _ in scene.render_frames((swing_cameras(api, scene_base, 1, offset=500 + k)[0] for k in range(6)), shadows=shadows, depth=depth):
                pass
            t0 = time.perf_counter()
            for _ in scene.render_frames(cams(0), shadows=shadows, depth=depth):
                pass
            dt = (time.perf_counter() - t0) / n
            scene.camera, scene.debug_camera = base_cameras
            return dt * 1e3

        scene_base = scene
        scene.draw_debug_frustum = False
        host_ms = host_median(True)
        host_same_ms = host_median(False)
        pipe_ms = pipelined()
        scene.draw_debug_frustum = True
        host_overlay_ms = host_median(True)
        host_overlay_same_ms = host_median(False)
        pipe_overlay_ms = pipelined()
        scene.draw_debug_frustum = False
        # SURVEY 8(d)'s protocol: one frame at a time, HIP events around the whole device section of mr_render
        # (per-frame constants, the three kernels, the copy of the uint8 frame into page-locked host memory), median
        spans = []
        for k in range(40):
            scene.camera, scene.debug_camera = swing_cameras(api, scene_base, 1, offset=k)[0]
            backend.render(scene, shadows=shadows, counters=False, keep_buffers=False, timing=True)
            st = backend.stats()
            if k >= 8:
                spans.append((st["gpu_ms_total"], st["gpu_ms_copy"]))
        scene.camera, scene.debug_camera = base_cameras
        spans.sort()
        protocol_ms, protocol_copy_ms = spans[len(spans) // 2]

    if rank == 0:
        per_frame = elapsed / n_frames
        alg_full = algorithmic_bytes(full, W * H, n_faces, skybox)
        alg = algorithmic_bytes(part_stats, part_px, n_faces, skybox)
        # k_tile does the reference's three loops (coverage + z, stencil, shading + finalise); the
        # per-primitive attribute reads belong to k_setup
        tile_alg = alg["visibility"] + alg["shading"]
        k_ms = ktimes["tile"]
        achieved = tile_alg / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r03_traffic.json")
        if not os.path.exists(tpath):
            tpath = os.path.join(ROOT, "profiles", "r02_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as fh:
                traffic = json.load(fh).get(args.config, {}).get("k_tile")
        mean_frags = total_frags / n_frames
        line = {
            "metric": "Mfrag/s (1920x1080, 200k tris + shadow volumes)" if args.config == "c4"
                      else f"Mfrag/s ({label.split(' (BASELINE')[0]})",
            "value": round(total_frags / elapsed / 1e6, 2),
            "unit": "Mfrag/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "ms_per_frame": round(per_frame * 1e3, 5),
            "frames_per_s": round(n_frames / elapsed, 2),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": label, "name": args.config, "frames_per_step": fps, "timed_frames": n_frames,
                       "timed_seconds": round(elapsed, 4),
                       "fragments_per_frame": frags_base, "fragments_per_frame_mean_over_views": round(mean_frags, 1),
                       "unique_fragments_per_frame": full["frag_tri"] + full["frag_quad"],
                       "faces": n_faces, "views": N_VIEWS, "frames_in_flight": args.frames_in_flight,
                       "regime": "frames of a camera swing (8 views, 0.05 deg apart), per-frame constants new each frame, "
                                 "frame left in HBM, fragment counters off",
                       "parallelism": (f"screen tiles x{world} ({partition})" + (" + 1 RCCL all-gather" if world > 1 else ""))},
            "mfrag_unique_per_s": round(total_unique / elapsed / 1e6, 2),
            "latency_ms_single": round(per_frame_solo * 1e3, 5),
            "latency_ms_single_device": round(ktimes_solo["frame"], 5),
            "value_single_frame": round(mean_frags / per_frame_solo / 1e6, 2),
            "value_counters_on": round(mean_frags / per_frame_counted / 1e6, 2),
            "scene_render_ms_host": None if host_ms is None else round(host_ms, 4),
            "value_scene_render_host": None if host_ms is None else round(frags_base / host_ms / 1e3, 2),
            "scene_render_ms_host_with_overlay": None if host_overlay_ms is None else round(host_overlay_ms, 4),
            "scene_render_ms_host_same_camera": None if host_same_ms is None else round(host_same_ms, 4),
            "scene_render_ms_host_with_overlay_same_camera": None if host_overlay_same_ms is None else round(host_overlay_same_ms, 4),
            "scene_render_ms_pipelined": None if pipe_ms is None else round(pipe_ms, 4),
            "scene_render_ms_pipelined_with_overlay": None if pipe_overlay_ms is None else round(pipe_overlay_ms, 4),
            "value_protocol": None if not protocol_ms else round(frags_base / protocol_ms / 1e3, 2),
            "protocol": None if not protocol_ms else {
                "what": "SURVEY 8(d): one frame at a time, HIP events around mr_render's device section (three kernels + the "
                        "copy of the uint8 frame into page-locked host memory), median of 32 frames of a moving camera",
                "ms_per_frame": round(protocol_ms, 5), "ms_copy": round(protocol_copy_ms, 5),
                "frames_per_s": round(1e3 / protocol_ms, 1)},
            "reference_numpy_mfrag_s": REFERENCE_MFRAGS[args.config],
            # unless --all-marks, the timed region marks only frame start / tile start / tile end (a mark costs ~5 us
            # between two kernels) and the library files the span ahead of k_tile under its third slot
            "gpu_ms_per_kernel": ({k: round(v, 5) for k, v in ktimes.items()} if args.all_marks else
                                  {"setup_and_bin_work": round(ktimes["bin_work"], 5), "tile": round(ktimes["tile"], 5),
                                   "frame": round(ktimes["frame"], 5)}),
            "gpu_ms_per_kernel_solo": {k: round(v, 5) for k, v in ktimes_solo.items()},
            "frame_algorithmic_gb": round(alg_full["total"] / 1e9, 4),
            "frame_hbm_frac": round(alg_full["total"] / per_frame / 1e9 / HBM_PEAK_GBS, 5),
            "frame_hbm_frac_single": round(alg_full["total"] / per_frame_solo / 1e9 / HBM_PEAK_GBS, 5),
            "roofline": {"bound": "hbm", "kernel": "k_tile", "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": traffic, "algorithmic_bytes_per_launch": int(tile_alg),
                         "avg_launch_ms": round(k_ms, 5), "launches_averaged": n_avg,
                         "frames_in_flight": args.frames_in_flight,
                         "note": "avg_launch_ms is the HIP-event span of k_tile in the timed region, where it shares the "
                                 "device with the kernels of the other frames in flight; solo_* is the same kernel with one "
                                 "frame at a time, right after the timed region.  The kernel's arithmetic is float64 on the "
                                 "vector pipe (the reference's dtype); DESIGN.md section 5 has its VALU-side bound next to this one",
                         "solo_launch_ms": round(ktimes_solo["tile"], 5),
                         "solo_achieved": round(tile_alg / (ktimes_solo["tile"] * 1e-3) / 1e9, 2),
                         "solo_frac": round(tile_alg / (ktimes_solo["tile"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
            "roofline_valu": valu_roofline(args.config, k_ms, ktimes_solo["tile"]),
        }
        if world == 1 and not args.no_cpu_baseline:
            scene.camera, scene.debug_camera = base_cameras
            line["cpu_baseline"] = cpu_baseline(scene, frags_base, args.config,
                                                budget_s=25.0 if args.config == "c5" else 12.0)
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
