"""GPU box: where the wall time of Scene.render() goes on the host, call by call (moving camera: new Camera objects
every frame), overlay off and on.  Wraps the binding's steps with timers; prints medians in microseconds."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import scenes
from py_numpy_renderer_amd import _native
api = scenes.product_api()
name = sys.argv[1] if len(sys.argv) > 1 else "c4_torus200k_1080p"
sc = scenes.build(api, name)
shadows = name not in scenes.NO_SHADOW
cam = sc.camera
kw = dict(fovy=cam.fovy, near=cam.near, far=cam.far, backface_culling=cam.backface_culling, up=cam.up,
          projection_type=cam.projection_type)
base = np.asarray(cam.position, dtype=np.float64)
times = collections.defaultdict(list)


def timed(obj, attr, label):
    fn = getattr(obj, attr)
    def wrap(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            times[label].append(time.perf_counter() - t0)
    setattr(obj, attr, wrap)


be = sc._backend()
for attr in ("sync_scene", "sync_skybox", "packed_frame", "sync_overlay"):
    timed(be, attr, attr)
timed(be.lib, "mr_render", "mr_render")
timed(be.lib, "mr_scene_set_overlay_cameras", "  set_overlay_cameras (C)")
timed(_native, "fill_frame_desc", "fill_frame_desc")
timed(be._pinned, "array", "pinned array")


def view(k):
    a = np.deg2rad(k * 0.05)
    pos = (base[0] * np.cos(a) + base[2] * np.sin(a), base[1], -base[0] * np.sin(a) + base[2] * np.cos(a))
    return api.Camera(pos, cam.center, **kw), api.Camera(pos, cam.center, **kw)


for overlay in (False, True):
    sc.draw_debug_frustum = overlay
    for k in range(4):
        sc.camera, sc.debug_camera = view(100 + k)
        sc.render(shadows=shadows)
    times.clear()
    for k in range(25):
        t0 = time.perf_counter()
        sc.camera, sc.debug_camera = view(k)
        times["new cameras"].append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        sc.render(shadows=shadows)
        times["Scene.render() total"].append(time.perf_counter() - t0)
    print(name, "overlay" if overlay else "no overlay", {k: round(sorted(v)[len(v) // 2] * 1e6, 1) for k, v in times.items()}, flush=True)
    st = be.stats()
    print("   last frame gpu ms:", {k: round(v, 4) for k, v in st.items() if k.startswith("gpu_ms")})

# the pipelined call: frames of a moving camera through Scene.render_frames (2 and 3 in flight)
for overlay in (False, True):
    sc.draw_debug_frustum = overlay
    for depth in (1, 2, 3):
        n = 60
        list(sc.render_frames((view(200 + k) for k in range(6)), shadows=shadows, depth=depth))
        t0 = time.perf_counter()
        for _ in sc.render_frames((view(k) for k in range(n)), shadows=shadows, depth=depth):
            pass
        print(name, "overlay" if overlay else "no overlay", "render_frames depth", depth, "us/frame",
              round((time.perf_counter() - t0) / n * 1e6, 1), flush=True)
