"""GPU box: wall time of Scene.render() when the camera MOVES every call (new Camera objects: nothing cached
on the host), against the same call repeated with unchanged cameras; overlay off and on (upstream's default)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import scenes
api = scenes.product_api()
name = sys.argv[1] if len(sys.argv) > 1 else "c4_torus200k_1080p"
sc = scenes.build(api, name)
shadows = name not in scenes.NO_SHADOW
cam = sc.camera
kw = dict(fovy=cam.fovy, near=cam.near, far=cam.far, backface_culling=cam.backface_culling, up=cam.up,
          projection_type=cam.projection_type)
base = np.asarray(cam.position, dtype=np.float64)


def view(k):
    a = np.deg2rad(k * 0.05)
    pos = (base[0] * np.cos(a) + base[2] * np.sin(a), base[1], -base[0] * np.sin(a) + base[2] * np.cos(a))
    return api.Camera(pos, cam.center, **kw), api.Camera(pos, cam.center, **kw)


for overlay in (False, True):
    sc.draw_debug_frustum = overlay
    for moving in (False, True):
        for k in range(3):
            if moving:
                sc.camera, sc.debug_camera = view(100 + k)
            sc.render(shadows=shadows)
        ts = []
        for k in range(15):
            if moving:
                sc.camera, sc.debug_camera = view(k)
            t0 = time.perf_counter()
            sc.render(shadows=shadows)
            ts.append(time.perf_counter() - t0)
        print(name, "overlay" if overlay else "no overlay", "moving camera" if moving else "same camera",
              "median ms", round(sorted(ts)[len(ts) // 2] * 1e3, 3), flush=True)
