"""Generates the golden vectors under tests/golden/ by running the REFERENCE renderer.

Run in the build container only (needs /root/reference; it does not exist on the GPU box):

    python tests/golden/make_golden.py [--full] [names...]

The reference is imported unmodified.  Harness-side arrangements (none touch its files):
  * a stub ``numba`` module (imported at obj/triangular.py:3, never used, not installed);
  * module globals wrapped to observe intermediates: ``core.rasterize`` (face counter, pass
    transition, per-face status), ``triangular.general_shading`` (who wrote each pixel in
    pass 1), ``triangular.barycentric`` / ``triangular.quad_test`` (fragment counts),
    ``core.resterize_quadrangle`` (quad count), ``core.draw_view_frustum`` (float frame
    before the overlay; the overlay itself is skipped unless asked for);
  * "no shadows" scenes replace ``core.shadow_volumes`` with a no-op (the reference has no
    switch: Scene(shadows=...) is dead, obj/core.py:568).
Only data (inputs and the reference's outputs) is written; no reference source is copied.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.dont_write_bytecode = True

REF = "/root/reference"


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("the reference is not mounted here; golden vectors can only be "
                         "regenerated in the build container")
    sys.path[:0] = [REF, os.path.join(REF, "obj")]
    stub = types.ModuleType("numba")
    stub.jit = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda fn: fn))
    sys.modules.setdefault("numba", stub)
    warnings.simplefilter("ignore")
    import core
    import triangular
    import transformation
    from obj.lightning import Lightning      # the module object triangular.py compares against (as obj/main.py:9)
    from obj.cube_map import CubeMap         # the class core.py's isinstance() test refers to (obj/core.py:8)
    api = types.SimpleNamespace(
        Model=core.Model, Camera=core.Camera, Light=core.Light, Scene=core.Scene, Lightning=Lightning, CubeMap=CubeMap,
        SYSTEM=transformation.SYSTEM, SUBSYSTEM=transformation.SUBSYSTEM,
        PROJECTION_TYPE=transformation.PROJECTION_TYPE, scale=transformation.scale,
        translation=transformation.translation, rotate_xyz=transformation.rotate_xyz)
    return api, core, triangular


class Capture:
    """Wraps the reference's module globals for one render and collects intermediates."""

    def __init__(self, core, triangular, shadows=True, overlay=False):
        self.core, self.tri = core, triangular
        self.shadows, self.overlay = shadows, overlay
        self.saved = {}

    def __enter__(self):
        core, tri = self.core, self.tri
        self.saved = {(core, n): getattr(core, n) for n in
                      ("rasterize", "resterize_quadrangle", "draw_view_frustum", "shadow_volumes")}
        self.saved.update({(tri, n): getattr(tri, n) for n in ("general_shading", "barycentric", "quad_test")})
        st = self.state = types.SimpleNamespace(
            calls=0, n_faces=None, pass2=False, winner=None, z1=None, stencil=None, frame1=None,
            frame=None, status=[], frag_tri=[0, 0], frag_quad=0, n_quads=0, shaded=[0, 0],
            bbox_tri=0, bbox_quad=0, z=None)
        orig = {n: f for (_, n), f in self.saved.items()}

        def rasterize(face, frame, z_buffer, light, camera, stencil_buffer=None, debug_camera=None):
            if stencil_buffer is not None and not st.pass2:
                st.pass2 = True
                st.n_faces = st.calls
                st.calls = 0
                st.z1, st.stencil, st.frame1 = z_buffer.copy(), stencil_buffer.copy(), frame.copy()
            if st.winner is None:
                st.winner = np.full(z_buffer.shape, -1, np.int32)
            st.current = st.calls
            rc = orig["rasterize"](face, frame, z_buffer, light, camera, stencil_buffer, debug_camera)
            if st.pass2:
                st.status.append(int(rc.value) if rc else 0)
            st.calls += 1
            return rc

        def general_shading(face, bar, light, camera, frame, x, y, first_pass):
            if first_pass:
                st.winner[x, y] = st.current
            st.shaded[0 if first_pass else 1] += len(x)
            return orig["general_shading"](face, bar, light, camera, frame, x, y, first_pass)

        def barycentric(a, b, c, p):
            out = orig["barycentric"](a, b, c, p)
            st.bbox_tri += len(p) if not st.pass2 else 0
            if out is not None:
                st.frag_tri[1 if st.pass2 else 0] += int((out >= 0).all(axis=1).sum())
            return out

        def quad_test(points, polygon, callback):
            out = orig["quad_test"](points, polygon, callback)
            st.frag_quad += int(out.sum())
            st.bbox_quad += len(points)
            return out

        def resterize_quadrangle(*a, **k):
            st.n_quads += 1
            return orig["resterize_quadrangle"](*a, **k)

        def draw_view_frustum(frame, camera, positioned_object, z_buffer, sign):
            st.frame = frame.copy()
            st.z = z_buffer.copy()
            if self.overlay:
                orig["draw_view_frustum"](frame, camera, positioned_object, z_buffer, sign)
                st.frame_overlay = frame.copy()
                st.z_overlay = z_buffer.copy()

        core.rasterize, core.resterize_quadrangle = rasterize, resterize_quadrangle
        core.draw_view_frustum = draw_view_frustum
        tri.general_shading, tri.barycentric, tri.quad_test = general_shading, barycentric, quad_test
        if not self.shadows:
            core.shadow_volumes = lambda *a, **k: None
        return st

    def __exit__(self, *exc):
        for (mod, name), fn in self.saved.items():
            setattr(mod, name, fn)


def render_reference(api, core, triangular, name, shadows=True, overlay=False):
    import scenes
    scene = scenes.build(api, name)
    log = io.StringIO()
    t0 = time.time()
    with Capture(core, triangular, shadows, overlay) as st, contextlib.redirect_stdout(log):
        out = scene.render()
    seconds = time.time() - t0
    sil = []
    for mi, model in enumerate(scene.models):
        sil += [(mi, int(e[0]), int(e[1])) for e in model.silhouette]
    cam, dbg = scene.camera, scene.debug_camera
    host = dict(mvp=cam.MVP, viewport=cam.viewport, debug_mvp=dbg.MVP, planes=cam.frustum_planes,
                light_dir=np.asarray(scene.light.direction, dtype=np.float64))
    counts = dict(frag_tri_pass1=st.frag_tri[0], frag_tri_pass2=st.frag_tri[1], frag_quad=st.frag_quad,
                  shaded_pass1=st.shaded[0], shaded_pass2=st.shaded[1], bbox_px_tri=st.bbox_tri,
                  bbox_px_quad=st.bbox_quad, n_quads=len(sil), n_quads_drawn=None,
                  n_faces=int(st.n_faces), render_seconds=round(seconds, 3))
    return types.SimpleNamespace(out=out, frame=st.frame, frame1=st.frame1, z=st.z1, z_final=st.z,
                                 frame_overlay=getattr(st, "frame_overlay", None), z_overlay=getattr(st, "z_overlay", None),
                                 stencil=st.stencil, winner=st.winner,
                                 status=np.array(st.status, np.uint8), silhouette=np.array(sil, np.int32).reshape(-1, 3),
                                 host=host, counts=counts, stdout=log.getvalue())


def z_row_sums(z):
    return z.view(np.uint64).sum(axis=1, dtype=np.uint64)


def save_small(name, r):
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"),
                        out=r.out, frame=r.frame, z=r.z, stencil=r.stencil, winner=r.winner,
                        face_status=r.status, silhouette=r.silhouette, **{f"host_{k}": v for k, v in r.host.items()})
    with open(os.path.join(HERE, f"{name}.json"), "w") as fh:
        json.dump(dict(counts=r.counts, stdout=r.stdout), fh, indent=1)


def save_overlay(name, r):
    """Overlay-on variant: the uint8 frame, float frame and z-buffer AFTER obj/core.py:638."""
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), out=r.out, frame_overlay=r.frame_overlay,
                        z_overlay=r.z_overlay)


def save_full(name, r):
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"),
                        out=r.out, stencil=r.stencil, winner=r.winner, z_row_sums=z_row_sums(r.z),
                        face_status=r.status, silhouette=r.silhouette, **{f"host_{k}": v for k, v in r.host.items()})
    with open(os.path.join(HERE, f"{name}.json"), "w") as fh:
        json.dump(dict(counts=r.counts, stdout=r.stdout), fh, indent=1)


def texture_digest(arr):
    """What pins a loaded texture without storing it: shape, dtype, tangent flag, exact sum of the
    float32 bit patterns and a few texels."""
    arr = np.asarray(arr)
    meta = arr.dtype.metadata or {}
    flat = np.ascontiguousarray(arr).reshape(-1)
    return dict(shape=list(arr.shape), dtype=arr.dtype.name, tangent=meta.get("tangent"),
                bits_sum=int(flat.view(np.uint32).sum(dtype=np.uint64)) if arr.dtype == np.float32 else None,
                probe=[float(v) for v in flat[:: max(1, flat.size // 7)][:8]])


def material_record(mat):
    """Every attribute the loader set on a Material (instance __dict__): scalars and small arrays by
    value (with dtype), texture maps by digest."""
    rec = {}
    for key, val in sorted(vars(mat).items()):
        if isinstance(val, np.ndarray) and val.ndim == 3:
            rec[key] = dict(kind="texture", **texture_digest(val))
        elif isinstance(val, np.ndarray):
            rec[key] = dict(kind="array", dtype=val.dtype.name, value=[float(v) for v in val.ravel()])
        else:
            rec[key] = dict(kind=type(val).__name__, value=val)
    return rec


def loader_kat(api):
    """Model.load_model / parse_mtl / TextureMaps.register of the reference on the synthetic files of
    scenes.kat_files() and on cube.obj + cube.mtl (obj/core.py:72-105,257-348; obj/materials.py:57-77)."""
    import scenes
    files = dict(scenes.kat_files())
    files["cube"] = os.path.join(scenes.ASSETS, "cube", "cube.obj")
    arrays, meta = {}, {}
    for key, path in files.items():
        log = io.StringIO()
        with contextlib.redirect_stdout(log):
            m = api.Model.load_model(path)
        for name in ("vertices", "uv", "normals", "_faces"):
            val = getattr(m, name)
            if val is not None:
                arrays[f"{key}.{name}"] = np.asarray(val)
        meta[key] = dict(
            stdout=log.getvalue().replace(scenes.GENERATED, "<generated>"),
            dtypes={n: (None if getattr(m, n) is None else np.asarray(getattr(m, n)).dtype.name)
                    for n in ("vertices", "uv", "normals", "_faces")},
            material_group=list(m.material_group),
            materials={name: material_record(mat) for name, mat in m.materials.items()})
    # TextureMaps.register: default normalize (x*2-1) with and without the tangent flag, and a colour map
    m = api.Model.load_model(files["cube"])
    tex = os.path.join(scenes.ASSETS, "floor_nm_tangent.tga")
    m.textures.register("normals", tex, tangent=True)
    m.textures.register("diffuse", os.path.join(scenes.ASSETS, "floor_diffuse.tga"), normalize=False)
    m.textures.register("specular", tex)
    meta["register"] = material_record(m.materials["default"])
    np.savez_compressed(os.path.join(HERE, "loader_kat.npz"), **arrays)
    with open(os.path.join(HERE, "loader_kat.json"), "w") as fh:
        json.dump(meta, fh, indent=1)
    print("loader_kat:", {k: v.shape for k, v in arrays.items()}, flush=True)


def main():
    import scenes
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*")
    ap.add_argument("--full", action="store_true", help="also the 1080p BASELINE configs (minutes)")
    ap.add_argument("--loader", action="store_true", help="only the loader known-answer fixture")
    args = ap.parse_args()
    api, core, triangular = import_reference()
    if args.loader:
        loader_kat(api)
        return
    small = list(scenes.SMALL) + ["diablo_small_noshadow"] + list(scenes.OVERLAY)
    names = args.names or (small + (list(scenes.FULL) if args.full else []))
    for name in names:
        shadows = name not in scenes.NO_SHADOW
        if name.endswith("_overlay"):
            r = render_reference(api, core, triangular, name[:-len("_overlay")], shadows=shadows, overlay=True)
            save_overlay(name, r)
        else:
            r = render_reference(api, core, triangular, name, shadows=shadows)
            (save_full if name in scenes.FULL or name in scenes.HUGE else save_small)(name, r)
        print(f"{name}: {r.counts}", flush=True)


if __name__ == "__main__":
    main()
