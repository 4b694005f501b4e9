"""Scene recipes shared by the golden-vector generator and the tests.

Every recipe is written against an ``api`` namespace exposing the reference's public names
(``Model, Camera, Light, Scene, Lightning, SYSTEM, SUBSYSTEM, scale, translation,
rotate_xyz``).  ``tests/golden/make_golden.py`` passes the reference's own modules (build
container only); the tests pass ``py_numpy_renderer_amd``.  Both therefore construct the
same scenes from the same files, which is what makes the captured buffers golden vectors.

Recipes follow SURVEY.md section 8(d): camera (0.5,1,2)->origin fovy 60 near 0.1 far 20, a debug
camera with identical arguments, point light (2,3,4) ambient 0.1 specular 0.1, RH/DirectX.
"""
import math
import os
import tempfile
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ASSETS = os.path.join(HERE, "assets")
GENERATED = os.path.join(tempfile.gettempdir(), "mi355rast_generated")


def product_api():
    """The product's host API as a recipe namespace."""
    import py_numpy_renderer_amd as pkg
    from py_numpy_renderer_amd import transformation as tr
    return SimpleNamespace(Model=pkg.Model, Camera=pkg.Camera, Light=pkg.Light, Scene=pkg.Scene,
                           Lightning=pkg.Lightning, SYSTEM=pkg.SYSTEM, SUBSYSTEM=pkg.SUBSYSTEM, CubeMap=pkg.CubeMap,
                           PROJECTION_TYPE=pkg.PROJECTION_TYPE, scale=tr.scale, translation=tr.translation, rotate_xyz=tr.rotate_xyz)


# --------------------------------------------------------------------------- generated meshes
def _write_if_changed(path, text):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    if os.path.exists(path):
        with open(path) as fh:
            if fh.read() == text:
                return path
    tmp = f"{path}.{os.getpid()}.tmp"
    with open(tmp, "w") as fh:
        fh.write(text)
    os.replace(tmp, path)
    return path


def floor_obj():
    """Two triangles (+-2, -1, +-2), normal +y (upstream's floor.obj is not shipped)."""
    text = ("v -2 -1 -2\nv 2 -1 -2\nv 2 -1 2\nv -2 -1 2\n"
            "vt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvn 0 1 0\n"
            "f 1/1/1 3/3/1 2/2/1\nf 1/1/1 4/4/1 3/3/1\n")
    return _write_if_changed(os.path.join(GENERATED, "floor.obj"), text)


def torus_obj(nu, nv, R=0.6, r=0.25, tilt_deg=35.0):
    """Torus with nu x nv cells (2 triangles each), tilted about x; seam duplicated in vt only."""
    path = os.path.join(GENERATED, f"torus_{nu}x{nv}.obj")
    if os.path.exists(path):
        return path
    i = np.arange(nu)[:, None]
    j = np.arange(nv)[None, :]
    u = 2 * np.pi * i / nu
    v = 2 * np.pi * j / nv
    ct, st = math.cos(math.radians(tilt_deg)), math.sin(math.radians(tilt_deg))

    def tilt(x, y, z):
        return x, ct * y - st * z, st * y + ct * z

    px, py, pz = tilt((R + r * np.cos(v)) * np.cos(u), r * np.sin(v) + 0 * u, (R + r * np.cos(v)) * np.sin(u))
    nx, ny, nz = tilt(np.cos(v) * np.cos(u), np.sin(v) + 0 * u, np.cos(v) * np.sin(u))
    lines = []
    for a, b, c in zip(px.ravel(), py.ravel(), pz.ravel()):
        lines.append("v %.6f %.6f %.6f" % (a, b, c))
    for ii in range(nu + 1):
        for jj in range(nv + 1):
            lines.append("vt %.6f %.6f" % (ii / nu, jj / nv))
    for a, b, c in zip(nx.ravel(), ny.ravel(), nz.ravel()):
        lines.append("vn %.6f %.6f %.6f" % (a, b, c))

    def vid(ii, jj):
        return (ii % nu) * nv + (jj % nv) + 1

    def tid(ii, jj):
        return ii * (nv + 1) + jj + 1

    for ii in range(nu):
        for jj in range(nv):
            a = (ii, jj); b = (ii + 1, jj); c = (ii + 1, jj + 1); d = (ii, jj + 1)
            for tri in ((a, c, b), (a, d, c)):
                lines.append("f " + " ".join("%d/%d/%d" % (vid(*p), tid(*p), vid(*p)) for p in tri))
    return _write_if_changed(path, "\n".join(lines) + "\n")


def bare_tetra_obj():
    """Tetrahedron with uv but no normals (``v/vt/``): exercises the face-normal shading path."""
    text = ("v 0 0.6 0\nv -0.5 -0.3 0.4\nv 0.5 -0.3 0.4\nv 0 -0.3 -0.5\n"
            "vt 0 0\nvt 1 0\nvt 0.5 1\n"
            "f 1/1/ 2/2/ 3/3/\nf 1/1/ 3/2/ 4/3/\nf 1/1/ 4/2/ 2/3/\nf 2/1/ 4/2/ 3/3/\n")
    return _write_if_changed(os.path.join(GENERATED, "bare_tetra.obj"), text)


def kat_files():
    """Synthetic OBJ/MTL pair for the loader known-answer test that is also
    renderable: negative (relative) indices, quads and a 5-gon (fan triangulation), three
    ``usemtl`` groups (one of them never defined in the library -> falls back to 'default'),
    a fractional ``Ns``, ``map_Kd`` + ``map_bump`` (tangent-space ``norm``) and a texture file
    that does not exist (the loader prints a hint and goes on).  The textures are copies of the
    reference's floor textures, placed next to the library as ``.mtl`` paths are relative."""
    import shutil
    os.makedirs(GENERATED, exist_ok=True)
    for src, dst in (("floor_diffuse.tga", "kat_diffuse.tga"), ("floor_nm_tangent.tga", "kat_bump.tga")):
        if not os.path.exists(os.path.join(GENERATED, dst)):
            shutil.copy(os.path.join(ASSETS, src), os.path.join(GENERATED, dst))
    mtl = ("# loader known-answer library\n\n"
           "newmtl brick\nNs 17.3\nKa 0.1 0.1 0.1\nKd 0.7 0.35 0.2\nKs 0.5 0.5 0.5\nd 1.0\nillum 2\n"
           "map_Ks kat_missing.png\n\n"
           "newmtl tiles\nNs 40\nKd 0.5 0.5 0.5\nKs 0.25 0.5 0.75\nmap_Kd kat_diffuse.tga\nmap_bump kat_bump.tga\n")
    _write_if_changed(os.path.join(GENERATED, "kat.mtl"), mtl)
    # a house: square base (quad, 'tiles'), four walls (quads, 'brick'), a pentagonal gable written with
    # negative indices, and a roof triangle pair in a group the library does not define ('slate')
    obj = ("mtllib kat.mtl\n"
           "v -0.5 -0.4 -0.5\nv 0.5 -0.4 -0.5\nv 0.5 -0.4 0.5\nv -0.5 -0.4 0.5\n"
           "v -0.5 0.3 -0.5\nv 0.5 0.3 -0.5\nv 0.5 0.3 0.5\nv -0.5 0.3 0.5\n"
           "v 0 0.75 0.5 1.0\nv 0 0.75 -0.5\n"
           "vt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvt 0.5 1.4 0.0\n"
           "vn 0 -1 0\nvn 0 0 -1\nvn 1 0 0\nvn 0 0 1\nvn -1 0 0\nvn 0.7071 0.7071 0\nvn -0.7071 0.7071 0\n"
           "usemtl tiles\nf 1/1/1 2/2/1 3/3/1 4/4/1\n"
           "usemtl brick\n"
           "f 1/1/2 5/4/2 6/3/2 2/2/2\nf 2/1/3 6/4/3 7/3/3 3/2/3\nf 4/2/5 8/3/5 5/4/5 1/1/5\n"
           "f -7/1/4 -8/2/4 -4/3/4 -2/5/4 -3/4/4\n"
           "usemtl slate\n"
           "f 6/1/6 10/4/6 9/3/6 7/2/6\nf -6/2/-1 -3/1/-1 -2/4/-1 -1/3/-1\n"
           "usemtl brick\nf 5/1/2 10/5/2 6/2/2\n")
    _write_if_changed(os.path.join(GENERATED, "kat.obj"), obj)
    # no uv at all: v//vn corners (-1 in the uv column), with a quad and negative indices
    nouv = ("v -0.3 -0.4 0.9\nv 0.3 -0.4 0.9\nv 0.3 0.1 0.9\nv -0.3 0.1 0.9\nv 0 0.1 1.3\n"
            "vn 0 0 -1\nvn 0 1 0\nvn 0 0 1\n"
            "f 1//1 4//1 3//1 2//1\nf -1//2 -3//2 -2//2\nf 1//3 2//3 5//3\n")
    _write_if_changed(os.path.join(GENERATED, "kat_nouv.obj"), nouv)
    # shapes the reference parses but cannot render: v/vt corners and bare v corners
    _write_if_changed(os.path.join(GENERATED, "kat_v_vt.obj"),
                      "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nf 1/1 2/2 3/3 4/4\n")
    _write_if_changed(os.path.join(GENERATED, "kat_v.obj"), "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 -1\n")
    return {k: os.path.join(GENERATED, f"{k}.obj") for k in ("kat", "kat_nouv", "kat_v_vt", "kat_v")}


def gizmo_files():
    """``obj_loader_test/sphere.obj`` and ``obj_loader_test/camera.obj`` for the ``show=True`` gizmos
    (obj/core.py:532-552 loads them by these relative names; upstream does not ship them).  Returns the
    directory to run in.  Sphere: 10 x 6 lat-long mesh of radius 1; camera: a box body with a pyramid lens."""
    lines = []
    nu, nv = 10, 6
    for j in range(nv + 1):
        th = math.pi * j / nv
        for i in range(nu):
            ph = 2 * math.pi * i / nu
            lines.append("v %.6f %.6f %.6f" % (math.sin(th) * math.cos(ph), math.cos(th), math.sin(th) * math.sin(ph)))
    for j in range(nv + 1):
        for i in range(nu + 1):
            lines.append("vt %.6f %.6f" % (i / nu, 1 - j / nv))
    for j in range(nv + 1):
        th = math.pi * j / nv
        for i in range(nu):
            ph = 2 * math.pi * i / nu
            lines.append("vn %.6f %.6f %.6f" % (math.sin(th) * math.cos(ph), math.cos(th), math.sin(th) * math.sin(ph)))
    vid = lambda i, j: j * nu + (i % nu) + 1
    tid = lambda i, j: j * (nu + 1) + i + 1
    for j in range(nv):
        for i in range(nu):
            quad = ((i, j), (i + 1, j), (i + 1, j + 1), (i, j + 1))
            tris = ([quad[0], quad[1], quad[2]] if j > 0 else []), ([quad[0], quad[2], quad[3]] if j < nv - 1 else [])
            if j == 0:
                tris = ([quad[0], quad[2], quad[3]],)
            elif j == nv - 1:
                tris = ([quad[0], quad[1], quad[2]],)
            for tri in tris:
                if tri:
                    lines.append("f " + " ".join("%d/%d/%d" % (vid(a, b), tid(a, b), vid(a, b)) for a, b in tri))
    _write_if_changed(os.path.join(GENERATED, "obj_loader_test", "sphere.obj"), "\n".join(lines) + "\n")
    cam = ("v -0.6 -0.4 0\nv 0.6 -0.4 0\nv 0.6 0.4 0\nv -0.6 0.4 0\n"
           "v -0.6 -0.4 1.4\nv 0.6 -0.4 1.4\nv 0.6 0.4 1.4\nv -0.6 0.4 1.4\n"
           "v 0 0 0\nv -0.5 -0.35 -0.8\nv 0.5 -0.35 -0.8\nv 0.5 0.35 -0.8\nv -0.5 0.35 -0.8\n"
           "vt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\n"
           "vn 0 0 -1\nvn 0 0 1\nvn -1 0 0\nvn 1 0 0\nvn 0 -1 0\nvn 0 1 0\n"
           "f 1/1/1 4/4/1 3/3/1 2/2/1\nf 5/1/2 6/2/2 7/3/2 8/4/2\n"
           "f 1/1/3 5/2/3 8/3/3 4/4/3\nf 2/1/4 3/4/4 7/3/4 6/2/4\n"
           "f 1/1/5 2/2/5 6/3/5 5/4/5\nf 4/1/6 8/4/6 7/3/6 3/2/6\n"
           "f 9/1/5 11/3/5 10/2/5\nf 9/1/4 12/3/4 11/2/4\nf 9/1/6 13/3/6 12/2/6\nf 9/1/3 10/3/3 13/2/3\n")
    _write_if_changed(os.path.join(GENERATED, "obj_loader_test", "camera.obj"), cam)
    return GENERATED


def fins_obj():
    """A non-manifold "fin" mesh: four vertical spine edges, each shared by THREE or FOUR triangles that fan
    out from it at different angles and with mixed windings (obj/triangular.py:286-302 toggles an edge once per
    light-facing incident face: with three or four of them an edge is added, discarded and added again, and
    keeps the orientation of the LAST insert).  No closed surface anywhere: every outer edge has one face."""
    # x offset, z offset, [(fin angle in degrees, faces the light at (2, 3, 4)?)]: the winding of every fin is chosen
    # so that it does or does not face that light -- spine 1: one of three does (the edge is inserted once), spine 2:
    # three of four (inserted, discarded, inserted again), spine 3: two of four (inserted and discarded: not on the
    # silhouette), spine 4: all three (the surviving entry has the orientation of the third face)
    wanted = (
        (-0.75, 0.0, ((10, False), (130, True), (250, False))),
        (-0.25, 0.1, ((40, True), (100, True), (200, False), (320, True))),
        (0.25, -0.1, ((0, True), (90, False), (180, True), (270, False))),
        (0.75, 0.0, ((60, True), (180, True), (300, True))),
    )
    # the normal of (bottom, top, tip) is (sin a, 0, -cos a) up to scale
    spines = tuple((sx, sz, tuple((ang, 1 if ((2 * math.sin(math.radians(ang)) - 4 * math.cos(math.radians(ang))) > 0) == lit
                                   else -1) for ang, lit in fins)) for sx, sz, fins in wanted)
    verts, normals, faces = [], [], []
    for sx, sz, fins in spines:
        bottom, top = len(verts) + 1, len(verts) + 2
        verts += [(sx, -0.45, sz), (sx, 0.35, sz)]
        for ang, wind in fins:
            a = math.radians(ang)
            tip = (sx + 0.22 * math.cos(a), -0.05 + 0.002 * ang / 10.0, sz + 0.22 * math.sin(a))
            verts.append(tip)
            tri = (bottom, top, len(verts)) if wind > 0 else (top, bottom, len(verts))
            pa, pb, pc = (np.array(verts[i - 1]) for i in tri)
            n = np.cross(pb - pa, pc - pa)
            n = n / np.linalg.norm(n)
            normals.append(tuple(n))
            faces.append((tri, len(normals)))
    lines = ["v %.6f %.6f %.6f" % v for v in verts] + ["vn %.6f %.6f %.6f" % n for n in normals]
    lines += ["f " + " ".join("%d//%d" % (i, ni) for i in tri) for tri, ni in faces]
    return _write_if_changed(os.path.join(GENERATED, "fins.obj"), "\n".join(lines) + "\n")


def wall_files():
    """A 3 x 3 wall of quads, every quad in a ``usemtl`` group of its own: nine materials in one library
    (different Kd / Ks / Ns, whole and fractional exponents, two of them with a ``map_Kd``) -- with the floor's
    and the cube's that is more than the tile kernel keeps in LDS."""
    import shutil
    os.makedirs(GENERATED, exist_ok=True)
    for src, dst in (("grid.tga", "wall_grid.tga"), ("floor_diffuse.tga", "wall_floor.tga")):
        if not os.path.exists(os.path.join(GENERATED, dst)):
            shutil.copy(os.path.join(ASSETS, src), os.path.join(GENERATED, dst))
    mtl, obj = ["# nine materials"], ["mtllib wall.mtl"]
    for j in range(4):
        for i in range(4):
            obj.append("v %.6f %.6f %.6f" % (-0.9 + 0.6 * i, -0.5 + 0.45 * j, -0.3 + 0.05 * i))
    obj += ["vt 0 0", "vt 1 0", "vt 1 1", "vt 0 1", "vn 0 0 1"]
    for k in range(9):
        i, j = k % 3, k // 3
        mtl += ["", f"newmtl m{k}", "Ns %s" % (8 + 7 * k if k % 2 == 0 else 5.5 + 3.25 * k),
                "Kd %.3f %.3f %.3f" % (0.15 + 0.09 * k, 0.9 - 0.08 * k, 0.3 + 0.05 * ((k * 5) % 9)),
                "Ks %.3f %.3f %.3f" % (0.2 + 0.08 * k, 0.5, 1.0 - 0.1 * k)]
        if k == 4:
            mtl.append("map_Kd wall_grid.tga")
        if k == 7:
            mtl.append("map_Kd wall_floor.tga")
        a, b, c, d = j * 4 + i + 1, j * 4 + i + 2, (j + 1) * 4 + i + 2, (j + 1) * 4 + i + 1
        obj += [f"usemtl m{k}", f"f {a}/1/1 {b}/2/1 {c}/3/1 {d}/4/1"]
    _write_if_changed(os.path.join(GENERATED, "wall.mtl"), "\n".join(mtl) + "\n")
    return _write_if_changed(os.path.join(GENERATED, "wall.obj"), "\n".join(obj) + "\n")


def neg_uv_obj():
    """A tilted quad cut into a 3 x 3 grid whose texture coordinates run from -0.6 to 1.4 in both directions:
    negative ``vt`` truncate to negative texel indices, which Python wraps from the far side of the texture
    (obj/core.py:138-143); values above 1 are clipped (u) or go negative through ``1 - v`` (rows)."""
    lines = []
    for j in range(4):
        for i in range(4):
            lines.append("v %.6f %.6f %.6f" % (-0.8 + 1.6 * i / 3, -0.5 + 1.1 * j / 3, 0.2 - 0.25 * j / 3))
    for j in range(4):
        for i in range(4):
            lines.append("vt %.6f %.6f" % (-0.6 + 2.0 * i / 3, -0.6 + 2.0 * j / 3))
    lines.append("vn 0 0.2 1")
    for j in range(3):
        for i in range(3):
            a, b, c, d = j * 4 + i + 1, j * 4 + i + 2, (j + 1) * 4 + i + 2, (j + 1) * 4 + i + 1
            lines.append(f"f {a}/{a}/1 {b}/{b}/1 {c}/{c}/1 {d}/{d}/1")
    return _write_if_changed(os.path.join(GENERATED, "neg_uv.obj"), "\n".join(lines) + "\n")


# --------------------------------------------------------------------------- building blocks
def _std_cameras(api, **over):
    kw = dict(fovy=60, near=0.1, far=20, backface_culling=True)
    kw.update(over)
    return (api.Camera((0.5, 1, 2), (0, 0, 0), **kw), api.Camera((0.5, 1, 2), (0, 0, 0), **kw))


def _std_light(api, **over):
    kw = dict(ambient_strength=0.1, specular_strength=0.1)
    kw.update(over)
    return api.Light((2, 3, 4), **kw)


def _diablo(api):
    d = os.path.join(ASSETS, "diablo3_pose")
    m = api.Model.load_model(os.path.join(d, "diablo3_pose.obj"))
    m.textures.register("normals", os.path.join(d, "diablo3_pose_nm_tangent.tga"), tangent=True)
    m.textures.register("diffuse", os.path.join(d, "diablo3_pose_diffuse.tga"), normalize=False)
    return m


def _floor(api, textured=True):
    m = api.Model.load_model(floor_obj())
    if textured:
        m.textures.register("diffuse", os.path.join(ASSETS, "floor_diffuse.tga"), normalize=False)
    return m


def _torus(api, nu, nv):
    m = api.Model.load_model(torus_obj(nu, nv))
    m.textures.register("diffuse", os.path.join(ASSETS, "grid.tga"), normalize=False)
    m.textures.register("normals", os.path.join(ASSETS, "floor_nm_tangent.tga"), tangent=True)
    return m


def _scene(api, cam, dbg, light, resolution, models, **kw):
    sc = api.Scene(cam, light, debug_camera=dbg, resolution=resolution, **kw)
    for m in models:
        sc.add_model(m)
    # The captures under tests/golden/ were taken with the reference's debug-frustum overlay patched out
    # (make_golden.py), the *_overlay ones with it left on; the product's switch for it defaults to ON like
    # upstream, so the recipes turn it off and the overlay tests turn it back on.  (On the reference's
    # Scene this is just an unused attribute.)
    sc.draw_debug_frustum = False
    return sc


# --------------------------------------------------------------------------- recipes
def cube_small(api, resolution=(120, 160)):
    """G1: cube.obj with .mtl (map_Kd, map_Ks, Ns 32), point light."""
    cam, dbg = _std_cameras(api)
    cube = api.Model.load_model(os.path.join(ASSETS, "cube", "cube.obj"))
    return _scene(api, cam, dbg, _std_light(api), resolution, [cube])


def cube_outward(api, resolution=(120, 160)):
    """Cube with normals flipped outward (lit faces show the specular map) and f64 vertices
    (``Model @ scale`` promotes float32 vertices to float64)."""
    cam, dbg = _std_cameras(api)
    cube = api.Model.load_model(os.path.join(ASSETS, "cube", "cube.obj"))
    cube.normals = -cube.normals
    cube = cube @ api.scale(0.45)
    floor = _floor(api, textured=False)
    return _scene(api, cam, dbg, _std_light(api), resolution, [cube, floor])


def diablo_small(api, resolution=(240, 320)):
    """G2: diablo, tangent-space normal map, shadows onto itself."""
    cam, dbg = _std_cameras(api)
    return _scene(api, cam, dbg, _std_light(api), resolution, [_diablo(api)])


def diablo_floor_lh_gl(api, resolution=(270, 480)):
    """G3: upstream main.py's parameters (obj/main.py:63-92,117-129): LH/OpenGL, directional light
    (w=2 extrusion), culling off, and a debug camera that really clips."""
    light = api.Light((5, 5, 0), light_type=api.Lightning.DIRECTIONAL_LIGHTNING, center=(0, 0.5, 0.5),
                      fovy=90, linear=0.000000001, quadratic=0.0000000001,
                      ambient_strength=0.1, specular_strength=0.1)
    cam = api.Camera((0.5, 3, 5), up=np.array((0, 1, 0)), fovy=90, near=0.0001, far=400,
                     backface_culling=False, center=(0, 0, 0))
    dbg = api.Camera((0, 3, 0.01), up=np.array((0, 1, 0)), fovy=80, near=1, far=3,
                     backface_culling=True, center=(0, 0, 0))
    return _scene(api, cam, dbg, light, resolution, [_diablo(api), _floor(api)],
                  system=api.SYSTEM.LH, subsystem=api.SUBSYSTEM.OPENGL)


def diablo_floor(api, resolution=(270, 480)):
    """c3 at reduced size: diablo + floor, RH/DirectX, point light, shadows."""
    cam, dbg = _std_cameras(api)
    return _scene(api, cam, dbg, _std_light(api), resolution, [_diablo(api), _floor(api)])


def torus_spot(api, resolution=(180, 320), nu=40, nv=25):
    """G4: 2 000-triangle torus + floor under a spot light."""
    cam, dbg = _std_cameras(api)
    light = api.Light((2, 3, 4), light_type=api.Lightning.SPOT_LIGHTNING,
                      ambient_strength=0.1, specular_strength=0.1)
    return _scene(api, cam, dbg, light, resolution, [_torus(api, nu, nv), _floor(api)])


def torus_floor(api, resolution=(1080, 1920), nu=500, nv=200):
    """c4 (nu=500, nv=200 -> 200 000 triangles) and its reduced variants."""
    cam, dbg = _std_cameras(api)
    return _scene(api, cam, dbg, _std_light(api), resolution, [_torus(api, nu, nv), _floor(api)])


def _cubemap(api):
    d = os.path.join(ASSETS, "cubemap")            # face assignment of obj/main.py:101-106
    return api.CubeMap(back=os.path.join(d, "neg-z.jpg"), front=os.path.join(d, "pos-z.jpg"),
                       top=os.path.join(d, "pos-y.jpg"), bottom=os.path.join(d, "neg-y.jpg"),
                       left=os.path.join(d, "neg-x.jpg"), right=os.path.join(d, "pos-x.jpg"))


def cube_skybox(api, resolution=(135, 240)):
    """G5: the outward cube + floor in front of the 512^2 cubemap skybox."""
    sc = cube_outward(api, resolution=resolution)
    sc.skybox = _cubemap(api)
    return sc


def torus_skybox(api, resolution=(2160, 3840), nu=1000, nv=500):
    """c5: 1M-triangle torus + floor + cubemap skybox at 3840x2160 (and reduced variants)."""
    sc = torus_floor(api, resolution=resolution, nu=nu, nv=nv)
    sc.skybox = _cubemap(api)
    return sc


def tetra_bare(api, resolution=(120, 160)):
    """Model without vertex normals and without textures on a textured floor: Kd colour,
    face-normal shading, GL/RH projection."""
    cam, dbg = _std_cameras(api)
    tet = api.Model.load_model(bare_tetra_obj())
    return _scene(api, cam, dbg, _std_light(api), resolution, [tet, _floor(api)],
                  system=api.SYSTEM.RH, subsystem=api.SUBSYSTEM.OPENGL)


def diablo_closeup(api, resolution=(200, 200)):
    """Camera close enough that triangles cross the frustum sides: exercises the per-fragment
    clip (obj/triangular.py:80-87) and screen-edge bounding boxes."""
    kw = dict(fovy=40, near=0.3, far=5, backface_culling=True)
    cam = api.Camera((0.2, 0.5, 0.9), (0, 0.3, 0), **kw)
    dbg = api.Camera((0.2, 0.5, 0.9), (0, 0.3, 0), **kw)
    return _scene(api, cam, dbg, _std_light(api), resolution, [_diablo(api)])


def diablo_nm_object(api, resolution=(240, 320)):
    """Object-space normal map (``register('normals', ..., tangent=False)``, obj/core.py:175-181):
    the texel is the normal, no TBN."""
    cam, dbg = _std_cameras(api)
    d = os.path.join(ASSETS, "diablo3_pose")
    m = api.Model.load_model(os.path.join(d, "diablo3_pose.obj"))
    m.textures.register("normals", os.path.join(d, "diablo3_pose_nm.tga"), tangent=False)
    m.textures.register("diffuse", os.path.join(d, "diablo3_pose_diffuse.tga"), normalize=False)
    return _scene(api, cam, dbg, _std_light(api), resolution, [m])


def diablo_closeup_noclip(api, resolution=(200, 200)):
    """``Model.clip = False`` (obj/triangular.py:80): the per-fragment frustum test is skipped, so
    the fragments the close-up camera's planes would cut are kept."""
    sc = diablo_closeup(api, resolution=resolution)
    sc.models[0].clip = False
    return sc


def kat_house(api, resolution=(150, 200)):
    """The loader known-answer meshes rendered: several material groups, fractional Ns
    (``**`` takes the pow() path), map_bump from the library, a model without uv."""
    files = kat_files()
    cam, dbg = _std_cameras(api)
    house = api.Model.load_model(files["kat"])
    porch = api.Model.load_model(files["kat_nouv"])
    return _scene(api, cam, dbg, _std_light(api, specular_strength=0.4), resolution, [house, porch, _floor(api)])


def cube_tetra_nodepth(api, resolution=(120, 160)):
    """``Model.depth_test = False`` (obj/core.py:235-241, obj/triangular.py:117): a tetrahedron that is tested
    against the z-buffer but never writes to it, between two z-writing models in model order, poking
    through the cube (so it wins some pixels, loses others, and later faces draw over it)."""
    cam, dbg = _std_cameras(api)
    cube = api.Model.load_model(os.path.join(ASSETS, "cube", "cube.obj"))
    cube.normals = -cube.normals
    cube = cube @ api.scale(0.45)
    tet = api.Model.load_model(bare_tetra_obj())
    tet.depth_test = False
    tet = tet @ api.scale(1.3)
    return _scene(api, cam, dbg, _std_light(api), resolution, [cube, tet, _floor(api)])


def gizmos_small(api, resolution=(150, 200)):
    """``show=True`` on the light and on the debug camera (obj/core.py:532-552): a sphere at the light's
    place and a camera body at the debug camera's, both ``clip = False``, in front of the cube and the floor.
    The scene is constructed from the directory that holds ``obj_loader_test/`` (upstream's relative paths)."""
    here = os.getcwd()
    os.chdir(gizmo_files())
    try:
        kw = dict(fovy=60, near=0.1, far=20, backface_culling=True)
        cam = api.Camera((0.5, 1, 2), (0, 0, 0), **kw)
        dbg = api.Camera((1.1, 0.5, 0.2), (0, 0, 0), show=True, **kw)
        light = api.Light((-0.9, 1.0, 0.6), ambient_strength=0.1, specular_strength=0.1, show=True)
        sc = api.Scene(cam, light, debug_camera=dbg, resolution=resolution)
    finally:
        os.chdir(here)
    cube = api.Model.load_model(os.path.join(ASSETS, "cube", "cube.obj")) @ api.scale(0.5)
    for m in (cube, _floor(api)):
        sc.add_model(m)
    sc.draw_debug_frustum = False
    return sc


def tetra_ortho(api, resolution=(120, 160)):
    """Orthographic camera (obj/transformation.py:139-154; only OpenGL + LH exists upstream):
    near = |position| (obj/core.py:389), float32 projection matrix."""
    kw = dict(projection_type=api.PROJECTION_TYPE.ORTHOGRAPHIC, fovy=35, far=20, backface_culling=True)
    cam = api.Camera((0.5, 1, 2), (0, 0, 0), **kw)
    dbg = api.Camera((0.5, 1, 2), (0, 0, 0), **kw)
    tet = api.Model.load_model(bare_tetra_obj())
    return _scene(api, cam, dbg, _std_light(api), resolution, [tet, _floor(api)],
                  system=api.SYSTEM.LH, subsystem=api.SUBSYSTEM.OPENGL)


def fins_nonmanifold(api, resolution=(150, 200)):
    """Silhouette edges with three and four incident faces (obj/triangular.py:286-302), culling off so that
    both sides of the fins are drawn, over a floor that catches their shadow volumes."""
    cam, dbg = _std_cameras(api, backface_culling=False)
    fins = api.Model.load_model(fins_obj())
    return _scene(api, cam, dbg, _std_light(api), resolution, [fins, _floor(api)])


def wall_nine_materials(api, resolution=(150, 200)):
    """More materials in one scene than the tile kernel stages in LDS: nine in the wall's library, the cube's
    (map_Kd + map_Ks) and the floor's."""
    cam, dbg = _std_cameras(api)
    wall = api.Model.load_model(wall_files())
    cube = api.Model.load_model(os.path.join(ASSETS, "cube", "cube.obj"))
    cube.normals = -cube.normals
    cube = cube @ api.scale(0.3) @ api.translation((0.2, -0.3, 0.6))
    return _scene(api, cam, dbg, _std_light(api, specular_strength=0.3), resolution, [wall, cube, _floor(api)])


def quad_negative_uv(api, resolution=(150, 200)):
    """Texture coordinates below 0 and above 1 on a textured, normal-mapped quad (obj/core.py:138-143)."""
    cam, dbg = _std_cameras(api)
    quad = api.Model.load_model(neg_uv_obj())
    quad.textures.register("diffuse", os.path.join(ASSETS, "grid.tga"), normalize=False)
    quad.textures.register("normals", os.path.join(ASSETS, "floor_nm_tangent.tga"), tangent=True)
    return _scene(api, cam, dbg, _std_light(api), resolution, [quad, _floor(api)])


# name -> (builder, kwargs); the small ones have full golden buffers committed
SMALL = {
    "cube_small": (cube_small, {}),
    "cube_outward": (cube_outward, {}),
    "diablo_small": (diablo_small, {}),
    "diablo_floor_lh_gl": (diablo_floor_lh_gl, {}),
    "diablo_floor_small": (diablo_floor, {}),
    "torus_spot": (torus_spot, {}),
    "tetra_bare": (tetra_bare, {}),
    "diablo_closeup": (diablo_closeup, {}),
    "cube_skybox": (cube_skybox, {}),
    "torus_skybox_small": (torus_skybox, {"resolution": (216, 384), "nu": 60, "nv": 30}),
    "diablo_nm_object": (diablo_nm_object, {}),
    "diablo_closeup_noclip": (diablo_closeup_noclip, {}),
    "kat_house": (kat_house, {}),
    "tetra_ortho": (tetra_ortho, {}),
    "cube_tetra_nodepth": (cube_tetra_nodepth, {}),
    "gizmos_small": (gizmos_small, {}),
    "fins_nonmanifold": (fins_nonmanifold, {}),
    "wall_nine_materials": (wall_nine_materials, {}),
    "quad_negative_uv": (quad_negative_uv, {}),
}

# BASELINE.json configs at full size: only the uint8 frame, winner map, stencil and z row sums are kept
FULL = {
    "c1_diablo_800x600": (diablo_small, {"resolution": (600, 800)}),       # BASELINE configs[0]; shadows off
    "c2_diablo_1080p": (diablo_small, {"resolution": (1080, 1920)}),       # rendered with shadows off
    "c3_diablo_floor_1080p": (diablo_floor, {"resolution": (1080, 1920)}),
    "c4_torus200k_1080p": (torus_floor, {"resolution": (1080, 1920), "nu": 500, "nv": 200}),
}
# BASELINE.json configs[4]; its capture takes the reference ~8 minutes and is generated separately
HUGE = {
    "c5_torus1m_4k_skybox": (torus_skybox, {"resolution": (2160, 3840), "nu": 1000, "nv": 500}),
}
NO_SHADOW = {"c1_diablo_800x600", "c2_diablo_1080p", "diablo_small_noshadow"}
# the same scenes with upstream's debug-frustum overlay left on (obj/core.py:638)
OVERLAY = ["diablo_small_overlay", "diablo_floor_lh_gl_overlay", "cube_outward_overlay"]


def build(api, name):
    if name == "diablo_small_noshadow":
        return diablo_small(api)
    fn, kw = {**SMALL, **FULL, **HUGE}[name]
    return fn(api, **kw)
