"""Scene API: ``Model / Camera / Light / Scene`` with ``Scene.render() -> uint8 (H, W, 3)``.

Drop-in for the classes of the reference's ``obj/core.py`` (Model :231, Camera :432,
Light :444, Scene :558): same constructor arguments, same attribute names, same OBJ/MTL
ingest rules.  What differs is where the work happens: ``Scene.render`` packs the scene into
flat arrays (``_pack.py``) and hands it to the HIP library through the C ABI declared in
``include/mi355rast.h``; nothing is rasterised or shaded on the host.
"""
import os
from functools import cached_property
from typing import Iterable, List

import numpy as np
from PIL import Image

from . import _fp
from .constants import *  # noqa: F401,F403  (re-exported like the reference's core module)
from .constants import PROJECTION_TYPE, SUBSYSTEM, SYSTEM, mat3x3
from .lightning import Lightning
from .materials import Material
from .transformation import (ViewPort, look_at_rotate_lh, look_at_rotate_rh, looka_at_translate,
                             normalize, perspectives, scale)


def triangulate_int(polygon):
    """Fan triangulation of one parsed ``f`` record (reference: ``obj/core.py:72-74``)."""
    for i in range(1, len(polygon) - 1):
        yield np.array([polygon[0], polygon[i], polygon[i + 1]], dtype=np.int32)


class TextureMaps:
    """``model.textures.register(kind, path, normalize=True, tangent=False)``
    (reference: ``obj/core.py:77-105``)."""

    texture_map = {
        "diffuse": "map_Kd",
        "ambient": "map_Ka",
        "specular": "map_Ks",
        "shininess": "map_Ns",
        "transparency": "map_d",
        "normals": "norm",
    }

    def __init__(self, model):
        self.model = model

    def register(self, attr_name, path, normalize=True, tangent=False):
        key = self.texture_map.get(attr_name)
        if key is None:
            raise ValueError(f"{attr_name} not recognized.\nSupported: {self.texture_map.keys()}")
        texels = self.load_texture(path)
        if normalize:                       # [0,1] -> [-1,1]: meant for normal maps, and the default
            texels = texels * 2 - 1
        tagged = np.dtype(np.float32, metadata={"tangent": tangent})
        setattr(self.model.materials["default"], key, np.array(texels, dtype=tagged))
        self.model._revision += 1

    @staticmethod
    def load_texture(name):
        with Image.open(name) as img:
            return np.asarray(img.convert("RGB")) / 255


class Model:
    """Triangle mesh: ``vertices`` (V,4), ``uv`` (T,3), ``normals`` (N,3) and ``_faces``
    (F,3,4) holding per corner ``[vertex, uv, normal, material-group]`` indices."""

    def __init__(self, vertices, uv, normals, faces, shadowing=False, materials=None,
                 material_group=None, clip=True, depth_test=True):
        self.vertices = vertices
        self.uv = uv
        self.normals = normals
        self._faces = faces
        self.shadowing = shadowing          # kept for signature parity; the reference never reads it
        self.clip = clip
        self.depth_test = depth_test
        self.materials = materials or {"default": Material()}
        self.material_group = material_group or ["default"]
        self.textures = TextureMaps(self)
        self.shape = None
        self.silhouette = set()             # filled by Scene.render with the last frame's edges
        self._revision = 0                  # bumped whenever device copies go stale

    # -- ingest ---------------------------------------------------------------------------
    @classmethod
    def load_model(cls, name, shadowing=True):
        """Parse a Wavefront OBJ (reference: ``obj/core.py:257-318``).

        ``v`` gets ``w = 1`` appended, 2-component ``vt`` is padded with 0, polygons are
        fan-triangulated, a missing index (``1//3``) becomes -1, positive indices become
        0-based and negative ones are left as they are (NumPy-relative)."""
        verts, uvs, norms, tris = [], [], [], []
        groups = ["default"]
        current = "default"
        materials = {"default": Material()}
        folder = os.path.dirname(name)
        with open(name) as fh:
            for line in fh:
                tag, _, rest = line.partition(" ")
                if tag == "v":
                    xyz = rest.split()
                    verts.append(xyz + [1] if len(xyz) == 3 else xyz)
                elif tag == "vt":
                    st = rest.split()
                    uvs.append(st + [0] if len(st) == 2 else st)
                elif tag == "vn":
                    norms.append(rest.split())
                elif tag == "f":
                    group_id = groups.index(current) + 1
                    corners = [[(ref if ref != "" else -1) for ref in corner.split("/")] + [group_id]
                               for corner in rest.split()]
                    tris.extend(triangulate_int(corners))
                elif tag == "usemtl":
                    current = rest.split()[0]
                    if current not in groups:
                        groups.append(current)
                elif tag == "mtllib":
                    lib = os.path.join(folder, rest.split()[0])
                    if os.path.exists(lib):
                        materials.update(cls.parse_mtl(lib))
        faces = np.array(tris)
        faces = np.where(faces > 0, faces - 1, faces)
        return cls(np.array(verts, dtype=np.float32),
                   np.array(uvs, dtype=np.float32) if uvs else None,
                   np.array(norms, dtype=np.float32) if norms else None,
                   faces, shadowing, materials=materials, material_group=groups)

    @staticmethod
    def parse_mtl(mtllib):
        """``.mtl`` -> {name: Material}; ``map*``/``disp`` keys load the image next to the
        library, ``map_bump`` is stored as a tangent-space ``norm`` (``obj/core.py:321-348``)."""
        library = {}
        folder = os.path.dirname(mtllib)
        material = None
        with open(mtllib) as fh:
            for line in fh:
                if line.startswith("#") or line == "\n":
                    continue
                key, *val = line.split()
                if key == "newmtl":
                    material = library[val[0]] = Material()
                elif key.startswith("map") or key == "disp":
                    path = os.path.join(folder, val[0])
                    if not os.path.exists(path):
                        print(f"{key} {path} is not found. Recommend manually assign texture by descriptor "
                              f"Model.texture.register")
                        continue
                    dtype = np.float32
                    if key == "map_bump":
                        key, dtype = "norm", np.dtype(np.float32, metadata={"tangent": True})
                    setattr(material, key, np.array(TextureMaps.load_texture(path), dtype=dtype))
                else:
                    setattr(material, key, val)
        return library

    def __matmul__(self, other):
        self.vertices = self.vertices @ other
        self._revision += 1
        return self

    def invalidate(self):
        """Tell the renderer that ``vertices`` / ``uv`` / ``normals`` / ``_faces`` were edited IN PLACE.
        Replacing an array, ``Model @ M``, ``textures.register`` and material assignments are noticed
        by themselves, and so are in-place edits that change a sampled checksum of the arrays (scaling,
        ``[:] =``); a change to a few single elements needs this call."""
        self._revision += 1

    def face_material(self, group_index):
        """Material of a face whose first corner carries *group_index* (``obj/core.py:125``)."""
        return self.materials.get(self.material_group[group_index], self.materials["default"])


_PROJECTIONS = {}


class PositionedObject:
    def __init__(self, position, center=np.array([0, 0, 0])):
        self.scene = None
        self.position = position
        self.center = center

    @property
    def direction(self):
        # (asked for once per frame for the light, which rarely moves: the last answer is kept with the bytes it was
        # computed from)
        p, c = np.asarray(self.position), np.asarray(self.center)
        key = (p.tobytes(), c.tobytes(), p.dtype.str, c.dtype.str)
        hit = self.__dict__.get("_direction_memo")
        if hit is None or hit[0] != key:
            hit = self.__dict__["_direction_memo"] = (key, normalize(p - c).ravel())
        return hit[1].copy()

    def direction_to(self, other):
        return normalize(self.direction - other)

    def set_position(self, new_position):
        self.position = new_position
        return self


class TransformationMatrixMixin:
    """View / projection matrices of a positioned object (``obj/core.py:373-429``).

    ``lookat`` and ``MVP`` are cached on first use, as in the reference.  The 4x4 products
    use explicit fma chains (``_fp.matmul_chain``) so the constants handed to the device are
    the same on every host."""

    def __init__(self, x_offset=0, y_offset=0, projection_type=PROJECTION_TYPE.PERSPECTIVE,
                 up=np.array([0, 1, 0]), near=0.001, far=6, fovy=90):
        self.up = up
        self.projection_type = projection_type
        self.near = (np.linalg.norm(self.position)
                     if projection_type == PROJECTION_TYPE.ORTHOGRAPHIC else near)
        self.far = far
        self.fovy = fovy
        self.x_offset = x_offset
        self.y_offset = y_offset
        self.scene = None

    @property
    def projection(self):
        height, width = self.scene.resolution
        build = perspectives[self.scene.subsystem][self.projection_type][self.scene.system]
        # (a camera that moves keeps its lens: the matrix of the last few parameter sets is kept)
        try:
            key = (build, float(self.fovy), width / height, float(self.near), float(self.far))
            hit = _PROJECTIONS.get(key)
        except (TypeError, ValueError):
            return build(self.fovy, width / height, self.near, self.far)
        if hit is None:
            if len(_PROJECTIONS) > 64:
                _PROJECTIONS.clear()
            hit = _PROJECTIONS[key] = build(self.fovy, width / height, self.near, self.far)
        return hit.copy()

    @property
    def rotate(self):
        # argument order (center, position) is the reference's: the camera looks along -forward
        if self.scene.system == SYSTEM.LH:
            return look_at_rotate_lh(self.center, self.position, self.up)
        return look_at_rotate_rh(self.center, self.position, self.up)

    @property
    def translate(self):
        return looka_at_translate(self.position)

    def _constants_native(self):
        """look-at, MVP and the frustum planes in ONE call into the library's host code
        (``mr_host_camera_constants``: the same operations in the same order as the properties below, bit for bit --
        ``tests/test_host_api.py``); False where the library is not built.  A camera that moves is a new object every
        frame, and the NumPy route costs a tenth of a millisecond per camera."""
        fast = _fp.camera_constants()
        if not fast or "lookat" in self.__dict__ or "MVP" in self.__dict__:
            return False
        try:
            vec = lambda v: np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(3))
            eye, center, up, pos = vec(self.center), vec(self.position), vec(self.up), vec(self.position)
            proj = np.ascontiguousarray(self.projection, dtype=np.float64)
        except (TypeError, ValueError, AttributeError, KeyError):      # (e.g. the look-at of an object whose scene is still being built)
            return False
        if proj.shape != (4, 4):
            return False
        lookat, mvp, planes = np.empty((4, 4)), np.empty((4, 4)), np.empty((6, 4))
        fast(eye.ctypes.data, center.ctypes.data, up.ctypes.data, pos.ctypes.data, proj.ctypes.data,
             1 if self.scene.system == SYSTEM.LH else 0, lookat.ctypes.data, mvp.ctypes.data, planes.ctypes.data)
        self.__dict__["lookat"], self.__dict__["MVP"] = lookat, mvp
        self.__dict__["_planes_memo"] = (mvp, planes)
        return True

    @cached_property
    def lookat(self):
        if self._constants_native():
            return self.__dict__["lookat"]
        return _fp.matmul_chain(self.translate, self.rotate)

    @cached_property
    def MVP(self):
        if self._constants_native():
            return self.__dict__["MVP"]
        return _fp.matmul_chain(self.lookat, self.projection)

    @property
    def frustum_planes(self):
        from .plane_intersection import extract_frustum_planes
        mvp = self.MVP
        memo = self.__dict__.get("_planes_memo")
        if memo is not None and memo[0] is mvp:
            return memo[1].copy()
        return extract_frustum_planes(mvp)

    @property
    def viewport(self):
        return ViewPort(self.scene.resolution, self.far, self.near,
                        x_offset=self.x_offset, y_offset=self.y_offset)


class Camera(PositionedObject, TransformationMatrixMixin):
    def __init__(self, position, center, show=False, backface_culling=True, **kwargs):
        PositionedObject.__init__(self, np.array(position), center)
        TransformationMatrixMixin.__init__(self, **kwargs)
        self.show = show
        self.backface_culling = backface_culling


class Light(PositionedObject, TransformationMatrixMixin):
    """Point / directional / spot light with the reference's attenuation model
    ``1 / (constant + d (linear + quadratic d))`` (``obj/core.py:444-524``)."""

    def __init__(self, position, light_type=Lightning.POINT_LIGHTNING, center=(0, 0, 0),
                 color=(1., 1., 1.), ambient_strength=0, diffuse=1, specular_strength=0.5,
                 show=False, constant=1, linear=0.14, quadratic=0.07, **kwargs):
        self.color = np.array(color)
        self.light_type = light_type
        PositionedObject.__init__(self, np.array(position), np.array(center))
        self.ambient = ambient_strength * self.color
        self.show = show
        self.diffuse = diffuse
        self.specular_strength = specular_strength
        self.constant = constant
        self.linear = linear
        self.quadratic = quadratic
        TransformationMatrixMixin.__init__(self, **kwargs)

    @staticmethod
    def reflect(I, N):
        return normalize(I - 2. * (N * I).sum(axis=1)[..., np.newaxis] * N)

    @staticmethod
    def smoothstep(edge0, edge1, x_array):
        t = np.clip((x_array - edge0) / (edge1 - edge0), 0.0, 1.0)
        return t * t * (3 - 2 * t)

    def attenuation(self, fragment_position):
        d = np.linalg.norm(self.position - fragment_position, axis=1)
        return 1.0 / (self.constant + d * (self.linear + self.quadratic * d))[..., np.newaxis]


class Bound:
    """Descriptor tying a camera / light to the scene it is assigned to
    (``obj/core.py:527-555``).  State is kept per scene instance (the reference keeps it on
    the descriptor, so all its scenes share the last assignment)."""

    def __set_name__(self, owner, name):
        self._slot = "_bound_" + name

    def __set__(self, instance, value):
        instance.__dict__[self._slot] = value
        if value is None:
            return
        value.scene = instance
        if getattr(value, "show", False):
            instance.add_model(_gizmo(value))

    def __get__(self, instance, owner):
        if instance is None:
            return self
        return instance.__dict__.get(self._slot)


def _gizmo(value):
    """The little mesh that marks a ``show=True`` camera or light (``obj/core.py:532-552``): a sphere
    for a light, a camera body for a camera, scaled to a tenth and carried to the object's place by the
    inverse of its look-at matrix.  The meshes are loaded from ``obj_loader_test/sphere.obj`` /
    ``obj_loader_test/camera.obj`` relative to the working directory, exactly as upstream does -- which
    does not ship them, so without files of your own there this raises ``FileNotFoundError`` there and here.
    Vertices and normals leave as float64 (float32 @ float64), as upstream's do; the device keeps normals
    in float32 (the rounding is 6e-8 of a unit vector, far inside the float frame's 2e-6)."""
    from .transformation import scale
    is_light = isinstance(value, Light)
    sub = Model.load_model("obj_loader_test/sphere.obj" if is_light else "obj_loader_test/camera.obj",
                           shadowing=False)
    sub.clip = False
    sub = sub @ scale(0.1)
    lookat = value.lookat
    if is_light:            # a light straight above its centre has a singular look-at: upstream falls back to pinv
        try:
            sub = sub @ np.linalg.inv(lookat)
        except np.linalg.LinAlgError:
            sub = sub @ np.linalg.pinv(lookat)
        try:
            sub.normals = -sub.normals @ np.linalg.inv(lookat[mat3x3])
        except np.linalg.LinAlgError:
            sub.normals = -sub.normals @ np.linalg.pinv(lookat[mat3x3])
    else:
        sub = sub @ np.linalg.inv(lookat)
        sub.normals = -sub.normals @ np.linalg.inv(lookat[mat3x3])
    return sub


class PendingFrame:
    """A frame enqueued by ``Scene.render_async``: ``result()`` waits for it and returns the ``uint8 (H, W, 3)`` array."""

    def __init__(self, scene, lane, out, shadows, cameras):
        self._scene, self._lane, self._out, self._shadows, self._cameras = scene, lane, out, shadows, cameras
        self._done = False

    def result(self):
        if not self._done:
            scene = self._scene
            ok = scene._backend().render_wait(self._lane)
            scene.__dict__.get("_pending", {}).pop(self._lane, None)
            if not ok:          # a work list overflowed (now grown): render this frame's view again, synchronously
                now = (scene.camera, scene.debug_camera)
                scene.camera, scene.debug_camera = self._cameras
                try:
                    self._out = scene.render(shadows=self._shadows)
                finally:
                    scene.camera, scene.debug_camera = now
            self._done = True
        return self._out


class Scene:
    """``Scene(camera, light, shadows, debug_camera, resolution=(H, W), system, subsystem,
    skymap)`` -- reference ``obj/core.py:558-640``.

    * ``debug_camera`` also clips fragments, exactly as in the reference
      (``obj/triangular.py:39,83``); ``None`` means "same as ``camera``" (the reference raises).
    * ``shadows`` is accepted and ignored like upstream (``obj/core.py:568``): the stencil
      pass always runs.  Use ``render(shadows=False)`` to skip it explicitly.
    """

    camera = Bound()
    light = Bound()
    debug_camera = Bound()

    def __init__(self, camera=None, light=None, shadows=False, debug_camera=None,
                 resolution=(1500, 1500), system=SYSTEM.RH, subsystem=SUBSYSTEM.DIRECTX,
                 skymap=None, device=None):
        self.system = system
        self.subsystem = subsystem
        self.models: List[Model] = []
        self.camera = camera if camera is not None else Camera(position=(0, 0, 1), center=(0, 0, 0))
        self.light = light if light is not None else Light(position=(1, 1, 1))
        self.debug_camera = debug_camera
        self.resolution = resolution
        self.skybox = skymap
        self.shadows = shadows
        self.device = device
        self.draw_debug_frustum = True      # like the reference, which always overlays it (core.py:638); switchable here
        self.verbose = False                # the reference always prints its face histogram (core.py:634-636)
        self._renderer = None

    @property
    def last_stats(self):
        """Statistics of the last frame (``mr_stats``), fetched from the device when asked for."""
        return None if self._renderer is None else self._renderer.last_stats

    def add_model(self, model):
        self.models.append(model)

    # -- rendering ------------------------------------------------------------------------
    def _backend(self):
        if self._renderer is None:
            from ._native import DeviceRenderer
            self._renderer = DeviceRenderer(self.device)
        return self._renderer

    def render(self, shadows=True, row_band=None):
        """Render one frame on the GPU and return ``uint8 (H, W, 3)`` (row 0 = top).

        Unlike the reference, repeated calls give the same frame: the silhouette is rebuilt
        from scratch every frame instead of being toggled in ``model.silhouette``.  Like the
        reference, the frame carries the debug camera's frustum as red lines (``obj/core.py:638``);
        ``scene.draw_debug_frustum = False`` leaves it out.  With ``scene.verbose`` (off by default)
        the three lines the reference prints per model after its lit pass (``obj/core.py:634-636``)
        are reproduced from the device's per-face codes."""
        backend = self._backend()
        report = self.verbose and row_band is None
        # the debug camera's frustum, drawn by the device into its own frame and z-buffer right after the
        # tile kernel (obj/core.py:638).  A row band of a multi-GPU split cannot draw it alone (the lines test z at
        # pixels other devices own): multigpu.BandRenderer(overlay=True) gathers the touched pixels' state with
        # the rows and replays the overlay on the assembled frame
        overlay = self.draw_debug_frustum and row_band is None
        out = backend.render(self, shadows=shadows, row_band=row_band, face_status=report, counters=False,
                             keep_buffers=False, timing=False, overlay=overlay)
        if report:
            self._print_face_report(backend.read_face_status())
        return out

    def render_async(self, shadows=True):
        """``render()`` without the wait: the frame's kernels and the copy of its uint8 rows to the host are
        enqueued and the call returns a ``PendingFrame``; ``.result()`` hands out the array (the same bytes
        ``render()`` would have returned).  Up to four frames may be pending on one scene; the copy of frame i
        (longer than the frame's kernels at 1080p) then runs beside the kernels of frame i + 1 and beside the
        host's preparation of frame i + 2.  The reference has no such call: it is an addition for sequences."""
        backend = self._backend()
        pending = self.__dict__.setdefault("_pending", {})
        lane = next((k for k in range(backend.ASYNC_LANES) if k not in pending), None)
        if lane is None:
            raise RuntimeError("four frames of this scene are already pending: take one's result() first")
        out = backend.render_async(self, lane, shadows=shadows, overlay=self.draw_debug_frustum)
        frame = PendingFrame(self, lane, out, shadows, (self.camera, self.debug_camera))
        pending[lane] = frame
        return frame

    def render_frames(self, views, shadows=True, depth=2):
        """Frames of a sequence: for every ``(camera, debug_camera)`` pair of *views* the scene's cameras are set
        and a frame is rendered; yields the uint8 arrays in order, *depth* frames in flight (``render_async``)."""
        queue = []
        try:
            for camera, debug_camera in views:
                self.camera, self.debug_camera = camera, debug_camera
                queue.append(self.render_async(shadows=shadows))
                if len(queue) >= max(1, int(depth)):
                    yield queue.pop(0).result()
            while queue:
                yield queue.pop(0).result()
        finally:
            for frame in queue:
                frame.result()

    def _print_face_report(self, status):
        from .triangular import Errors
        first = 0
        for model in self.models:
            codes = status[first:first + len(model._faces)]
            first += len(model._faces)
            histogram = {err: int((codes == err.value).sum()) for err in Errors}
            print('Total faces', len(model._faces))
            print('Face rendered', int((codes == 0).sum()))
            print('Discarded', histogram)

    def close(self):
        if self._renderer is not None:
            self._renderer.close()
            self._renderer = None
