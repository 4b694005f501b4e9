"""ctypes binding of ``libmi355rast.so`` (C ABI: ``include/mi355rast.h``).

``DeviceRenderer`` is what ``Scene.render`` talks to: it mirrors the scene into the
library (re-uploading only when a model or texture changed), fills the per-frame constant
block and calls ``mr_render``.  There is no CPU fallback: if the library is missing or no
GPU is visible the calls raise ``RuntimeError``.
"""
import ctypes as C
import os
import zlib

import numpy as np

from ._pack import pack_frame, pack_model

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libmi355rast.so"
LIB_PATH = os.path.join(_HERE, LIB_NAME)

MR_OK = 0
MR_E_OVERFLOW = -4
FRAME_SHADOWS, FRAME_KEEP_FLOAT, FRAME_FACE_STATUS, FRAME_LIGHT_TIMING, FRAME_SKYBOX, FRAME_COUNTERS = 1, 2, 4, 8, 16, 32
FRAME_KEEP_BUFFERS = 64
FRAME_NO_TIMING = 128
FRAME_OVERLAY = 256
ABI_VERSION = 3
TILE_RECORD_WORDS = 12


class FrameDesc(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("system", C.c_int32),
                ("backface_culling", C.c_int32), ("light_type", C.c_int32), ("flags", C.c_int32),
                ("row_begin", C.c_int32), ("row_end", C.c_int32),
                ("stripe_count", C.c_int32), ("stripe_index", C.c_int32),
                ("mvp", C.c_double * 16), ("viewport", C.c_double * 16), ("debug_mvp", C.c_double * 16),
                ("frustum_planes", C.c_double * 24), ("z_near", C.c_double), ("z_far", C.c_double),
                ("camera_pos", C.c_double * 3), ("light_pos", C.c_double * 3), ("light_dir", C.c_double * 3),
                ("light_color", C.c_double * 3), ("light_ambient", C.c_double * 3),
                ("specular_strength", C.c_double), ("att_constant", C.c_double), ("att_linear", C.c_double),
                ("att_quadratic", C.c_double), ("spot_edge0", C.c_double), ("spot_edge1", C.c_double),
                ("background", C.c_float * 3), ("background_u8", C.c_uint32),
                ("sky_tri", C.c_int32 * 12), ("sky_rays", C.c_double * 18)]


class MaterialDesc(C.Structure):
    _fields_ = [("kd", C.c_double * 3), ("ks255", C.c_double * 3), ("ns", C.c_double),
                ("tex_kd", C.c_int32), ("tex_norm", C.c_int32), ("tex_ks", C.c_int32),
                ("norm_tangent", C.c_int32)]


class ModelDesc(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("uv", C.c_void_p), ("normals", C.c_void_p), ("faces", C.c_void_p),
                ("materials", C.POINTER(MaterialDesc)), ("edge_ids", C.c_void_p),
                ("n_vertices", C.c_int32), ("n_uv", C.c_int32), ("n_normals", C.c_int32),
                ("n_faces", C.c_int32), ("n_materials", C.c_int32),
                ("vertices_are_f32", C.c_int32), ("clip", C.c_int32), ("depth_test", C.c_int32)]


class OverlayDesc(C.Structure):
    _fields_ = [("n_segments", C.c_int32), ("n_points", C.c_int32), ("height", C.c_int32), ("width", C.c_int32),
                ("seg_first", C.c_void_p), ("seg_count", C.c_void_p), ("target", C.c_void_p), ("z", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = ([(n, C.c_int64) for n in (
        "frag_tri", "frag_quad", "covered_px", "lit_px", "stencil_updates", "n_faces", "n_faces_setup",
        "n_quads", "n_quads_drawn", "tri_bin_entries", "quad_bin_entries")] +
        [(n, C.c_float) for n in ("gpu_ms_total", "gpu_ms_setup", "gpu_ms_binning", "gpu_ms_tile", "gpu_ms_copy")])

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# every symbol include/mi355rast.h declares: (restype, argtypes)
_PROTOTYPES = {
    "mr_init": (C.c_int, [C.c_int]),
    "mr_device_available": (C.c_int, []),
    "mr_abi_version": (C.c_int, []),
    "mr_abi_struct_size": (C.c_int, [C.c_int]),
    "mr_scene_create": (C.c_void_p, []),
    "mr_scene_destroy": (None, [C.c_void_p]),
    "mr_scene_add_texture": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "mr_scene_set_skybox": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "mr_scene_add_model": (C.c_int, [C.c_void_p, C.POINTER(ModelDesc)]),
    "mr_scene_clear": (C.c_int, [C.c_void_p]),
    "mr_scene_set_overlay": (C.c_int, [C.c_void_p, C.POINTER(OverlayDesc)]),
    "mr_scene_set_overlay_cameras": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                                               C.c_int32, C.c_int32, C.c_int32]),
    "mr_scene_set_list_capacities": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "mr_render": (C.c_int, [C.c_void_p, C.POINTER(FrameDesc), C.c_void_p, C.POINTER(Stats)]),
    "mr_overlay_state_bytes": (C.c_int64, [C.c_void_p]),
    "mr_overlay_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "mr_render_async": (C.c_int, [C.c_void_p, C.POINTER(FrameDesc), C.c_void_p, C.c_int32]),
    "mr_render_wait": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(Stats)]),
    "mr_host_alloc": (C.c_void_p, [C.c_uint64]),
    "mr_host_free": (None, [C.c_void_p]),
    "mr_render_device": (C.c_int, [C.c_void_p, C.POINTER(FrameDesc), C.c_void_p, C.c_void_p]),
    "mr_get_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "mr_get_kernel_times": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_int]),
    "mr_get_stream_kernel_times": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_int]),
    "mr_read_z": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mr_read_stencil": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mr_read_winner": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mr_read_frame_f32": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mr_read_face_status": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mr_read_silhouette": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "mr_host_matmul_chain": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "mr_host_overlay_build": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int32,
                                        C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mr_host_overlay_fetch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mr_debug_read_tile_records": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "mr_debug_clusters_culled": (C.c_int, [C.c_void_p]),
    "mr_host_camera_constants": (None, [C.c_void_p] * 5 + [C.c_int32] + [C.c_void_p] * 3),
    "mr_debug_read_tile_order": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "mr_last_error": (C.c_char_p, []),
}
EXPORTED_SYMBOLS = tuple(_PROTOTYPES)

_lib = None


def _prefer_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own ``libamdhip64.so``; ``multigpu.py`` hands
    torch's streams and tensors to this library, so both must run on the same copy.  Whichever copy is
    loaded first serves both (same SONAME) -- but a torch imported AFTER the system copy finds no
    device.  So when torch is installed its copy is loaded first, without importing torch itself."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
    except (OSError, ImportError, ValueError):
        pass                                    # no torch, or not a ROCm build: the system runtime serves


def load_library():
    """Load the HIP library; raises RuntimeError (never falls back) when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                               "g.build()'` (hipcc --offload-arch=gfx950); there is no CPU fallback")
        _prefer_torch_hip_runtime()
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.mr_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_NAME}: ABI version {lib.mr_abi_version()}, this binding speaks {ABI_VERSION}; rebuild")
        for which, struct in enumerate((FrameDesc, MaterialDesc, ModelDesc, Stats, OverlayDesc)):
            if lib.mr_abi_struct_size(which) != C.sizeof(struct):
                raise RuntimeError(f"{LIB_NAME}: layout of {struct.__name__} differs from the binding "
                                   f"({lib.mr_abi_struct_size(which)} vs {C.sizeof(struct)} bytes); rebuild")
        _lib = lib
    return _lib


def _check(rc, what):
    if rc < 0:
        msg = load_library().mr_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed ({rc}): {msg}")
    return rc


def stripe_out_rows(height, stripe_count):
    """Rows of a device's output buffer in the striped layout (``mr_frame_desc.stripe_count``)."""
    tile_rows = -(-int(height) // 16)
    return -(-tile_rows // int(stripe_count)) * 16


def fill_frame_desc(pf, row_band=None, keep_float=False, light_timing=False, face_status=False, counters=False,
                    keep_buffers=False, stripe=None, no_timing=False, overlay=False):
    d = FrameDesc()
    d.width, d.height, d.system = pf.width, pf.height, pf.system
    d.backface_culling, d.light_type = int(pf.backface_culling), pf.light_type
    d.flags = ((FRAME_SHADOWS if pf.shadows else 0) | (FRAME_KEEP_FLOAT if keep_float else 0)
               | (FRAME_LIGHT_TIMING if light_timing else 0) | (FRAME_FACE_STATUS if face_status else 0)
               | (FRAME_COUNTERS if counters else 0) | (FRAME_KEEP_BUFFERS if keep_buffers else 0)
               | (FRAME_NO_TIMING if no_timing else 0) | (FRAME_OVERLAY if overlay else 0))
    d.row_begin, d.row_end = (0, pf.height) if row_band is None else (int(row_band[0]), int(row_band[1]))
    if stripe is not None:                      # (index, count): interleaved tile rows, see mi355rast.h
        d.stripe_index, d.stripe_count = int(stripe[0]), int(stripe[1])
    for name in ("mvp", "viewport", "debug_mvp", "frustum_planes", "camera_pos", "light_pos", "light_dir",
                 "light_color", "light_ambient"):
        src = np.ascontiguousarray(getattr(pf, name), dtype=np.float64)
        C.memmove(getattr(d, name), src.ctypes.data, src.nbytes)
    d.z_near, d.z_far = pf.z_near, pf.z_far
    d.specular_strength = pf.specular_strength
    d.att_constant, d.att_linear, d.att_quadratic = pf.att_constant, pf.att_linear, pf.att_quadratic
    d.spot_edge0, d.spot_edge1 = pf.spot_edge0, pf.spot_edge1
    for i in range(3):
        d.background[i] = float(pf.background[i])
    # the finalised background exactly as the reference computes it (float32 ** 0.8 * 255 -> uint8)
    bg = (np.asarray(pf.background, dtype=np.float32) ** 0.8 * 255).astype(np.uint8)
    d.background_u8 = int(bg[0]) | int(bg[1]) << 8 | int(bg[2]) << 16 | 1 << 24
    if pf.sky_tri is not None:
        d.flags |= FRAME_SKYBOX
        for i, v in enumerate(np.asarray(pf.sky_tri, dtype=np.int32).ravel()):
            d.sky_tri[i] = int(v)
        for i, v in enumerate(np.asarray(pf.sky_rays, dtype=np.float64).ravel()):
            d.sky_rays[i] = v
    return d


class _PinnedPool:
    """Page-locked output buffers for ``mr_render``, recycled when the NumPy array that wraps one is
    garbage collected: a render loop that drops each frame before asking for the next allocates once."""

    def __init__(self, lib, keep=4):
        self.lib, self.keep = lib, keep
        self.free = {}                       # bytes -> [pointers]

    def array(self, shape):
        import weakref
        nbytes = int(np.prod(shape))
        stock = self.free.setdefault(nbytes, [])
        ptr = stock.pop() if stock else self.lib.mr_host_alloc(nbytes)
        if not ptr:
            return np.empty(shape, dtype=np.uint8)           # pageable memory still works, just slower
        raw = (C.c_uint8 * nbytes).from_address(ptr)
        arr = np.frombuffer(raw, dtype=np.uint8).reshape(shape)
        weakref.finalize(raw, self._release, nbytes, ptr)       # raw lives as long as any view of arr does
        return arr

    def _release(self, nbytes, ptr):
        stock = self.free.setdefault(nbytes, [])
        if len(stock) < self.keep:
            stock.append(ptr)
        else:
            self.lib.mr_host_free(ptr)

    def close(self):
        for stock in self.free.values():
            while stock:
                self.lib.mr_host_free(stock.pop())


class DeviceRenderer:
    """Owns one ``mr_scene`` handle and keeps it in step with a Python ``Scene``."""

    def __init__(self, device=None):
        self.lib = load_library()
        _check(self.lib.mr_init(-1 if device is None else int(device)), "mr_init")
        self.handle = self.lib.mr_scene_create()
        if not self.handle:
            raise RuntimeError("mr_scene_create failed: " + self.lib.mr_last_error().decode())
        self._signature = None
        self._sky_key = None
        self._last_stats = {}
        self._frame = None
        self._packed = (None, None)          # (key, PackedFrame) of the last frame's per-frame constants
        self._pinned = _PinnedPool(self.lib)

    # -- scene upload ---------------------------------------------------------------------
    @staticmethod
    def _fingerprint(arr):
        """Cheap identity of an array's CONTENT: where it lives, its layout, and a checksum of ~64 WHOLE rows
        spread evenly over the array (a few microseconds).  Catches replacement and wholesale in-place edits,
        including an edit of a single column (every row changes, so every sampled row does; sampling flat
        elements at a stride that shares a factor with the row width would skip whole columns); a poke at a
        few single elements needs ``Model.invalidate()``."""
        if arr is None:
            return None
        a = np.asarray(arr)
        rows = a.reshape(a.shape[0], -1) if a.ndim > 1 else a.reshape(-1, 1)
        step = max(1, rows.shape[0] // 64)
        return (a.__array_interface__["data"][0], a.shape, a.dtype.str, zlib.crc32(np.ascontiguousarray(rows[::step])))

    @classmethod
    def _scene_signature(cls, scene):
        from .materials import Material
        return (Material.revision,) + tuple(
            (id(m), m._revision, cls._fingerprint(m.vertices), cls._fingerprint(m._faces), cls._fingerprint(m.uv),
             cls._fingerprint(m.normals), bool(m.clip), bool(m.depth_test), tuple(m.material_group))
            for m in scene.models)

    def sync_scene(self, scene):
        sig = self._scene_signature(scene)
        if sig == self._signature:
            return
        _check(self.lib.mr_scene_clear(self.handle), "mr_scene_clear")
        textures, seen, uploaded = [], {}, 0
        for model in scene.models:
            pm = pack_model(model, textures, seen)
            for tex in textures[uploaded:]:
                _check(self.lib.mr_scene_add_texture(self.handle, tex.ctypes.data, tex.shape[0], tex.shape[1]),
                       "mr_scene_add_texture")
            uploaded = len(textures)
            mats = (MaterialDesc * len(pm.materials))()
            for j, m in enumerate(pm.materials):
                for i in range(3):
                    mats[j].kd[i], mats[j].ks255[i] = m.kd[i], m.ks255[i]
                mats[j].ns = m.ns
                mats[j].tex_kd, mats[j].tex_norm, mats[j].tex_ks = m.tex_kd, m.tex_norm, m.tex_ks
                mats[j].norm_tangent = int(m.norm_tangent)
            d = ModelDesc()
            d.vertices = pm.vertices.ctypes.data
            d.uv = pm.uv.ctypes.data if pm.uv is not None else None
            d.normals = pm.normals.ctypes.data if pm.normals is not None else None
            d.faces = pm.faces.ctypes.data
            d.materials = mats
            d.edge_ids = pm.edge_ids.ctypes.data if pm.edge_ids is not None else None
            d.n_vertices = len(pm.vertices)
            d.n_uv = 0 if pm.uv is None else len(pm.uv)
            d.n_normals = 0 if pm.normals is None else len(pm.normals)
            d.n_faces, d.n_materials = len(pm.faces), len(pm.materials)
            d.vertices_are_f32, d.clip, d.depth_test = int(pm.vertices_are_f32), int(pm.clip), int(pm.depth_test)
            _check(self.lib.mr_scene_add_model(self.handle, C.byref(d)), "mr_scene_add_model")
        self._signature = sig

    def sync_skybox(self, scene):
        sky = scene.skybox if hasattr(scene.skybox, "texels") else None
        key = None if sky is None else id(sky.texels)
        if key == self._sky_key:
            return
        if sky is None:
            _check(self.lib.mr_scene_set_skybox(self.handle, None, 0), "mr_scene_set_skybox")
        else:
            tex = np.ascontiguousarray(sky.texels, dtype=np.uint8)
            _check(self.lib.mr_scene_set_skybox(self.handle, tex.ctypes.data, tex.shape[1]), "mr_scene_set_skybox")
        self._sky_key = key

    def sync_overlay(self, scene):
        """Debug-frustum overlay (obj/core.py:638): the statement lists depend on the two cameras and the
        resolution only (a camera's MVP is cached for life), so they are built and uploaded once."""
        cam = scene.camera
        dbg = scene.debug_camera if scene.debug_camera is not None else cam
        key = (id(cam), id(dbg), id(cam.__dict__.get("MVP")), id(dbg.__dict__.get("MVP")), tuple(scene.resolution),
               int(scene.system), int(scene.subsystem))
        if getattr(self, "_overlay_key", None) == key or getattr(self, "_overlay_pinned", False):
            return
        # the key holds ids: keep the objects alive while it is cached, so that no other camera or matrix can
        # be given a recycled address and pass for them
        from .frustums import frustum_corners
        # the two inverses / vector products of the recipe stay with NumPy (obj/frustums.py:52-60); everything after
        # them -- clipping, projection, DDA, dashes, the per-pixel lists and their upload -- is one call into the library
        corners, inside = frustum_corners(cam, dbg)
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        corners, planes, mvp, viewport = f64(corners), f64(cam.frustum_planes), f64(cam.MVP), f64(cam.viewport)
        height, width = (int(v) for v in scene.resolution)
        _check(self.lib.mr_scene_set_overlay_cameras(self.handle, corners.ctypes.data, planes.ctypes.data, mvp.ctypes.data,
                                                     viewport.ctypes.data, float(cam.near), float(cam.far), int(inside),
                                                     height, width), "mr_scene_set_overlay_cameras")
        self._overlay_key = (id(cam), id(dbg), id(cam.__dict__.get("MVP")), id(dbg.__dict__.get("MVP")),
                             tuple(scene.resolution), int(scene.system), int(scene.subsystem))
        self._overlay_refs = (cam, dbg, cam.__dict__.get("MVP"), dbg.__dict__.get("MVP"))

    def set_overlay_lists(self, ops, pin=True):
        """``mr_scene_set_overlay`` with explicit statement lists (a ``frustums.OverlayOps``); ``sync_overlay`` builds
        the same inside the library.  While *pin*ned, frames drawn with the overlay keep these lists instead of
        rebuilding them from the scene's cameras (``pin=False`` hands the overlay back to the cameras)."""
        d = OverlayDesc()
        d.n_segments, d.n_points, d.height, d.width = len(ops.seg_first), ops.n_points, ops.height, ops.width
        d.seg_first, d.seg_count = ops.seg_first.ctypes.data, ops.seg_count.ctypes.data
        d.target, d.z = ops.target.ctypes.data, ops.z.ctypes.data
        _check(self.lib.mr_scene_set_overlay(self.handle, C.byref(d) if ops.n_points else None), "mr_scene_set_overlay")
        self._overlay_key = None
        self._overlay_pinned = ops is not None and pin

    # -- frames ---------------------------------------------------------------------------
    @staticmethod
    def _frame_key(scene, shadows):
        """Everything ``pack_frame`` reads, cheaply: the camera objects (their MVP is cached for life, as in
        the reference) and the values of the few scalars and 3-vectors a caller may have reassigned."""
        cam, light = scene.camera, scene.light
        dbg = scene.debug_camera if scene.debug_camera is not None else cam

        def vec(x):
            return tuple(np.asarray(x, dtype=np.float64).ravel().tolist())
        return (id(cam), id(dbg), id(cam.__dict__.get("MVP")), id(dbg.__dict__.get("MVP")), tuple(scene.resolution),
                int(scene.system), int(scene.subsystem), bool(shadows), bool(cam.backface_culling), float(cam.near),
                float(cam.far), cam.x_offset, cam.y_offset, vec(cam.position), id(light), vec(light.position),
                vec(light.center), vec(light.color), vec(light.ambient), str(light.light_type),
                float(light.specular_strength), float(light.constant), float(light.linear), float(light.quadratic),
                id(scene.skybox) if scene.skybox is None or hasattr(scene.skybox, "textures") else vec(scene.skybox))

    def packed_frame(self, scene, shadows):
        """``pack_frame`` with a one-entry cache: a render loop that changes nothing between two frames does
        not rebuild the matrices and planes (150 us of NumPy scalar arithmetic, more than the frame takes)."""
        key = self._frame_key(scene, shadows)
        if self._packed[0] != key or "MVP" not in scene.camera.__dict__:
            self._pack_serial = getattr(self, "_pack_serial", 0) + 1
            self._packed = (None, pack_frame(scene, shadows))
            self._packed = (self._frame_key(scene, shadows), self._packed[1])    # the MVPs exist (and are cached) now
            # the key identifies cameras, light and matrices by id(): hold them while the entry lives, so that
            # CPython cannot hand their addresses to the objects of a later frame
            cam = scene.camera
            dbg = scene.debug_camera if scene.debug_camera is not None else cam
            self._packed_refs = (cam, dbg, cam.__dict__.get("MVP"), dbg.__dict__.get("MVP"), scene.light, scene.skybox)
        return self._packed[1]

    def render(self, scene, shadows=True, row_band=None, keep_float=False, face_status=False, counters=True,
               keep_buffers=None, stripe=None, timing=True, overlay=False):
        """``mr_render``: returns the uint8 rows ``(rows, W, 3)`` as a NumPy array.

        ``counters=True`` (``MR_FRAME_COUNTERS``) also keeps the reference-equivalent fragment
        counters in ``last_stats`` and the reference's stencil values at uncovered pixels;
        ``keep_buffers`` (``MR_FRAME_KEEP_BUFFERS``, default: same as *counters*) also writes the
        z-buffer / stencil / winner map to device memory for ``read_z`` and friends -- what the
        parity tests and tools look at.  ``Scene.render`` passes ``False`` for both: only the
        frame is asked for, those buffers never leave the chip, and shadow quads that cannot pass
        the depth test are skipped (``last_stats`` then reports the fragment counters as -1).
        ``stripe=(index, count)`` renders the interleaved tile rows of one device of a
        multi-GPU split; the rows come back in the striped layout (``multigpu.unstripe``).
        ``timing=False`` (``MR_FRAME_NO_TIMING``) records no HIP events: each one costs a few
        microseconds between two kernels; ``last_stats['gpu_ms_*']`` are then 0.
        ``overlay=True`` (``MR_FRAME_OVERLAY``) draws the debug camera's frustum into the frame on the
        device, as the reference's ``render()`` always does (obj/core.py:638)."""
        desc, out = self._prepare(scene, shadows, row_band, keep_float, face_status, counters, keep_buffers, stripe, timing, overlay)
        stats = Stats()
        _check(self.lib.mr_render(self.handle, C.byref(desc), out.ctypes.data, C.byref(stats) if counters else None),
               "mr_render")
        # a frame rendered for the frame's sake fetches its statistics only if somebody looks at them
        self._last_stats = stats.as_dict() if counters else None
        return out

    def _prepare(self, scene, shadows, row_band, keep_float, face_status, counters, keep_buffers, stripe, timing, overlay):
        """Everything ``render`` / ``render_async`` do before the library call: scene, skybox and overlay in step
        with the Python objects, the frame descriptor, a page-locked output array."""
        self.sync_scene(scene)
        self.sync_skybox(scene)
        pf = self.packed_frame(scene, shadows)
        if overlay:
            self.sync_overlay(scene)
        if keep_buffers is None:
            keep_buffers = counters
        dkey = (getattr(self, "_pack_serial", 0), row_band, keep_float, face_status, counters, keep_buffers, stripe, timing,
                overlay)
        if getattr(self, "_desc_key", None) != dkey:
            self._desc_cached = fill_frame_desc(pf, row_band, keep_float, face_status=face_status, counters=counters,
                                                keep_buffers=keep_buffers, stripe=stripe, no_timing=not timing,
                                                overlay=overlay)
            self._desc_key = dkey
        desc = self._desc_cached
        self._n_faces = sum(len(m._faces) for m in scene.models)
        rows = desc.row_end - desc.row_begin if stripe is None else stripe_out_rows(pf.height, stripe[1])
        self._frame = (pf.height, pf.width)
        return desc, self._pinned.array((rows, pf.width, 3))

    ASYNC_LANES = 4

    def render_async(self, scene, lane, shadows=True, overlay=False):
        """``mr_render_async``: the frame-only render of ``Scene.render`` enqueued on *lane*; returns the (page-locked)
        array the frame will land in.  ``render_wait(lane)`` blocks until it has."""
        desc, out = self._prepare(scene, shadows, None, False, False, False, False, None, False, overlay)
        _check(self.lib.mr_render_async(self.handle, C.byref(desc), out.ctypes.data, int(lane)), "mr_render_async")
        self._last_stats = None
        return out

    def render_wait(self, lane):
        """True when the lane's frame is complete in its array; False when it overflowed a work list (the lists
        have been grown: render the frame again)."""
        rc = self.lib.mr_render_wait(self.handle, int(lane), None)
        if rc == MR_E_OVERFLOW:
            return False
        _check(rc, "mr_render_wait")
        return True

    @property
    def last_stats(self):
        if self._last_stats is None and self.handle:
            st = Stats()
            _check(self.lib.mr_get_stats(self.handle, C.byref(st)), "mr_get_stats")
            self._last_stats = st.as_dict()
        return self._last_stats

    @last_stats.setter
    def last_stats(self, value):
        self._last_stats = value

    def render_device(self, scene, d_out_ptr, stream_ptr=0, shadows=True, row_band=None, light_timing=False,
                      counters=False, stripe=None, no_timing=False, overlay=False):
        """``mr_render_device``: enqueue a frame whose uint8 band lands at device pointer *d_out_ptr*.  With *overlay*
        a device that renders the whole frame draws the debug frustum; one that renders part of it appends the state of
        the touched pixels it owns to its rows (``overlay_state_bytes``, ``overlay_apply``)."""
        self.sync_scene(scene)
        self.sync_skybox(scene)
        pf = self.packed_frame(scene, shadows)
        if overlay:
            self.sync_overlay(scene)
        desc = fill_frame_desc(pf, row_band, False, light_timing, counters=counters, stripe=stripe, no_timing=no_timing,
                               overlay=overlay)
        _check(self.lib.mr_render_device(self.handle, C.byref(desc), C.c_void_p(d_out_ptr),
                                         C.c_void_p(stream_ptr)), "mr_render_device")
        self._frame = (pf.height, pf.width)
        self._desc = desc
        return desc

    @staticmethod
    def with_timing(desc, mode):
        """Copy of a frame descriptor with its event marks set to *mode*: "all" stages, "light" (frame +
        tile kernel) or "none"."""
        out = FrameDesc.from_buffer_copy(desc)
        out.flags &= ~(FRAME_LIGHT_TIMING | FRAME_NO_TIMING)
        out.flags |= {"all": 0, "light": FRAME_LIGHT_TIMING, "none": FRAME_NO_TIMING}[mode]
        return out

    def enqueue(self, desc, d_out_ptr, stream_ptr=0):
        """Re-issue a prepared frame descriptor (no Python-side packing in the timed loop)."""
        _check(self.lib.mr_render_device(self.handle, C.byref(desc), C.c_void_p(d_out_ptr),
                                         C.c_void_p(stream_ptr)), "mr_render_device")

    def overlay_state_bytes(self):
        """Bytes a device of a split frame appends to its rows for the overlay (``mr_overlay_state_bytes``)."""
        return int(_check(self.lib.mr_overlay_state_bytes(self.handle), "mr_overlay_state_bytes"))

    def overlay_apply(self, d_parts_ptr, part_stride, state_offset, world, striped, system, d_frame_ptr, stream_ptr=0):
        """``mr_overlay_apply``: replay the overlay on the frame assembled from *world* parts."""
        _check(self.lib.mr_overlay_apply(self.handle, C.c_void_p(d_parts_ptr), int(part_stride), int(state_offset), int(world),
                                         int(bool(striped)), int(system), C.c_void_p(d_frame_ptr), C.c_void_p(stream_ptr)),
               "mr_overlay_apply")

    def set_list_capacities(self, small_pairs=0, big_pairs=0, quads=0, work=0):
        """Where the per-tile work lists start (they grow on overflow); a tuning / test hook."""
        _check(self.lib.mr_scene_set_list_capacities(self.handle, small_pairs, big_pairs, quads, work),
               "mr_scene_set_list_capacities")

    def overflowed(self):
        """True when a frame enqueued with ``render_device`` overflowed a work list (the lists have then
        been grown and the frame must be enqueued again); synchronises the device."""
        st = Stats()
        rc = self.lib.mr_get_stats(self.handle, C.byref(st))
        if rc == MR_E_OVERFLOW:
            return True
        _check(rc, "mr_get_stats")
        self.last_stats = st.as_dict()
        return False

    def stats(self):
        st = Stats()
        _check(self.lib.mr_get_stats(self.handle, C.byref(st)), "mr_get_stats")
        self.last_stats = st.as_dict()
        return self.last_stats

    KERNEL_TIME_NAMES = ("vertex_mfma", "setup", "bin_work", "tile", "frame")

    def kernel_times(self, n_frames):
        """Average per-stage device milliseconds over the last *n_frames* frames (syncs)."""
        buf = (C.c_float * 5)()
        n = _check(self.lib.mr_get_kernel_times(self.handle, int(n_frames), buf, 5), "mr_get_kernel_times")
        return dict(zip(self.KERNEL_TIME_NAMES, (float(v) for v in buf))), n

    def stream_kernel_times(self, stream_ptr, n_frames):
        """The same over the marked frames of one stream; returns (dict, frames averaged)."""
        buf = (C.c_float * 5)()
        n = _check(self.lib.mr_get_stream_kernel_times(self.handle, C.c_void_p(stream_ptr), int(n_frames), buf, 5),
                   "mr_get_stream_kernel_times")
        return dict(zip(self.KERNEL_TIME_NAMES, (float(v) for v in buf))), n

    # -- debug taps -----------------------------------------------------------------------
    def _tap(self, fn, dtype, extra=()):
        h, w = self._frame
        out = np.empty((h, w) + tuple(extra), dtype=dtype)
        _check(fn(self.handle, out.ctypes.data), fn.__name__)
        return out

    def read_z(self):
        return self._tap(self.lib.mr_read_z, np.float64)

    def read_stencil(self):
        return self._tap(self.lib.mr_read_stencil, np.int16)

    def read_winner(self):
        return self._tap(self.lib.mr_read_winner, np.int32)

    def read_frame_f32(self):
        return self._tap(self.lib.mr_read_frame_f32, np.float32, (3,))

    def read_tile_records(self):
        """Per-tile diagnostics of the tile kernel: (n_tiles, 12) uint32, see mi355rast.h."""
        n = _check(self.lib.mr_debug_read_tile_records(self.handle, (C.c_uint32 * TILE_RECORD_WORDS)(), 0),
                   "mr_debug_read_tile_records")
        out = np.empty((max(n, 1), TILE_RECORD_WORDS), dtype=np.uint32)
        _check(self.lib.mr_debug_read_tile_records(self.handle, out.ctypes.data, n), "mr_debug_read_tile_records")
        return out[:n]

    def clusters_culled(self):
        """Clusters of 64 faces the last frame's set-up dropped whole (counted only under MR_CLUSTER_CULL=count)."""
        return int(_check(self.lib.mr_debug_clusters_culled(self.handle), "mr_debug_clusters_culled"))

    def read_tile_order(self):
        """The order in which the last frame's tile kernel took its tiles: (n_tiles,) uint32."""
        n = _check(self.lib.mr_debug_read_tile_order(self.handle, (C.c_uint32 * 1)(), 0), "mr_debug_read_tile_order")
        out = np.empty(max(n, 1), dtype=np.uint32)
        _check(self.lib.mr_debug_read_tile_order(self.handle, out.ctypes.data, n), "mr_debug_read_tile_order")
        return out[:n]

    def read_face_status(self):
        """One ``Errors`` value (0 = rendered) per face, models concatenated in scene order."""
        out = np.empty(max(self._n_faces, 1), dtype=np.uint8)
        _check(self.lib.mr_read_face_status(self.handle, out.ctypes.data), "mr_read_face_status")
        return out[:self._n_faces]

    def read_silhouette(self):
        n = _check(self.lib.mr_read_silhouette(self.handle, None, 0), "mr_read_silhouette")
        out = np.empty((max(n, 1), 3), dtype=np.int32)
        _check(self.lib.mr_read_silhouette(self.handle, out.ctypes.data, n), "mr_read_silhouette")
        return out[:n]

    def close(self):
        if self.handle:
            self.lib.mr_scene_destroy(self.handle)
            self.handle = None
            self._pinned.close()

    def __del__(self):  # best effort
        try:
            self.close()
        except Exception:
            pass
