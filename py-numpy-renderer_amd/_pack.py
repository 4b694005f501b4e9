"""Scene -> flat arrays.

Turns the ``Scene`` object graph into the plain arrays and scalars the C ABI takes
(``include/mi355rast.h``): per-frame constants in float64 and, per model, the vertex /
uv / normal / index arrays with every index made non-negative and every material group
resolved to a small record.  Pure host bookkeeping; no rasterisation happens here.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from .lightning import Lightning

_DEFAULT_BACKGROUND = (64 / 255, 0.5, 198 / 255)       # obj/core.py:600


@dataclass
class PackedMaterial:
    kd: np.ndarray
    ks255: np.ndarray
    ns: float
    tex_kd: int = -1
    tex_norm: int = -1
    tex_ks: int = -1
    norm_tangent: bool = False


@dataclass
class PackedModel:
    vertices: np.ndarray            # float64 (V, 4)
    uv: Optional[np.ndarray]        # float32 (T, 3)
    normals: Optional[np.ndarray]   # float32 (N, 3)
    faces: np.ndarray               # int32 (F, 3, 4), all >= 0
    materials: List[PackedMaterial]
    vertices_are_f32: bool
    clip: bool
    depth_test: bool
    # (F, 3) int32: the vertex column of Model._faces as loaded (negative = relative).  The reference's
    # silhouette set identifies an edge by these raw values (obj/triangular.py:286-302); None when they
    # equal faces[..., 0]
    edge_ids: Optional[np.ndarray] = None


@dataclass
class PackedFrame:
    width: int
    height: int
    system: int
    backface_culling: bool
    light_type: int
    shadows: bool
    mvp: np.ndarray
    viewport: np.ndarray
    debug_mvp: np.ndarray
    frustum_planes: np.ndarray
    z_near: float
    z_far: float
    camera_pos: np.ndarray
    light_pos: np.ndarray
    light_dir: np.ndarray
    light_color: np.ndarray
    light_ambient: np.ndarray
    specular_strength: float
    att_constant: float
    att_linear: float
    att_quadratic: float
    spot_edge0: float
    spot_edge1: float
    background: np.ndarray
    sky_tri: Optional[np.ndarray] = None      # (2, 3, 2) int32, cubemap skybox only
    sky_rays: Optional[np.ndarray] = None     # (2, 3, 3) float64


@dataclass
class PackedScene:
    frame: PackedFrame
    models: List[PackedModel]
    textures: List[np.ndarray] = field(default_factory=list)   # float32 (h, w, 3), C order


def _vec3(x):
    v = np.asarray(x, dtype=np.float64).ravel()
    if v.size == 1:
        v = np.repeat(v, 3)
    if v.size != 3:
        raise ValueError(f"expected a 3-vector, got shape {np.shape(x)}")
    return np.ascontiguousarray(v)


def _wrap(idx, n, what):
    """Python-style negative indices -> non-negative, bounds-checked."""
    idx = np.where(idx < 0, idx + n, idx)
    if idx.size and (idx.min() < 0 or idx.max() >= n):
        raise IndexError(f"{what} index out of range for {n} entries")
    return idx


def pack_frame(scene, shadows=True) -> PackedFrame:
    cam, light = scene.camera, scene.light
    dbg = scene.debug_camera if scene.debug_camera is not None else cam
    height, width = (int(v) for v in scene.resolution)
    sky = scene.skybox
    sky_tri = sky_rays = None
    if sky is not None and hasattr(sky, "textures"):
        from .cube_map import sky_frame_constants
        sky_tri, sky_rays = sky_frame_constants(cam)
        background = np.zeros(3, dtype=np.float32)          # uncovered pixels stay black (frame starts at 0)
    elif sky is not None:
        background = np.asarray(np.array(sky), dtype=np.float32).ravel()
        if background.size != 3:
            raise ValueError("skymap colour must have 3 components")
    else:
        background = np.asarray(_DEFAULT_BACKGROUND, dtype=np.float32)
    kind = light.light_type.value if isinstance(light.light_type, Lightning) else int(light.light_type)
    return PackedFrame(
        width=width, height=height, system=int(scene.system),
        backface_culling=bool(cam.backface_culling), light_type=kind, shadows=bool(shadows),
        mvp=np.ascontiguousarray(cam.MVP, dtype=np.float64),
        viewport=np.ascontiguousarray(cam.viewport, dtype=np.float64),
        debug_mvp=np.ascontiguousarray(dbg.MVP, dtype=np.float64),
        frustum_planes=np.ascontiguousarray(cam.frustum_planes, dtype=np.float64),
        z_near=float(cam.near), z_far=float(cam.far),
        camera_pos=_vec3(cam.position),
        light_pos=_vec3(light.position), light_dir=_vec3(light.direction),
        light_color=_vec3(light.color), light_ambient=_vec3(light.ambient),
        specular_strength=float(light.specular_strength),
        att_constant=float(light.constant), att_linear=float(light.linear),
        att_quadratic=float(light.quadratic),
        spot_edge0=float(np.cos(np.deg2rad(20))), spot_edge1=float(np.cos(np.deg2rad(10))),
        background=background, sky_tri=sky_tri, sky_rays=sky_rays)


def _texture_id(tex, textures, seen):
    key = id(tex)
    if key not in seen:
        arr = np.asarray(tex)
        if arr.ndim != 3 or arr.shape[2] < 3:
            raise ValueError(f"texture must be (h, w, 3), got {arr.shape}")
        seen[key] = len(textures)
        textures.append(np.ascontiguousarray(arr[..., :3], dtype=np.float32))
    return seen[key]


def pack_model(model, textures, seen) -> PackedModel:
    verts = np.asarray(model.vertices)
    if verts.ndim != 2 or verts.shape[1] != 4:
        raise ValueError(f"Model.vertices must be (V, 4), got {verts.shape}")
    faces = np.asarray(model._faces)
    if faces.ndim != 3 or faces.shape[1:] != (3, 4):
        raise ValueError("Model._faces must be (F, 3, 4): every face corner needs v/vt/vn indices "
                         f"(got {faces.shape})")
    uv = None if model.uv is None else np.ascontiguousarray(model.uv, dtype=np.float32)
    normals = None if model.normals is None else np.ascontiguousarray(model.normals, dtype=np.float32)
    if uv is not None and (uv.ndim != 2 or uv.shape[1] < 2):
        raise ValueError(f"Model.uv must be (T, 3), got {uv.shape}")
    if uv is not None and uv.shape[1] != 3:
        uv = np.ascontiguousarray(np.pad(uv[:, :3], ((0, 0), (0, 3 - min(uv.shape[1], 3)))))

    groups = list(model.material_group)
    mats = []
    for g in range(len(groups)):
        mat = model.face_material(g)
        ks = np.asarray(mat.Ks)
        rec = PackedMaterial(kd=_vec3(mat.Kd), ks255=_vec3(ks * 255), ns=float(mat.Ns))
        if hasattr(mat, "map_Kd"):
            rec.tex_kd = _texture_id(mat.map_Kd, textures, seen)
        if hasattr(mat, "norm"):
            rec.tex_norm = _texture_id(mat.norm, textures, seen)
            rec.norm_tangent = mat.is_tangent_space("norm")
        if hasattr(mat, "map_Ks"):
            rec.tex_ks = _texture_id(mat.map_Ks, textures, seen)
        mats.append(rec)
        if (rec.tex_kd >= 0 or rec.tex_norm >= 0 or rec.tex_ks >= 0) and uv is None:
            raise ValueError("model has texture maps but no uv coordinates")
        if rec.norm_tangent and normals is None:
            raise ValueError("tangent-space normal map needs vertex normals")

    out = np.empty(faces.shape, dtype=np.int32)
    out[..., 0] = _wrap(faces[..., 0].astype(np.int64), len(verts), "vertex")
    out[..., 1] = _wrap(faces[..., 1].astype(np.int64), len(uv), "uv") if uv is not None else 0
    out[..., 2] = _wrap(faces[..., 2].astype(np.int64), len(normals), "normal") if normals is not None else 0
    out[..., 3] = _wrap(faces[..., 3].astype(np.int64), len(groups), "material group")
    raw = faces[..., 0].astype(np.int64)
    edge_ids = np.ascontiguousarray(raw, dtype=np.int32) if (raw < 0).any() else None
    return PackedModel(
        vertices=np.ascontiguousarray(verts, dtype=np.float64), uv=uv, normals=normals,
        faces=np.ascontiguousarray(out), materials=mats,
        vertices_are_f32=(verts.dtype == np.float32),
        clip=bool(model.clip), depth_test=bool(model.depth_test), edge_ids=edge_ids)


def pack_scene(scene, shadows=True) -> PackedScene:
    textures, seen = [], {}
    models = [pack_model(m, textures, seen) for m in scene.models]
    return PackedScene(frame=pack_frame(scene, shadows), models=models, textures=textures)
