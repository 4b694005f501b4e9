"""Cubemap skybox (reference: ``obj/cube_map.py``).

``CubeMap`` loads six face images into one ``(6, S, S, 3)`` stack, each face pre-flipped /
rotated so that a direction's major axis and sign select the face and the other two components
its texel (``obj/cube_map.py:24-34, 63-80``).  ``Scene(skymap=CubeMap(...))`` fills the
background from it: two screen-covering triangles, float32 barycentrics of *integer-cast*
vertices, view rays interpolated from the triangles' un-projected corners, nearest texel
(``obj/cube_map.py:83-101``).  The fill itself runs in the device's shading kernel; this module
loads the images and derives the per-frame constants (``sky_frame_constants``).
"""
import numpy as np
from PIL import Image

from . import _fp

# the two NDC triangles the reference fills the screen with (obj/cube_map.py:45-54)
SKY_TRIANGLES = (np.array([[-1, 1, 1, 1], [1, 1, 1, 1], [-1, -1, 1, 1]]),
                 np.array([[1, 1, 1, 1], [1, -1, 1, 1], [-1, -1, 1, 1]]))


class CubeMap:
    def __init__(self, left, right, top, bottom, front, back, normalize_input=True):
        load = self.load_texels
        if normalize_input:
            faces = [np.flip(load(right), axis=(0, 1)),
                     np.rot90(load(left).transpose((1, 0, 2)), -1),
                     load(top).transpose((1, 0, 2)),
                     np.rot90(load(bottom)),
                     np.rot90(load(front), -1),
                     load(back).transpose((1, 0, 2))]
        else:
            faces = [load(right), load(left), load(top), load(bottom), load(front), load(back)]
        self.texels = np.ascontiguousarray(np.stack(faces))          # uint8 (6, S, S, 3), what the device gets
        if self.texels.ndim != 4 or self.texels.shape[1] != self.texels.shape[2]:
            raise ValueError("cubemap faces must be square and of equal size")
        self.textures = self.texels / 255                             # float64, as the reference keeps them
        self.faces = [tri.copy() for tri in SKY_TRIANGLES]

    @staticmethod
    def load_texels(name):
        with Image.open(name) as img:
            return np.asarray(img)[..., :3].copy()

    @staticmethod
    def load_texture(name):
        return CubeMap.load_texels(name) / 255

    def __getitem__(self, vectors):
        """Texels seen along direction *vectors* (N,3): host restatement of the lookup the
        kernel performs (kept for API parity and used by the tests)."""
        vectors = np.asarray(vectors, dtype=np.float64)
        rows = np.arange(len(vectors))
        major = np.abs(vectors).argmax(axis=1)
        amp = vectors[rows, major]
        keep = np.ones(vectors.shape, dtype=bool)
        keep[rows, major] = False
        uv = (vectors[keep].reshape(len(vectors), 2) / amp[:, None] + 1) / 2
        side = (amp < 0) + major * 2
        size = self.textures.shape[1]
        ij = (uv.T * size - 1).astype(int)
        return self.textures[side.astype(int), ij[0], ij[1]]


def sky_frame_constants(camera):
    """Per-frame constants of the skybox fill: for each of the two triangles its screen
    vertices truncated to int (``obj/cube_map.py:88-89``) and the three un-projected corner rays
    ``face @ inv(view_without_translation @ projection)`` divided by w (``:95-98``)."""
    view = np.array(camera.lookat, dtype=np.float64, copy=True)
    view[3, :3] = 0
    unproject = np.linalg.inv(_fp.matmul_chain(view, camera.projection))
    viewport = camera.viewport
    tri_px = np.empty((2, 3, 2), dtype=np.int32)
    rays = np.empty((2, 3, 3), dtype=np.float64)
    for t, face in enumerate(SKY_TRIANGLES):
        screen = _fp.matmul_chain(face, viewport)
        tri_px[t] = screen[:, :2].astype(int)
        r = _fp.matmul_chain(face, unproject)
        r = r / r[:, [3]]
        rays[t] = r[:, :3]
    return tri_px, rays
