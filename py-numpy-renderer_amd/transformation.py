"""Camera / model transformation builders and small geometric helpers.

Same names and argument meaning as the reference's ``obj/transformation.py`` (row-vector
convention: ``clip = v @ MVP``).  Every builder returns float64 unless noted.
"""
import math

import numpy as np

from .constants import X, Y, SYSTEM, SUBSYSTEM, PROJECTION_TYPE, mat3x3  # noqa: F401 (re-exported)


def normalize(a, axis=-1, order=2):
    """``a / |a|`` along *axis*; zero vectors are returned unchanged
    (reference: ``obj/transformation.py:46-49``)."""
    a = np.asarray(a)
    length = np.atleast_1d(np.linalg.norm(a, order, axis))
    length[length == 0] = 1
    return a / np.expand_dims(length, axis)


def barycentric(a, b, c, p):
    """Screen-space barycentrics of integer sample points *p* (N,2) in triangle *abc*.

    The five dot products are formed in float64 and rounded to float32; everything after
    that is float32 arithmetic (reference: ``obj/transformation.py:12-32``).  Returns
    ``None`` for a degenerate triangle, else float32 (N,3) ``(u, v, w)``.
    This NumPy form is kept for API compatibility; the renderer itself evaluates the same
    formula inside the HIP visibility kernel.
    """
    e0, e1, rel = b - a, c - a, p - a
    d00, d01, d11 = (np.float32(e0 @ e0), np.float32(e0 @ e1), np.float32(e1 @ e1))
    d20, d21 = np.float32(rel @ e0), np.float32(rel @ e1)
    den = d00 * d11 - d01 * d01
    if den == 0:
        return None
    inv = np.float32(1.0) / den
    v = (d11 * d20 - d01 * d21) * inv
    w = (d00 * d21 - d01 * d20) * inv
    u = np.float32(1.0) - v - w
    return np.stack((u, v, w), axis=-1)


def bound_box(vert, height, width):
    """Screen-clamped, ceil'd bounding box ``(min_x, max_x, min_y, max_y)`` as int32, or
    ``None`` when empty (reference: ``obj/transformation.py:35-43``).  The pixel set of a
    primitive is the half-open product ``[min_x, max_x) x [min_y, max_y)``."""
    xs, ys = vert[X], vert[Y]
    lo_x, hi_x = max(xs.min(), 0), min(xs.max(), width)
    lo_y, hi_y = max(ys.min(), 0), min(ys.max(), height)
    if lo_x > hi_x or lo_y > hi_y:
        return None
    return np.ceil((lo_x, hi_x, lo_y, hi_y)).astype(np.int32)


# --------------------------------------------------------------------------- view matrices
def looka_at_translate(eye):
    m = np.eye(4)
    m[3, :3] = -np.asarray(eye, dtype=np.float64)
    return m


def _unit3(v):
    """``normalize`` of one 3-vector with scalar arithmetic: the same operations NumPy performs (squares, a
    left-to-right sum, sqrt, three divisions), a dozen array calls fewer -- this runs for every frame whose camera moved."""
    x, y, z = float(v[0]), float(v[1]), float(v[2])
    length = math.sqrt((x * x + y * y) + z * z)
    if length == 0:
        length = 1.0
    return np.array((x / length, y / length, z / length))


def _cross3(a, b):
    """``np.cross`` of two 3-vectors, component by component as NumPy does (two rounded products, one subtraction)."""
    a0, a1, a2 = float(a[0]), float(a[1]), float(a[2])
    b0, b1, b2 = float(b[0]), float(b[1]), float(b[2])
    return np.array((a1 * b2 - a2 * b1, a2 * b0 - a0 * b2, a0 * b1 - a1 * b0))


def _look_at_axes(eye, center, up):
    forward = _unit3(np.asarray(center, dtype=np.float64) - np.asarray(eye, dtype=np.float64))
    right = _unit3(_cross3(up, forward))
    return right, _cross3(forward, right), forward


def look_at_rotate_lh(eye, center, up):
    right, new_up, forward = _look_at_axes(eye, center, up)
    m = np.eye(4)
    m[mat3x3] = np.column_stack((right, new_up, -forward))
    return m


def look_at_rotate_rh(eye, center, up):
    right, new_up, forward = _look_at_axes(eye, center, up)
    m = np.eye(4)
    m[mat3x3] = np.column_stack((right, new_up, forward))
    return m


def ViewPort(resolution, far, near, x_offset=0, y_offset=0):
    """NDC -> screen; *resolution* is ``(height, width)`` (``obj/transformation.py:123-136``)."""
    height, width = resolution
    half_w, half_h, half_d = width / 2, height / 2, (far - near) / 2
    return np.array([[half_w, 0, 0, 0],
                     [0, half_h, 0, 0],
                     [0, 0, half_d, 0],
                     [half_w + x_offset, half_h + y_offset, half_d, 1]])


# --------------------------------------------------------------------------- projections
def _perspective(fovy, aspect, m22, m32, m23):
    t = 1.0 / np.tan(np.radians(fovy) / 2.0)
    m = np.zeros((4, 4))
    m[0, 0], m[1, 1] = t / aspect, t
    m[2, 2], m[3, 2], m[2, 3] = m22, m32, m23
    return m


def opengl_perspectiveLH(fovy, aspect, z_near, z_far):
    return _perspective(fovy, aspect, -(z_far + z_near) / (z_far - z_near),
                        2.0 * z_far * z_near / (z_far - z_near), 1.0)


def opengl_perspectiveRH(fovy, aspect, z_near, z_far):
    return _perspective(fovy, aspect, -(z_far + z_near) / (z_far - z_near),
                        -2.0 * z_far * z_near / (z_far - z_near), -1.0)


def directx_perspectiveRH(fovy, aspect, z_near, z_far):
    return _perspective(fovy, aspect, z_far / (z_near - z_far),
                        z_near * z_far / (z_near - z_far), -1.0)


def directx_perspectiveLH(fovy, aspect, z_near, z_far):
    return _perspective(fovy, aspect, -z_far / (z_far - z_near),
                        z_near * z_far / (z_far - z_near), 1.0)


def opengl_orthographicLH(fov, aspect_ratio, z_near, z_far):
    """float32, as in the reference (``obj/transformation.py:139-154``)."""
    top = np.tan(np.radians(fov / 2.0)) * z_near
    right = top * aspect_ratio
    span = z_far - z_near
    return np.array([[1 / right, 0, 0, 0],
                     [0, 1 / top, 0, 0],
                     [0, 0, -2 / span, 0],
                     [0, 0, (z_far + z_near) / span, 1]], dtype=np.float32)


# --------------------------------------------------------------------------- model transforms
def scale(factor):
    return np.diag([factor, factor, factor, 1])


def translation(vec):
    x, y, z = vec
    return np.array([[1, 0, 0, 0],
                     [0, 1, 0, 0],
                     [0, 0, 1, 0],
                     [x, y, z, 1]])


def rotate_xyz(a):
    """Rotation by Euler angles given in degrees; float32 like the reference, and with the
    reference's axis naming kept as is: ``a[0]`` turns about y, ``a[1]`` about x, ``a[2]``
    about z, composed ``Rz @ Ry @ Rx`` in row-vector form (``obj/transformation.py:230-263``)."""
    about_y, about_x, about_z = np.deg2rad(a)
    cx, sx = np.cos(about_x), np.sin(about_x)
    cy, sy = np.cos(about_y), np.sin(about_y)
    cz, sz = np.cos(about_z), np.sin(about_z)
    rx = np.array([[1, 0, 0, 0], [0, cx, sx, 0], [0, -sx, cx, 0], [0, 0, 0, 1]], dtype=np.float32)
    ry = np.array([[cy, 0, -sy, 0], [0, 1, 0, 0], [sy, 0, cy, 0], [0, 0, 0, 1]], dtype=np.float32)
    rz = np.array([[cz, -sz, 0, 0], [sz, cz, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)
    return rz @ ry @ rx


perspectives = {
    SUBSYSTEM.OPENGL: {
        PROJECTION_TYPE.PERSPECTIVE: {SYSTEM.RH: opengl_perspectiveRH, SYSTEM.LH: opengl_perspectiveLH},
        PROJECTION_TYPE.ORTHOGRAPHIC: {SYSTEM.LH: opengl_orthographicLH},
    },
    SUBSYSTEM.DIRECTX: {
        PROJECTION_TYPE.PERSPECTIVE: {SYSTEM.RH: directx_perspectiveRH, SYSTEM.LH: directx_perspectiveLH},
    },
}
