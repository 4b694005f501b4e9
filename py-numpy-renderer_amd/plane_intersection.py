"""Frustum planes and polygon clipping (host side).

``extract_frustum_planes`` feeds the per-frame constant block the device's shadow-quad
set-up kernel clips against; ``clipping`` is the same Sutherland-Hodgman walk as the kernel,
kept here for API parity with the reference's ``obj/plane_intersection.py:39-86``.
"""
import numpy as np

from ._fp import dot_chain, matmul_chain

left, right, bottom, top, near, far = range(6)


def normalize_plane(plane):
    """Divide by the 4-norm, the squared norm formed as a dot product (fma chain)."""
    return plane / np.sqrt(dot_chain(plane, plane))


def extract_frustum_planes(matrix):
    """Six planes ``(a, b, c, d)`` in the order left, right, bottom, top, near, far from a
    row-vector MVP: ``col3 +- col0/1/2``, each normalised (``plane_intersection.py:43-56``)."""
    m = np.asarray(matrix, dtype=np.float64)
    w = m[:, 3]
    planes = np.empty((6, 4))
    for axis in range(3):
        planes[2 * axis] = normalize_plane(w + m[:, axis])
        planes[2 * axis + 1] = normalize_plane(w - m[:, axis])
    return planes


def is_visible(point, plane):
    return dot_chain(plane, point) >= 0


def line_plane_intersection(line_point1, line_point2, plane_coefficients):
    direction = line_point2 - line_point1
    denominator = dot_chain(plane_coefficients, direction)
    if abs(denominator) < 1e-10:
        return None
    weight = -dot_chain(plane_coefficients, line_point1) / denominator
    if 0 <= weight <= 1:
        return line_point1 + weight * direction
    return None


def clipping(polygon_vertices, clipping_planes):
    """Clip a convex polygon (k,4) against each plane in turn; returns an (m,4) array.

    The same walk as ``is_visible`` / ``line_plane_intersection`` vertex by vertex (the reference's
    ``obj/plane_intersection.py:59-86``), with a plane's dot products taken for the whole polygon at once
    (same fma chains, one call)."""
    polygon = np.array([np.asarray(v, dtype=np.float64) for v in polygon_vertices], dtype=np.float64).reshape(-1, 4)
    for plane in clipping_planes:
        count = len(polygon)
        if count == 0:
            break
        plane = np.asarray(plane, dtype=np.float64)
        dist = matmul_chain(polygon, plane.reshape(4, 1))[:, 0]          # dot_chain(plane, vertex) per vertex
        inside = dist >= 0
        nxt = np.arange(1, count + 1)
        nxt[-1] = 0
        following = polygon[nxt]
        crossing = inside != inside[nxt]
        kept = []
        if crossing.any():
            direction = polygon - following                              # line_plane_intersection(following, current, plane)
            den = matmul_chain(direction, plane.reshape(4, 1))[:, 0]
            dist_following = dist[nxt]
        for i in range(count):
            if inside[i]:
                kept.append(polygon[i])
            if crossing[i] and not abs(den[i]) < 1e-10:
                weight = -dist_following[i] / den[i]
                if 0 <= weight <= 1:
                    kept.append(following[i] + weight * direction[i])
        polygon = np.array(kept, dtype=np.float64).reshape(-1, 4)
    return polygon
