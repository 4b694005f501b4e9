"""Debug-camera frustum overlay (host side).

The reference ends every ``render()`` by drawing the *debug camera's* view frustum as red,
z-tested lines into the float frame and the z-buffer (``obj/core.py:638``,
``obj/frustums.py:46-103``, ``obj/line.py:6-16``).  It is a debugging aid whose result depends
on the order of its own writes (each line segment is tested against z values earlier segments
left behind), a few thousand pixels in all, so it is applied on the host to the buffers the
device produced: ``Scene.render`` downloads the float frame and z-buffer only when
``scene.draw_debug_frustum`` is set (off by default; the reference cannot switch it off).
"""
import numpy as np

from .constants import W_COL, X, XY, XYZ, Y, Z, W
from .plane_intersection import clipping

# the clip-space cube and its six faces (vertex order matters: it fixes the line order)
CUBE = np.array([[-1.0, -1.0, 1.0, 1.0], [1.0, -1.0, 1.0, 1.0], [-1.0, 1.0, 1.0, 1.0], [1.0, 1.0, 1.0, 1.0],
                 [-1.0, 1.0, -1.0, 1.0], [1.0, 1.0, -1.0, 1.0], [-1.0, -1.0, -1.0, 1.0], [1.0, -1.0, -1.0, 1.0]])
CUBE_FACES = np.array([(2, 4, 5, 3), (0, 1, 7, 6), (0, 2, 3, 1), (5, 4, 6, 7), (3, 5, 7, 1), (4, 2, 0, 6)])
DASH = 13                 # pixels per dash of a hidden edge
RED = np.array((1.0, 0.0, 0.0))


def bresenham_line(start_point, end_point):
    """Points of a DDA walk from *start* to *end* (4-vectors: x, y, z, w), one per unit step
    along the longer screen axis, always walked towards decreasing x; the end point itself is
    not included (``obj/line.py:6-16``)."""
    delta = end_point - start_point
    if delta[X] > 0:
        return bresenham_line(end_point, start_point)
    steps = max(abs(delta[XY]))
    if steps == 0:
        return start_point[None]
    return start_point + np.arange(int(steps))[:, None] * (delta / steps)


def draw_view_frustum(frame, camera, positioned_object, z_buffer, sign):
    """Draw *positioned_object*'s frustum as seen by *camera* into ``frame`` / ``z_buffer`` in place."""
    corners = CUBE @ np.linalg.inv(positioned_object.MVP)
    corners /= corners[W_COL]
    planes = camera.frustum_planes
    probe = np.append(camera.position, 1) @ positioned_object.MVP
    camera_inside = all(-probe[3] < probe[k] < probe[3] for k in range(3))
    near_far = 2 * camera.near * camera.far
    last_row, last_col = np.array(camera.scene.resolution) - 1

    for quad in corners[CUBE_FACES]:
        poly = clipping(quad, planes)
        if poly.shape[0] < 3:
            continue
        poly = poly @ camera.MVP
        poly /= poly[W_COL]
        poly = poly @ camera.viewport
        a, b, c = poly[XYZ][:3]
        facing = np.cross(b - a, c - a)[2]
        poly[Z] = near_far / (camera.far + camera.near - poly[Z] * (camera.far - camera.near))
        count = len(poly)
        for i in range(count):
            pts = bresenham_line(poly[i], poly[(i + 1) % count])
            if facing > 0 and not camera_inside:                      # hidden edge: dashed
                pts = pts[((np.arange(len(pts)) // DASH) & 1).astype(bool)]
            col, row, z, _ = pts.T
            row = row.astype(np.int32) - 1
            col = col.astype(np.int32) - 1
            keep = (z_buffer[row, col] - z) * sign >= 0
            row, col, z = row[keep], col[keep], z[keep]
            z_buffer[row, col] = z
            frame[row, col] = RED
            for step in (-1, 1):                                      # one-pixel soft edge
                r2 = np.clip(row + step, a_min=0, a_max=last_row)
                c2 = np.clip(col + step, a_min=0, a_max=last_col)
                z_buffer[r2, col] = z
                z_buffer[row, c2] = z
                frame[r2, col] = frame[r2, col] * 0.5 + RED / 2
                frame[row, c2] = frame[row, c2] * 0.5 + RED / 2
