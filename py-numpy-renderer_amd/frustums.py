"""Debug-camera frustum overlay.

The reference ends every ``render()`` by drawing the *debug camera's* view frustum as red,
z-tested lines into the float frame and the z-buffer (``obj/core.py:638``,
``obj/frustums.py:46-103``, ``obj/line.py:6-16``).  The geometry -- clipping the frustum's six
faces, projecting them, walking their edges with a DDA, dashing the hidden ones -- is a few
thousand points of sequential float64 arithmetic and stays on the host (``overlay_segments``).
What the lines DO to the frame depends on the frame's own z-buffer and on the order of their
writes (each segment is tested against the z values earlier segments left behind), so that part
runs where the z-buffer is: ``OverlayOps`` flattens the segments into the statement lists the
device replays (``k_overlay`` in ``csrc/kernels_overlay.h``) right after the tile kernel, and
``draw_view_frustum`` replays the same lists on NumPy buffers (the CPU tests use it on the oracle's).
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


def frustum_corners(camera, positioned_object):
    """World-space corners of *positioned_object*'s frustum (8 x 4, divided by w) and whether *camera* sits inside
    it (``obj/frustums.py:52-60``): the part of the recipe that stays with NumPy (a matrix inverse)."""
    corners = CUBE @ np.linalg.inv(positioned_object.MVP)
    corners /= corners[W_COL]
    probe = np.append(camera.position, 1) @ positioned_object.MVP
    return corners, all(-probe[3] < probe[k] < probe[3] for k in range(3))


def overlay_segments(camera, positioned_object):
    """The line segments of *positioned_object*'s frustum as seen by *camera*, in drawing order: a
    list of ``(row, col, z)`` arrays (int32, int32, float64; row = screen y, not flipped), one per
    polygon edge (``obj/frustums.py:46-91``)."""
    corners = CUBE @ np.linalg.inv(positioned_object.MVP)
    corners /= corners[W_COL]
    planes = camera.frustum_planes
    probe = np.append(camera.position, 1) @ positioned_object.MVP
    camera_inside = all(-probe[3] < probe[k] < probe[3] for k in range(3))
    near_far = 2 * camera.near * camera.far
    segments = []
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
            segments.append((row.astype(np.int32) - 1, col.astype(np.int32) - 1, np.ascontiguousarray(z, dtype=np.float64)))
    return segments


class OverlayOps:
    """The overlay as flat statement lists (what ``mr_scene_set_overlay`` takes).

    Every kept point of a segment performs, in this order (``obj/frustums.py:93-103``): the centre write
    (z and RED), then for step -1 and +1: z into the row neighbour, z into the column neighbour, a
    half blend into the row neighbour, a half blend into the column neighbour -- each a NumPy fancy
    assignment over all kept points of the segment, i.e. the right-hand side is read before anything
    is written and, where several points hit one pixel, the last one wins.  Five target sets per
    point (centre, row-1, col-1, row+1, col+1 with the neighbours clipped to the frame), and per
    target set the link to the next point of the same segment with the same target, which is what
    "the last one wins" needs on a machine that writes them all at once."""

    N_TARGETS = 5

    def __init__(self, camera, positioned_object, resolution, native=None):
        """*native*: build the lists with the library's host helper (``mr_host_overlay_build``: the same arithmetic
        in C++, ten times faster) -- None: if the library is there; False: the NumPy walk below, which is the
        statement of what the lists are (``tests/test_overlay.py`` holds the two equal on random camera pairs)."""
        height, width = (int(v) for v in resolution)
        self.height, self.width = height, width
        if native is None or native:
            if self._build_native(camera, positioned_object, height, width):
                return
            if native:
                raise RuntimeError("the HIP library (its host helpers) is not available")
        rows, cols, zs, counts = [], [], [], []
        for row, col, z in overlay_segments(camera, positioned_object):
            if len(z) == 0:
                continue
            # Python / NumPy index semantics: a negative index counts from the end; anything else out of
            # range raises IndexError in the reference (there the frame is lost; here the point is)
            ok = (row >= -height) & (row < height) & (col >= -width) & (col < width)
            if not ok.all():
                row, col, z = row[ok], col[ok], z[ok]
            if len(z) == 0:
                continue
            rows.append(row); cols.append(col); zs.append(z); counts.append(len(z))
        if not counts:
            self.seg_first = self.seg_count = np.zeros(0, np.int32)
            self.target = self.next = np.zeros((self.N_TARGETS, 0), np.int32)
            self.z = np.zeros(0, np.float64)
            self.touched = np.zeros(0, np.int32)
            return
        # all segments at once from here on (a few array operations instead of dozens per segment)
        row = np.concatenate(rows).astype(np.int64)
        col = np.concatenate(cols).astype(np.int64)
        self.z = np.ascontiguousarray(np.concatenate(zs), dtype=np.float64)
        self.seg_count = np.asarray(counts, np.int32)
        self.seg_first = (np.cumsum(self.seg_count) - self.seg_count).astype(np.int32)
        n = len(row)
        segment = np.repeat(np.arange(len(counts), dtype=np.int64), self.seg_count)
        wrow, wcol = row % height, col % width                          # the centre index wraps like Python's
        # the neighbours are clipped from the RAW index (np.clip(x + i, 0, last)), so a point at -1
        # wraps to the last row but its neighbours are row 0 (obj/frustums.py:97-103)
        sets = ((wrow, wcol), (np.clip(row - 1, 0, height - 1), wcol), (wrow, np.clip(col - 1, 0, width - 1)),
                (np.clip(row + 1, 0, height - 1), wcol), (wrow, np.clip(col + 1, 0, width - 1)))
        self.target = np.empty((self.N_TARGETS, n), np.int32)
        self.next = np.full((self.N_TARGETS, n), -1, np.int32)
        for k, (r, c) in enumerate(sets):
            lin = r * width + c
            self.target[k] = lin
            # next point (later in the SAME segment) with the same target: in a stable sort by (segment, target)
            # the points of one target of one segment follow each other in segment order
            key = segment * (height * width) + lin
            order = np.argsort(key, kind="stable")
            same = key[order[1:]] == key[order[:-1]]
            self.next[k, order[:-1][same]] = order[1:][same]
        self.touched = np.unique(self.target).astype(np.int32)

    def _build_native(self, camera, positioned_object, height, width):
        try:
            from ._native import load_library
            lib = load_library()
        except Exception:
            return False
        import ctypes as C
        # the two inverses / vector products of the recipe stay with NumPy (obj/frustums.py:52-60)
        corners = CUBE @ np.linalg.inv(positioned_object.MVP)
        corners /= corners[W_COL]
        probe = np.append(camera.position, 1) @ positioned_object.MVP
        inside = all(-probe[3] < probe[k] < probe[3] for k in range(3))
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        corners, planes, mvp, viewport = f64(corners), f64(camera.frustum_planes), f64(camera.MVP), f64(camera.viewport)
        ns, npt, nt = C.c_int32(), C.c_int32(), C.c_int32()
        rc = lib.mr_host_overlay_build(corners.ctypes.data, planes.ctypes.data, mvp.ctypes.data, viewport.ctypes.data,
                                       float(camera.near), float(camera.far), int(inside), height, width,
                                       C.byref(ns), C.byref(npt), C.byref(nt))
        if rc != 0:
            return False
        n = npt.value
        self.seg_first = np.empty(ns.value, np.int32)
        self.seg_count = np.empty(ns.value, np.int32)
        self.target = np.empty((self.N_TARGETS, n), np.int32)
        self.next = np.empty((self.N_TARGETS, n), np.int32)
        self.z = np.empty(n, np.float64)
        self.touched = np.empty(nt.value, np.int32)
        lib.mr_host_overlay_fetch(self.seg_first.ctypes.data, self.seg_count.ctypes.data, self.target.ctypes.data,
                                  self.next.ctypes.data, self.z.ctypes.data, self.touched.ctypes.data)
        return True

    @property
    def n_points(self):
        return int(self.z.size)

    def replay(self, frame, z_buffer, sign):
        """Apply the statements to NumPy buffers in place (float32 (H, W, 3) frame, float64 (H, W) z)."""
        zf, ff = z_buffer.reshape(-1), frame.reshape(-1, 3)
        half_red = RED / 2
        for first, count in zip(self.seg_first, self.seg_count):
            sl = slice(first, first + count)
            z = self.z[sl]
            keep = (zf[self.target[0, sl]] - z) * sign >= 0
            zk = z[keep]
            centre = self.target[0, sl][keep]
            zf[centre] = zk
            ff[centre] = RED
            for step in (0, 2):                                       # row / col neighbours at -1, then at +1
                rn, cn = self.target[1 + step, sl][keep], self.target[2 + step, sl][keep]
                zf[rn] = zk
                zf[cn] = zk
                ff[rn] = ff[rn] * 0.5 + half_red
                ff[cn] = ff[cn] * 0.5 + half_red


def replay_bids(ops, frame, z_buffer, sign):
    """NumPy restatement of what the DEVICE does with the lists (``csrc/kernels_overlay.h``, three phases per
    segment): the kept (target set k, point p) of a segment bid for their target pixels with ``code = k << 26 | p``
    (``atomicMax``) and leave their k in a bit mask (``atomicOr``); the winning bidder of every pixel writes its own
    z (it is the last kept writer in statement order), turns the pixel red if a kept point has it as its centre and
    half-blends it once for every k = 1..4 that has a kept point on it.  The CPU tests hold this equal to
    ``OverlayOps.replay``, the statement-by-statement restatement of upstream."""
    zf, ff = z_buffer.reshape(-1), frame.reshape(-1, 3)
    win, anyk = np.zeros(zf.size, np.uint32), np.zeros(zf.size, np.uint32)
    for first, count in zip(ops.seg_first, ops.seg_count):
        sl = slice(first, first + count)
        z = ops.z[sl]
        keep = np.nonzero((zf[ops.target[0, sl]] - z) * sign >= 0)[0]
        if len(keep) == 0:
            continue
        ks = np.repeat(np.arange(5, dtype=np.uint32), len(keep))
        qs = np.tile(keep.astype(np.uint32), 5)
        X = ops.target[:, sl][:, keep].reshape(-1)
        code = (ks << np.uint32(26) | qs) + np.uint32(1)
        np.maximum.at(win, X, code)
        np.bitwise_or.at(anyk, X, np.uint32(1) << ks)
        won = win[X] == code
        Xw, qw = X[won], qs[won]
        zf[Xw] = z[qw]
        f = ff[Xw].copy()
        f[(anyk[Xw] & 1) != 0] = RED
        for k in range(1, 5):
            hit = (anyk[Xw] & (1 << k)) != 0
            f[hit] = (f[hit] * np.float32(0.5) + RED / 2).astype(np.float32)
        ff[Xw] = f
        win[Xw] = 0
        anyk[Xw] = 0


def draw_view_frustum(frame, camera, positioned_object, z_buffer, sign):
    """Draw *positioned_object*'s frustum as seen by *camera* into ``frame`` / ``z_buffer`` in place
    (host replay of the statement lists; ``obj/frustums.py:46-103``)."""
    OverlayOps(camera, positioned_object, z_buffer.shape).replay(frame, z_buffer, sign)
