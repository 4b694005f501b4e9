"""Index aliases and enums of the renderer API.

Mirrors the names exported by the reference's ``obj/constants.py:5-31`` so that user
code written against it (``pts[XY]``, ``SYSTEM.LH`` ...) keeps working.  ``SYSTEM`` is
used arithmetically by the reference (``np.inf * system``, ``obj/core.py:590``), hence
plain ints rather than ``enum`` members.
"""
import numpy as np

_last = lambda i: (Ellipsis, i)          # index the last axis

U = X = _last(0)
V = Y = _last(1)
Z = _last(2)
W = _last(3)
W_COL = _last([3])                       # keeps the axis: pts[:, [3]]
XY = _last((0, 1))
XZ = _last((0, 2))
YZ = _last((1, 2))
XYZ = _last(slice(None, 3))
XYZW = None
mat3x3 = (slice(None, 3), slice(None, 3))
add_dim = _last(np.newaxis)


class PROJECTION_TYPE:
    PERSPECTIVE = 1
    ORTHOGRAPHIC = 2


class SUBSYSTEM:
    DIRECTX = 1
    OPENGL = 2


class SYSTEM:
    LH = -1
    RH = 1
