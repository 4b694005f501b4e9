"""Per-face result codes of the rasteriser (reference: ``obj/triangular.py:15-20``).

The reference's ``rasterize`` returns one of these flags (or 0) per face and ``Scene.render``
prints their histogram for the lit pass (``obj/core.py:625-636``); here the codes come back from
the device as one byte per face (``mr_read_face_status``).
"""
from enum import Flag, auto


class Errors(Flag):
    BACK_FACE_CULLING = auto()
    WRONG_MIN_MAX = auto()
    EMPTY_B = auto()
    EMPTY_Z = auto()
    CLIPPED = auto()
