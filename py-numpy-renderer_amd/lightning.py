"""Light kinds (reference: ``obj/lightning.py:4-7``)."""
from enum import Enum


class Lightning(Enum):
    DIRECTIONAL_LIGHTNING = 0
    POINT_LIGHTNING = 1
    SPOT_LIGHTNING = 2
