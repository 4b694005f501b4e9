"""Public names of the package (what ``from py_numpy_renderer_amd import *`` gives)."""
from .constants import PROJECTION_TYPE, SUBSYSTEM, SYSTEM
from .core import Camera, Light, Model, Scene, TextureMaps
from .cube_map import CubeMap
from .lightning import Lightning
from .materials import Material
from .triangular import Errors
from .transformation import rotate_xyz, scale, translation

__all__ = ["Errors", "CubeMap", "Camera", "Light", "Model", "Scene", "TextureMaps", "Material", "Lightning",
           "PROJECTION_TYPE", "SUBSYSTEM", "SYSTEM", "scale", "translation", "rotate_xyz"]
