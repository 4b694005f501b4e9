"""Wavefront .mtl material record.

Behaviour follows the reference's ``obj/materials.py:47-77``: class-level defaults, values
assigned from a parsed ``.mtl`` line (a list of strings) collapse to a float when the list
has one entry and to a float32 array otherwise, and an unknown attribute raises
``AttributeError`` so ``hasattr(material, 'map_Kd')`` is the "has a texture" test used by
the shading path (``obj/core.py:146,163,176``).
"""
import numpy as np

_DEFAULTS = dict(
    Pm=0.5,                                   # metalness
    Pr=0.5,                                   # roughness
    Ka=np.array((0.3, 0.0, 0.0)),             # ambient colour
    Kd=np.array((0.8, 0.8, 0.8)),             # diffuse colour
    Ks=np.array((1.0, 1.0, 1.0)),             # specular colour
    d=1.0,                                    # opacity
    Tr=0,                                     # transparency
    Ns=64,                                    # specular exponent
    illum=1,
)


class Material:
    """Attribute bag; texture maps are float32 (H, W, 3) arrays under ``map_*`` / ``norm``.

    Every assignment bumps ``Material.revision`` (class-wide): the device copy of a scene is refreshed
    when it changed since the last frame, so ``material.Kd = [...]`` between two renders is picked up."""

    revision = 0

    def __setattr__(self, key, value):
        Material.revision += 1
        if len(value) == 1:
            item = value[0]
            try:
                item = float(item)
            except ValueError:
                pass
            object.__setattr__(self, key, item)
            return
        tangent = None
        meta = getattr(getattr(value, "dtype", None), "metadata", None)
        if meta and "tangent" in meta:
            tangent = bool(meta["tangent"])
        arr = np.array(value, dtype=np.float32)
        if tangent is not None and not (arr.dtype.metadata or {}).get("tangent") == tangent:
            arr = arr.astype(np.dtype(np.float32, metadata={"tangent": tangent}))
        object.__setattr__(self, key, arr)

    def __getattr__(self, item):
        # only reached when normal lookup failed
        if item in _DEFAULTS:
            return _DEFAULTS[item]
        raise AttributeError("No such attribute", item)

    def is_tangent_space(self, key="norm"):
        """True when the normal map stored under *key* is a tangent-space map."""
        tex = self.__dict__.get(key)
        meta = getattr(getattr(tex, "dtype", None), "metadata", None) or {}
        return bool(meta.get("tangent", False))
