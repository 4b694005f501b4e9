"""Importable alias for the ``py-numpy-renderer_amd/`` source directory.

A hyphen cannot appear in a Python module name, so the code lives in
``py-numpy-renderer_amd/`` (the layout the project asks for) and this package only
points its ``__path__`` there.  ``import py_numpy_renderer_amd.core`` therefore loads
``py-numpy-renderer_amd/core.py``.
"""
import os as _os

_src = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                     "py-numpy-renderer_amd")
if not _os.path.isdir(_src):  # pragma: no cover - broken checkout
    raise ImportError(f"source directory missing: {_src}")
__path__.insert(0, _src)

from ._exports import *  # noqa: E402,F401,F403
from ._exports import __all__  # noqa: E402,F401
