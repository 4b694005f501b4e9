"""CPU suite: the C-ABI library loads, exports every symbol include/mi355rast.h declares, and
fails loudly (never silently falls back) when there is no GPU.  No compute is launched."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as entry
    entry.build_native()
    from py_numpy_renderer_amd import _native
    return _native.load_library()


def test_every_declared_symbol_is_exported(lib):
    from py_numpy_renderer_amd import _native
    header = open(os.path.join(ROOT, "include", "mi355rast.h")).read()
    declared = set(re.findall(r"\b(mr_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_native.EXPORTED_SYMBOLS), declared ^ set(_native.EXPORTED_SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.mr_abi_version() == int(re.search(r"#define MR_ABI_VERSION (\d+)", header).group(1))


def test_struct_layouts_match_the_library(lib):
    from py_numpy_renderer_amd import _native
    for which, struct in enumerate((_native.FrameDesc, _native.MaterialDesc, _native.ModelDesc, _native.Stats)):
        assert lib.mr_abi_struct_size(which) == ctypes.sizeof(struct), struct.__name__
    assert lib.mr_abi_struct_size(99) == -1


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_gpu_is_a_loud_error(lib):
    import scenes
    assert lib.mr_device_available() == 0
    assert lib.mr_init(-1) < 0 and b"no HIP device" in lib.mr_last_error()
    scene = scenes.cube_small(scenes.product_api())
    with pytest.raises(RuntimeError, match="no HIP device"):
        scene.render()


def test_invalid_arguments_are_rejected_without_a_gpu(lib):
    handle = lib.mr_scene_create()
    assert handle
    assert lib.mr_scene_add_model(handle, None) < 0
    assert lib.mr_render(handle, None, None, None) < 0
    out = np.zeros(4, np.float64)
    assert lib.mr_read_z(handle, out.ctypes.data) < 0 and b"nothing rendered" in lib.mr_last_error()
    lib.mr_scene_destroy(handle)


def test_integration_md_stub_matches_the_library(lib):
    """The ctypes stub INTEGRATION.md tells a maintainer of the reference to paste (section B) must
    describe the structs of THIS build: extracted from the document, executed against the built
    library (its own assertions on mr_abi_struct_size run) and compared with the shipped binding."""
    from py_numpy_renderer_amd import _native
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = next(b for b in blocks if "class FrameDesc(C.Structure)" in b)
    stub = stub.replace('C.CDLL("libmi355rast.so")', f'C.CDLL({_native.LIB_PATH!r})')
    ns = {}
    exec(compile(stub, "INTEGRATION.md#B", "exec"), ns)
    for ours, theirs in ((_native.FrameDesc, ns["FrameDesc"]), (_native.MaterialDesc, ns["Material"]),
                         (_native.ModelDesc, ns["ModelDesc"])):
        assert ctypes.sizeof(ours) == ctypes.sizeof(theirs)
        assert [(n, ctypes.sizeof(t)) for n, t in ours._fields_] == [(n, ctypes.sizeof(t)) for n, t in theirs._fields_]
    header = open(os.path.join(ROOT, "include", "mi355rast.h")).read()
    body = re.search(r"typedef struct mr_frame_desc \{(.*?)\} mr_frame_desc;", header, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    declared = re.findall(r"\b([a-z_0-9]+)(?:\[\d+\])?\s*[,;]", body)
    assert declared == [n for n, _ in ns["FrameDesc"]._fields_]
