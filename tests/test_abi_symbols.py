"""CPU-side checks of the drop-in boundary: the library loads, exports every symbol the
headers declare, and refuses to run without a gfx950 device (no CPU fallback)."""
import ctypes
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import rio_gp
    rio_gp.build()
    return rio_gp


def _declared():
    names = set()
    for hdr in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = open(hdr).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(rio_(?:gp|op)_[a-z0-9_]+)\s*\(", src))
    return names


def test_library_exports_every_declared_symbol(built):
    L = ctypes.CDLL(built.LIB_PATH)
    names = _declared()
    assert len(names) >= 30
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing


def test_abi_version(built):
    assert built.lib().rio_gp_abi_version() == 1


def test_headers_cite_reference_lines():
    for hdr in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = open(hdr).read()
        assert re.search(r"local\.rs:\d+", src) and re.search(r"mod\.rs:\d+", src), hdr


def test_create_fails_loudly_without_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(built.ObjectPlacementError) as e:
        built.GpuPlacement(1000, 4)
    assert e.value.rc == built.ENODEV and "no CPU fallback" in str(e.value)


def test_bad_cfg_rejected(built):
    h = ctypes.c_void_p()
    cfg = built.Cfg(4, 0, 10, 2, 0, 0, 0)  # wrong struct_size
    assert built.lib().rio_gp_create(ctypes.byref(cfg), ctypes.byref(h)) == built.EINVAL
    assert not h.value
