"""The C-ABI library loads without a GPU and exports every symbol include/fem_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "fem_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fem_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from fem_amd.device import ABI_SYMBOLS
    assert sorted(ABI_SYMBOLS) == declared_symbols()


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from fem_amd.device import hip_library_path
    lib = ctypes.CDLL(hip_library_path())
    for name in declared_symbols():
        assert hasattr(lib, name), name


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful without a GPU")
def test_open_fails_loudly_without_a_gpu():
    from fem_amd import Device, FemError
    with pytest.raises(FemError):
        Device(0)
