"""The C-ABI shared library loads without a GPU and exports every symbol include/ambigram_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "ambigram_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ambi_[a-z_0-9]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    return os.path.join(ROOT, "ambigram_amd", "libambigram_hip.so")


def test_library_exports_declared_symbols(built):
    lib = ctypes.CDLL(built)
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), "missing export " + n


def test_binding_covers_header():
    from ambigram_amd import api
    api.load()
    assert sorted(api.EXPORTS) == _declared()


def test_backend_is_hip_and_has_no_cpu_fallback(built):
    from ambigram_amd import api
    lib = api.load()
    assert lib.ambi_backend_name() == b"hip"
    assert lib.ambi_abi_version() == 1
    n = ctypes.c_int(-1)
    assert lib.ambi_device_count(ctypes.byref(n)) == 0
    if n.value == 0:   # CPU container: the engine must refuse to run rather than fall back
        g = api.Graph(lib, os.path.join(ROOT, "tests", "data", "readme6.lh"))
        b = api.Batch(lib)
        b.add_chromosome_sol(g, 0, os.path.join(ROOT, "tests", "data", "readme6.sol"))
        with pytest.raises(api.AmbiError) as e:
            b.upload()
        assert e.value.code == -30


def test_missing_library_fails_loudly(tmp_path):
    from ambigram_amd import api
    with pytest.raises(RuntimeError):
        api.load(str(tmp_path / "libambigram_hip.so"))
