import os
import sys

# the engine honours its experiment / diagnostic switches (AMBI_*) only with this set (csrc/ambi_common.hpp: ambi_env); the tests use several
os.environ.setdefault("AMBI_EXPERIMENTS", "1")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure). Built on demand with g++."""
    from oracle import oracle_py
    oracle_py.build(ref=os.path.isdir("/root/reference/src"))
    return oracle_py


@pytest.fixture(scope="session")
def hostsim_lib():
    """C ABI + engine stage code on the 1-thread host group (tests/hostsim) -- CPU-only checks."""
    import subprocess
    from ambigram_amd import api
    d = os.path.join(ROOT, "tests", "hostsim")
    subprocess.check_call(["make", "-s", "-C", d])
    return api.load(os.path.join(d, "libambigram_hostsim.so"))


@pytest.fixture(scope="session")
def hip_lib():
    """The product: libambigram_hip.so. No fallback -- missing library or missing GPU is a hard failure."""
    from ambigram_amd import api
    lib = api.load()
    assert lib.ambi_backend_name() == b"hip"
    import ctypes
    n = ctypes.c_int(0)
    lib.ambi_device_count(ctypes.byref(n))
    assert n.value >= 1, "no HIP device visible"
    return lib


@pytest.fixture(scope="session")
def workdir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("ambi"))
