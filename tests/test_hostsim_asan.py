"""CPU-only: the engine's stage code (the source the HIP kernels instantiate) under AddressSanitizer + UBSan, through
the host simulation.  GPU sanitizers are not available on the pool, so this is where out-of-bounds indexing in the
kernels' algorithms is caught.  Runs in a child process because the sanitizer runtime has to be preloaded."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, tempfile
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import engine_checks as ec
from ambigram_amd import api
from oracle import oracle_py
oracle_py.build(ref=False)
lib = api.load(%(lib)r)
with tempfile.TemporaryDirectory() as d:
    ec.check_fixed_and_synthetic(lib, oracle_py, d, small_only=True)
    ec.check_search_budget(lib, oracle_py, d)
    ec.check_random_decompositions(lib, oracle_py, d, range(30), budget=2)
    ec.check_edge_cases(lib, oracle_py, d)
    ec.check_enumerate_variants(lib, oracle_py, d)
    ec.check_all_mode(lib, oracle_py, d, seeds=range(10))
    ec.check_injected_validity(lib, oracle_py, d)                 # parallel search stages, --all bitmaps with mixed verdicts
    ec.check_all_two_forms(lib, d, seeds=range(300, 330))         # one thread per order vs one wavefront per order
    ec.check_mixed_batch(lib, oracle_py, d)                       # > 32 units: ordinary chain; the runs above: express path
    ec.check_arena_limit(lib, d, n_units=40)                      # units the plan stage has no room for
    ec.check_arena_limit(lib, d, n_units=6, seeds=range(9400, 9406))
print("SANITIZED RUN CLEAN")
"""


def test_stage_code_under_asan_ubsan():
    def runtime(name):
        p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
        return p if os.path.isabs(p) and os.path.exists(p) else None
    asan, ubsan = runtime("libasan.so"), runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("sanitizer runtimes not installed")
    hs = os.path.join(ROOT, "tests", "hostsim")
    r = subprocess.run(["make", "-s", "-C", hs, "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, LD_PRELOAD=asan + " " + ubsan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    code = CHILD % dict(root=ROOT, tests=os.path.join(ROOT, "tests"), lib=os.path.join(hs, "libambigram_hostsim_asan.so"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=os.path.join(ROOT, "tests"), timeout=900)
    assert r.returncode == 0 and "SANITIZED RUN CLEAN" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
