"""ctypes binding of the CPU oracle (oracle/liboracle.so) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package (ambigram_amd/) never does.  See oracle/bfb_oracle.hpp for what the oracle restates and how it is pinned.
"""
import ctypes
import json
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FLAG_REVERSED = 1
FLAG_ALL = 2
FLAG_JUNC_INFO = 4
FLAG_KEEP_ORDERS = 8


def build(ref=True):
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    targets = ["all"] + (["ref"] if ref else [])
    subprocess.check_call(["make", "-s", "-C", _HERE] + targets)


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("AMBI_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")   # (another build of the oracle: sanitizer runs)
        if not os.path.exists(path):
            build(ref=False)
        L = ctypes.CDLL(path)
        L.oracle_run_bfb.restype = ctypes.c_void_p
        L.oracle_run_bfb.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int,
                                     ctypes.c_longlong, ctypes.POINTER(ctypes.c_double)]
        L.oracle_graph_dump.restype = ctypes.c_void_p
        L.oracle_graph_dump.argtypes = [ctypes.c_char_p]
        L.oracle_free.argtypes = [ctypes.c_void_p]
        _LIB = L
    return _LIB


def _take(ptr):
    s = ctypes.string_at(ptr).decode()
    lib().oracle_free(ptr)
    return json.loads(s)


_LIB_O0 = None


def lib_O0():
    """The oracle compiled at -O0, the optimisation level of the reference's shipped build (CMakeLists.txt:7-8)."""
    global _LIB_O0
    if _LIB_O0 is None:
        path = os.path.join(_HERE, "liboracle_O0.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s", "-C", _HERE, "O0"])
        L = ctypes.CDLL(path)
        L.oracle_run_bfb.restype = ctypes.c_void_p
        L.oracle_run_bfb.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int,
                                     ctypes.c_longlong, ctypes.POINTER(ctypes.c_double)]
        L.oracle_free.argtypes = [ctypes.c_void_p]
        _LIB_O0 = L
    return _LIB_O0


def run_bfb(lh, sols, juncs="", reversed_=False, all_=False, junc_info=False, keep_orders=False, max_orders=0, O0=False):
    """Whole `--op bfb` flow on the CPU oracle. `sols`: list of .sol paths, one per chromosome reaching the ILP."""
    flags = (FLAG_REVERSED if reversed_ else 0) | (FLAG_ALL if all_ else 0) | \
            (FLAG_JUNC_INFO if junc_info else 0) | (FLAG_KEEP_ORDERS if keep_orders else 0)
    sec = ctypes.c_double(0)
    p = (lib_O0() if O0 else lib()).oracle_run_bfb(lh.encode(), juncs.encode(), ",".join(sols).encode(), flags, max_orders,
                                                     ctypes.byref(sec))
    out = _take(p)
    out["seconds"] = sec.value
    return out


def run_sc_bfb(lhs, sols, reversed_=False, all_=False, max_orders=0):
    """Whole `--op sc_bfb` flow (localhap.cpp:390-679): `lhs` = the .lh files, `sols` = one joint .sol per chromosome that
    reaches the ILP."""
    L = lib()
    L.oracle_run_sc_bfb.restype = ctypes.c_void_p
    L.oracle_run_sc_bfb.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_longlong]
    flags = (FLAG_REVERSED if reversed_ else 0) | (FLAG_ALL if all_ else 0)
    return _take(L.oracle_run_sc_bfb(",".join(lhs).encode(), ",".join(sols).encode(), flags, max_orders))


def ilp_sc(lhs, chr_=0):
    """Joint ILP of `--op sc_bfb` (BFB_ILP_SC, LGM.cpp:4754-5093), literal restatement, as CSR + bounds."""
    L = lib()
    L.oracle_ilp_sc_json.restype = ctypes.c_void_p
    L.oracle_ilp_sc_json.argtypes = [ctypes.c_char_p, ctypes.c_int]
    return _take(L.oracle_ilp_sc_json(",".join(lhs).encode(), chr_))


def graph_dump(lh, juncs=None):
    """Parsed graph (after calculateCopyNum); with `juncs` also after readComponents, plus its components."""
    if juncs is None:
        return _take(lib().oracle_graph_dump(lh.encode()))
    L = lib()
    L.oracle_graph_dump_juncs.restype = ctypes.c_void_p
    L.oracle_graph_dump_juncs.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    return _take(L.oracle_graph_dump_juncs(lh.encode(), juncs.encode()))


def ilp(lh, chr_=0, juncs="", junc_info=False, literal=False):
    """ILP model of one chromosome (BFB_ILP, LGM.cpp:4397-4752) as CSR + bounds; `literal` runs the reference's
    O(numPat^2) coefficient loop instead of its closed form."""
    L = lib()
    L.oracle_ilp_json.restype = ctypes.c_void_p
    L.oracle_ilp_json.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                  ctypes.POINTER(ctypes.c_double)]
    sec = ctypes.c_double(0)
    out = _take(L.oracle_ilp_json(lh.encode(), juncs.encode(), chr_, 1 if junc_info else 0, 1 if literal else 0, ctypes.byref(sec)))
    out["seconds"] = sec.value
    return out


def translocation(lh, paths):
    """translocationBFB (LGM.cpp:4052-4193) alone: `paths` = list of per-chromosome paths (signed segment ids)."""
    L = lib()
    L.oracle_translocation_json.restype = ctypes.c_void_p
    L.oracle_translocation_json.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    txt = ";".join(",".join(str(v) for v in p) for p in paths)
    return _take(L.oracle_translocation_json(lh.encode(), txt.encode()))


def ref_graph_dump(lh, juncs=None, write_to=None):
    """Parsed graph from the REAL reference graph model (oracle/_ref, container-only). None if unavailable.
    With `juncs`: after the graph-level effects of readComponents, driven through the reference's graph API."""
    exe = os.path.join(_HERE, "_ref", "ref_graph_dump")
    if not os.path.exists(exe):
        return None
    env = dict(os.environ)
    if write_to:
        env["REF_WRITE_LH"] = write_to      # Graph::writeGraph on the graph as it stands at the end
    out = subprocess.run([exe, lh] + ([juncs] if juncs else []), capture_output=True, text=True, env=env)
    if out.returncode != 0:
        return {"ok": False, "err": "reference exited %d" % out.returncode}
    return json.loads(out.stdout.strip().splitlines()[-1])
