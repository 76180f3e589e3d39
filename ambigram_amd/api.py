"""ctypes binding of libambigram_hip.so (include/ambigram_hip.h) and the host-side mirror of the reference's
`--op bfb` driver (localhap.cpp:49-388).

The HIP library is the product.  There is NO CPU fallback: `load()` raises if the library is missing and the engine
returns AMBI_ERR_NO_DEVICE without a GPU.  (Tests that run without a GPU pass the path of the test-only host
simulation built under tests/hostsim explicitly; nothing here picks it up by itself.)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libambigram_hip.so")

FLAG_REVERSED = 1
FLAG_ALL = 2
FLAG_LAZY_ORDERS = 4

ST_OK, ST_SHORTCUT, ST_INFEASIBLE, ST_NO_VALID_ORDER = 0, 1, 2, 3


class UnitResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("bias", C.c_int32), ("n_nodes", C.c_int32), ("bkp_len", C.c_int32),
                ("path_len", C.c_int32), ("path_indel_len", C.c_int32), ("indel_printed", C.c_int32),
                ("n_out_junc", C.c_int32), ("first_forward", C.c_int32), ("evaluated", C.c_int32),
                ("num_orders", C.c_int64), ("first_valid", C.c_int64), ("inv_cn_sum", C.c_double),
                ("path_indel_stored", C.c_int32), ("reserved", C.c_int32)]


class RunsView(C.Structure):     # ambi_runs_view_t
    _fields_ = [("n_runs", C.c_int64), ("n_cells", C.c_int64), ("bytes", C.c_int64), ("copied_bytes", C.c_int64),
                ("lengths", C.POINTER(C.c_int32)), ("run_counts", C.POINTER(C.c_int32)),
                ("run_start", C.POINTER(C.c_int32)), ("run_len", C.POINTER(C.c_int32)), ("run_off", C.POINTER(C.c_int64))]


class AmbiError(RuntimeError):
    def __init__(self, lib, code, what=""):
        self.code = code
        msg = lib.ambi_error_string(code).decode() if lib is not None else str(code)
        super().__init__("%s: %s (%d)" % (what, msg, code))


_P = C.POINTER


def _declare(L):
    vp = C.c_void_p
    i32, i64, u32 = C.c_int32, C.c_int64, C.c_uint32
    pi32, pi64, pi8, pu8, pd = _P(i32), _P(i64), _P(C.c_int8), _P(C.c_uint8), _P(C.c_double)
    sig = {
        "ambi_error_string": (C.c_char_p, [C.c_int]),
        "ambi_abi_version": (C.c_int, []),
        "ambi_backend_name": (C.c_char_p, []),
        "ambi_device_count": (C.c_int, [_P(C.c_int)]),
        "ambi_set_device": (C.c_int, [C.c_int]),
        "ambi_graph_read_lh": (C.c_int, [C.c_char_p, _P(vp)]),
        "ambi_graph_destroy": (None, [vp]),
        "ambi_graph_sizes": (C.c_int, [vp, pi32, pi32, pi32]),
        "ambi_graph_segments": (C.c_int, [vp, pi32, pi32, pi32, pi32, pd, pd]),
        "ambi_graph_junctions": (C.c_int, [vp, pi32, pi8, pi32, pi8, pd, pd, pu8, pu8]),
        "ambi_graph_chromosome": (C.c_int, [vp, i32, pi32, pi32]),
        "ambi_graph_chrom_name": (i64, [vp, i32, C.c_char_p, i64]),
        "ambi_graph_read_juncs": (C.c_int, [vp, C.c_char_p]),
        "ambi_graph_log": (i64, [vp, C.c_char_p, i64]),
        "ambi_graph_props": (C.c_int, [vp, pi32, pi32, C.c_char_p, i64]),
        "ambi_graph_components": (C.c_int, [vp, pi32, i32, pi32, i32]),
        "ambi_batch_create": (C.c_int, [_P(vp)]),
        "ambi_batch_destroy": (None, [vp]),
        "ambi_batch_add_chromosome": (C.c_int, [vp, vp, i32, i32, pi32, pi32, i32]),
        "ambi_batch_add_chromosome_sol": (C.c_int, [vp, vp, i32, C.c_char_p]),
        "ambi_batch_add_chromosome_sol_block": (C.c_int, [vp, vp, i32, C.c_char_p, i32, i32]),
        "ambi_graph_recalculate": (C.c_int, [vp]),
        "ambi_graph_write_lh": (C.c_int, [vp, C.c_char_p]),
        "ambi_graph_trx_before": (C.c_int, [vp, pi32, i32]),
        "ambi_graph_trx_original": (C.c_int, [vp, _P(vp)]),
        "ambi_graph_trx_restore": (C.c_int, [vp, pi32, i32, i32, C.c_char_p, i64, pi64]),
        "ambi_ilp_build_sc": (C.c_int, [vp, i32, i32, pd, pd, _P(vp)]),
        "ambi_batch_add_unit": (C.c_int, [vp, i32, i32, pd, i32, pi32, pi32, pi8, pi8, pd, i32, pi32, pi32, pi32, pi32, i32, i32]),
        "ambi_batch_size": (C.c_int, [vp, pi32]),
        "ambi_batch_configure": (C.c_int, [vp, i64, i32, i32, i32]),
        "ambi_batch_debug_inject_validity": (C.c_int, [vp, i32, pi8, i64]),
        "ambi_batch_upload": (C.c_int, [vp]),
        "ambi_batch_run": (C.c_int, [vp, u32, vp]),
        "ambi_batch_wait": (C.c_int, [vp]),
        "ambi_batch_wait_results": (C.c_int, [vp]),
        "ambi_batch_download": (C.c_int, [vp]),
        "ambi_batch_fetch_paths": (C.c_int, [vp]),
        "ambi_batch_run_sharded": (C.c_int, [vp, u32, pi32, i32]),
        "ambi_batch_device_results": (C.c_int, [vp, _P(vp), pi64]),
        "ambi_batch_pack_paths": (C.c_int, [vp, i32, vp, vp, i64, vp, vp]),
        "ambi_batch_pack_runs": (C.c_int, [vp, i32, vp, vp, vp, vp, i64, vp, vp]),
        "ambi_expand_runs": (C.c_int, [vp, vp, vp, i64, vp, i64, vp]),
        "ambi_debug_stream_probe": (C.c_int, [vp, vp, _P(C.c_float)]),
        "ambi_batch_runs_to_host": (C.c_int, [vp, i32, i32, vp]),
        "ambi_batch_runs_wait": (C.c_int, [vp, i32, _P(RunsView)]),
        "ambi_batch_runs_unit_path": (C.c_int, [vp, i32, i32, pi32, i32]),
        "ambi_batch_unit_result": (C.c_int, [vp, i32, _P(UnitResult)]),
        "ambi_batch_unit_path": (C.c_int, [vp, i32, i32, pi32, i32]),
        "ambi_batch_unit_bkp": (C.c_int, [vp, i32, pi32, i32]),
        "ambi_batch_unit_prepare": (C.c_int, [vp, i32, pd, pd, pi32, pi32]),
        "ambi_batch_unit_dag": (C.c_int, [vp, i32, pi32, pi32, _P(C.c_uint64)]),
        "ambi_batch_unit_dag_words": (C.c_int, [vp, i32, _P(C.c_uint64)]),
        "ambi_batch_unit_dag_nwords": (C.c_int, [vp, i32, i32, _P(C.c_uint64)]),
        "ambi_batch_unit_out_juncs": (C.c_int, [vp, i32, pi32, pi32, pi32, i32]),
        "ambi_batch_unit_orders": (C.c_int, [vp, i32, i64, i64, pu8]),
        "ambi_batch_set_timing": (C.c_int, [vp, i32]),
        "ambi_batch_set_timing_mask": (C.c_int, [vp, C.c_uint32]),
        "ambi_batch_kernel_count": (C.c_int, [vp]),
        "ambi_batch_kernel_time": (C.c_int, [vp, i32, _P(C.c_char_p), _P(C.c_float)]),
        "ambi_batch_kernel_span": (C.c_int, [vp, i32, _P(C.c_float), _P(C.c_float)]),
        "ambi_batch_slices": (C.c_int, [vp]),
        "ambi_batch_all_count": (C.c_int, [vp, i32, i32, _P(C.c_int64)]),
        "ambi_batch_all_orders": (C.c_int, [vp, i32, i32, i64, i64, _P(C.c_int64)]),
        "ambi_batch_all_paths": (C.c_int, [vp, i32, i32, i64, i64, _P(C.c_int32), _P(C.c_int32), i64]),
        "ambi_batch_all_set_shard": (C.c_int, [vp, i32, i32]),
        "ambi_batch_all_device": (C.c_int, [vp, _P(vp), pi64]),
        "ambi_batch_all_finish": (C.c_int, [vp]),
        "ambi_batch_traffic": (C.c_int, [vp, pi64, pi64, pi64]),
        "ambi_format_path": (i64, [vp, pi32, i32, C.c_char_p, i64]),
        "ambi_translocation_bfb": (C.c_int, [vp, pi32, pi64, i32, pi32, i32]),
        "ambi_ilp_build": (C.c_int, [vp, i32, pd, pd, i32, C.c_double, i32, _P(vp)]),
        "ambi_ilp_build_device": (C.c_int, [vp, i32, pd, pd, i32, C.c_double, i32, _P(C.c_float), _P(vp)]),
        "ambi_ilp_destroy": (None, [vp]),
        "ambi_ilp_sizes": (C.c_int, [vp, pi64, pi64, pi32, pi32]),
        "ambi_ilp_copy": (C.c_int, [vp, pi64, pi32, pd, pd, pd, pd, pd, pd]),
        "ambi_ilp_write_lp": (C.c_int, [vp, C.c_char_p]),
        "ambi_ilp_write_mps": (C.c_int, [vp, C.c_char_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    return sorted(sig)


_LIBS = {}
EXPORTS = []
_RUNTIME = []


def _preload_hip_runtime():
    """One process must hold ONE HIP runtime.  PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME
    libamdhip64.so.7); if the engine pulled in the system copy first, a later `import torch` would load the bundled
    copy as a second runtime and lose the GPU.  So when a torch wheel is installed, its runtime is loaded first and
    the engine binds to it by SONAME; without torch the system ROCm runtime is used."""
    if _RUNTIME:
        return
    _RUNTIME.append(None)
    # four engine streams + the caller's (+ RCCL's) want more than the runtime's default of 4 hardware queues per process, or
    # two of them share a queue and serialize (bench.py, DESIGN.md "Launch order"); no effect if the runtime is already up
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec and spec.origin:
            cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
            if os.path.exists(cand):
                _RUNTIME[0] = C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        _RUNTIME[0] = None


def load(path=None):
    """Load the engine library.  Fails loudly when the HIP extension has not been built."""
    path = os.path.abspath(path or DEFAULT_LIB)
    if path not in _LIBS:
        if not os.path.exists(path):
            raise RuntimeError("ambigram_amd: %s not found -- build the HIP extension first "
                               "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback" % path)
        if os.path.basename(path) == os.path.basename(DEFAULT_LIB):
            _preload_hip_runtime()
        L = C.CDLL(path)
        names = _declare(L)
        if not EXPORTS:
            EXPORTS.extend(names)
        _LIBS[path] = L
    return _LIBS[path]


def _arr(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a, a.ctypes.data_as(_P(np.ctypeslib.as_ctypes_type(dtype)))


class Graph:
    """Parsed .lh (replaces Graph + calculateHapDepth + calculateCopyNum + readBFBProps, localhap.cpp:65-75)."""

    def __init__(self, lib, lh_path):
        self.lib = lib
        self.h = C.c_void_p()
        rc = lib.ambi_graph_read_lh(lh_path.encode(), C.byref(self.h))
        if rc != 0:
            raise AmbiError(lib, rc, "read_lh(%s)" % lh_path)
        n, m, c = C.c_int32(), C.c_int32(), C.c_int32()
        lib.ambi_graph_sizes(self.h, C.byref(n), C.byref(m), C.byref(c))
        self.n_seg, self.n_junc, self.n_chr = n.value, m.value, c.value

    def close(self):
        if self.h:
            self.lib.ambi_graph_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _refresh(self):
        n, m, c = C.c_int32(), C.c_int32(), C.c_int32()
        self.lib.ambi_graph_sizes(self.h, C.byref(n), C.byref(m), C.byref(c))
        self.n_seg, self.n_junc, self.n_chr = n.value, m.value, c.value

    def read_juncs(self, path):
        rc = self.lib.ambi_graph_read_juncs(self.h, path.encode())
        if rc != 0:
            raise AmbiError(self.lib, rc, "read_juncs")
        self._refresh()

    def write_lh(self, path):
        """Graph::writeGraph (Graph.cpp:239-266): the graph as it stands, as .lh text"""
        rc = self.lib.ambi_graph_write_lh(self.h, path.encode())
        if rc != 0:
            raise AmbiError(self.lib, rc, "write_lh")

    def trx_before(self):
        """PROP I1 / C1 (TRX-BFB): the map rebuilt segment id -> id in the file ([0] unused), or None for an ordinary graph."""
        n = self.lib.ambi_graph_trx_before(self.h, None, 0)
        if n <= 0:
            return None
        m = np.zeros(n, np.int32)
        self.lib.ambi_graph_trx_before(self.h, m.ctypes.data_as(_P(C.c_int32)), n)
        return m

    def trx_restore(self, path):
        """virusBFB (LGM.cpp:3839-3939): a path over the rebuilt graph -> (path over the segments of the file, lines the reference prints)."""
        buf = np.zeros(len(path) + 8, np.int32)
        buf[:len(path)] = path
        text = C.create_string_buffer(64 + 32 * (len(path) + 8) * 2)
        tl = C.c_int64()
        n = self.lib.ambi_graph_trx_restore(self.h, buf.ctypes.data_as(_P(C.c_int32)), len(path), len(buf), text, len(text), C.byref(tl))
        if n < 0:
            raise AmbiError(self.lib, n, "trx_restore")
        return buf[:n].copy(), text.value.decode().splitlines()

    def segments(self):
        n = self.n_seg
        sid, chr_, st, en = (np.zeros(n, np.int32) for _ in range(4))
        cov, cn = np.zeros(n), np.zeros(n)
        p = lambda a, t: a.ctypes.data_as(_P(t))
        self.lib.ambi_graph_segments(self.h, p(sid, C.c_int32), p(chr_, C.c_int32), p(st, C.c_int32), p(en, C.c_int32),
                                     p(cov, C.c_double), p(cn, C.c_double))
        return dict(id=sid, chr=chr_, start=st, end=en, cov=cov, cn=cn)

    def junctions(self):
        m = self.n_junc
        src, tgt = np.zeros(m, np.int32), np.zeros(m, np.int32)
        sd, td = np.zeros(m, np.int8), np.zeros(m, np.int8)
        cov, cn = np.zeros(m), np.zeros(m)
        inf, bnd = np.zeros(m, np.uint8), np.zeros(m, np.uint8)
        p = lambda a, t: a.ctypes.data_as(_P(t))
        self.lib.ambi_graph_junctions(self.h, p(src, C.c_int32), p(sd, C.c_int8), p(tgt, C.c_int32), p(td, C.c_int8),
                                      p(cov, C.c_double), p(cn, C.c_double), p(inf, C.c_uint8), p(bnd, C.c_uint8))
        return dict(src=src, sdir=sd, tgt=tgt, tdir=td, cov=cov, cn=cn, inferred=inf, bounded=bnd)

    def chromosome(self, c):
        s, e = C.c_int32(), C.c_int32()
        rc = self.lib.ambi_graph_chromosome(self.h, c, C.byref(s), C.byref(e))
        if rc != 0:
            raise AmbiError(self.lib, rc, "chromosome")
        return s.value, e.value

    def log(self):
        n = self.lib.ambi_graph_log(self.h, None, 0)
        buf = C.create_string_buffer(n + 1)
        self.lib.ambi_graph_log(self.h, buf, n + 1)
        return [l for l in buf.value.decode().split("\n") if l != ""]

    def props(self):
        ins, con = C.c_int32(), C.c_int32()
        buf = C.create_string_buffer(256)
        self.lib.ambi_graph_props(self.h, C.byref(ins), C.byref(con), buf, 256)
        return ins.value, con.value, buf.value.decode()

    def components(self):
        """Components collected by read_juncs (readComponents, LGM.cpp:5096-5156)."""
        # sizes first (null buffers: the call returns the component count, the last offset is the id count), then exact buffers
        n = self.lib.ambi_graph_components(self.h, None, 0, None, 0)
        if n <= 0:
            return []
        offs = np.zeros(n + 1, np.int32)
        self.lib.ambi_graph_components(self.h, None, 0, offs.ctypes.data_as(_P(C.c_int32)), n + 1)
        ids = np.zeros(max(int(offs[n]), 1), np.int32)
        self.lib.ambi_graph_components(self.h, ids.ctypes.data_as(_P(C.c_int32)), len(ids), offs.ctypes.data_as(_P(C.c_int32)), n + 1)
        return [ids[offs[c]:offs[c + 1]].tolist() for c in range(n)]

    def chrom_name(self, seg_id):
        buf = C.create_string_buffer(256)
        self.lib.ambi_graph_chrom_name(self.h, seg_id, buf, 256)
        return buf.value.decode()

    def dump(self):
        """The parsed graph in the shape of the reference-made fixtures tests/golden/graph_*.json."""
        s, j = self.segments(), self.junctions()
        sign = {1: "+", -1: "-"}
        num = lambda x: int(x) if float(x).is_integer() else float(x)
        return {
            "ok": True, "err": "",
            "segs": [[int(s["id"][i]), int(s["chr"][i]), self.chrom_name(int(s["id"][i])), int(s["start"][i]), int(s["end"][i]),
                      num(s["cov"][i]), num(s["cn"][i])] for i in range(self.n_seg)],
            "juncs": [[int(j["src"][i]), sign[int(j["sdir"][i])], int(j["tgt"][i]), sign[int(j["tdir"][i])], num(j["cov"][i]),
                       num(j["cn"][i]), int(j["inferred"][i]), int(j["bounded"][i])] for i in range(self.n_junc)],
            "sources": [self.chromosome(c)[0] for c in range(self.n_chr)],
            "sinks": [self.chromosome(c)[1] for c in range(self.n_chr)],
            "log": self.log(),
        }

    def format_path(self, path):
        a, p = _arr(path, np.int32)
        n = self.lib.ambi_format_path(self.h, p, len(a), None, 0)
        buf = C.create_string_buffer(n + 1)
        self.lib.ambi_format_path(self.h, p, len(a), buf, n + 1)
        return buf.value.decode()

    def translocation_bfb(self, paths):
        """BFB-TRX stitching (LGM.cpp:4052-4193). `paths`: list of per-chromosome int arrays (may be modified)."""
        offs = np.zeros(len(paths) + 1, np.int64)
        for i, q in enumerate(paths):
            offs[i + 1] = offs[i] + len(q)
        flat = np.ascontiguousarray(np.concatenate([np.asarray(q, np.int32) for q in paths]) if offs[-1] else np.zeros(0, np.int32))
        cap = int(offs[-1]) * 2 + 16
        out = np.zeros(cap, np.int32)
        n = self.lib.ambi_translocation_bfb(self.h, flat.ctypes.data_as(_P(C.c_int32)), offs.ctypes.data_as(_P(C.c_int64)),
                                            len(paths), out.ctypes.data_as(_P(C.c_int32)), cap)
        if n < 0:
            raise AmbiError(self.lib, n, "translocation_bfb")
        new_paths = [flat[offs[i]:offs[i + 1]].copy() for i in range(len(paths))]
        return out[:n].copy(), new_paths


class Batch:
    """A batch of units (chromosomes of samples) resident in HBM."""

    def __init__(self, lib):
        self.lib = lib
        self.h = C.c_void_p()
        rc = lib.ambi_batch_create(C.byref(self.h))
        if rc != 0:
            raise AmbiError(lib, rc, "batch_create")

    def close(self):
        if self.h:
            self.lib.ambi_batch_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc < 0:
            raise AmbiError(self.lib, rc, what)
        return rc

    def add_chromosome(self, graph, chr_, cols, vals, infeasible=False):
        c, pc = _arr(cols, np.int32)
        v, pv = _arr(vals, np.int32)
        return self._ck(self.lib.ambi_batch_add_chromosome(self.h, graph.h, chr_, len(c), pc, pv, 1 if infeasible else 0), "add_chromosome")

    def add_chromosome_sol(self, graph, chr_, sol_path):
        return self._ck(self.lib.ambi_batch_add_chromosome_sol(self.h, graph.h, chr_, sol_path.encode()), "add_chromosome_sol")

    def add_unit(self, n_seg, seg_base, seg_cn, j_src, j_tgt, j_sdir, j_tdir, j_cn, e_is_loop, e_a, e_b, e_cn,
                 infeasible=False, has_components=False):
        cn, pcn = _arr(seg_cn, np.float64)
        js, pjs = _arr(j_src, np.int32); jt, pjt = _arr(j_tgt, np.int32)
        jsd, pjsd = _arr(j_sdir, np.int8); jtd, pjtd = _arr(j_tdir, np.int8)
        jc, pjc = _arr(j_cn, np.float64)
        el, pel = _arr(e_is_loop, np.int32); ea, pea = _arr(e_a, np.int32); eb, peb = _arr(e_b, np.int32); ec, pec = _arr(e_cn, np.int32)
        return self._ck(self.lib.ambi_batch_add_unit(self.h, n_seg, seg_base, pcn, len(js), pjs, pjt, pjsd, pjtd, pjc, len(ea),
                                                      pel, pea, peb, pec, 1 if infeasible else 0, 1 if has_components else 0), "add_unit")

    def size(self):
        n = C.c_int32()
        self.lib.ambi_batch_size(self.h, C.byref(n))
        return n.value

    def configure(self, order_arena_bytes=-1, ideal_cap=0, first_budget=0, target_lanes=0):
        self._ck(self.lib.ambi_batch_configure(self.h, order_arena_bytes, ideal_cap, first_budget, target_lanes), "configure")

    def debug_inject_validity(self, unit, verdicts):
        """Diagnostics hook (include/ambigram_hip.h): verdict overrides for the 2*R order evaluations of one unit
        (forward-seed orientation first); 1 valid, 0 invalid, negative status, 127 = evaluate as usual."""
        v, vp = _arr(verdicts, np.int8)
        self._ck(self.lib.ambi_batch_debug_inject_validity(self.h, unit, vp, len(v)), "debug_inject_validity")

    def upload(self):
        self._ck(self.lib.ambi_batch_upload(self.h), "upload")

    def run(self, flags=0, stream=None):
        self._ck(self.lib.ambi_batch_run(self.h, flags, C.c_void_p(stream or 0)), "run")

    def wait(self):
        self._ck(self.lib.ambi_batch_wait(self.h), "wait")

    def wait_results(self):
        """Reconstruction results complete (small batches: before the order tables behind them are written)."""
        self._ck(self.lib.ambi_batch_wait_results(self.h), "wait_results")

    def download(self):
        self._ck(self.lib.ambi_batch_download(self.h), "download")

    def run_sharded(self, flags=0, devices=None, n_devices=0):
        """Upload + run + download with the units dealt round-robin over several devices, one host thread per device
        (ambi_batch_run_sharded).  devices: list of device ordinals (an ordinal may repeat) or None for the first n_devices
        visible devices (all if n_devices <= 0)."""
        if devices is None:
            self._ck(self.lib.ambi_batch_run_sharded(self.h, flags, None, n_devices), "run_sharded")
        else:
            d, pd = _arr(devices, np.int32)
            self._ck(self.lib.ambi_batch_run_sharded(self.h, flags, pd, len(d)), "run_sharded")

    def fetch_paths(self):
        """Headers, final paths and output junctions on the host (small batches: read from the pinned mailbox the
        reconstruction kernel wrote, no copy command); unit_result / unit_path / unit_out_juncs are valid afterwards."""
        self._ck(self.lib.ambi_batch_fetch_paths(self.h), "fetch_paths")

    def device_results(self):
        p, n = C.c_void_p(), C.c_int64()
        self._ck(self.lib.ambi_batch_device_results(self.h, C.byref(p), C.byref(n)), "device_results")
        return p.value, n.value

    def pack_runs(self, which, dev_lengths_ptr, dev_run_counts_ptr, dev_run_start_ptr, dev_run_len_ptr, run_cap, dev_totals_ptr, stream=None):
        """Final paths in run-length form into device buffers (ambi_batch_pack_runs): the payload of the exchange."""
        self._ck(self.lib.ambi_batch_pack_runs(self.h, which, C.c_void_p(dev_lengths_ptr), C.c_void_p(dev_run_counts_ptr),
                                                C.c_void_p(dev_run_start_ptr), C.c_void_p(dev_run_len_ptr), run_cap,
                                                C.c_void_p(dev_totals_ptr), C.c_void_p(stream or 0)), "pack_runs")

    def runs_to_host(self, which=1, slot=0, stream=None):
        """Queues the packing of the final paths (run-length form) and their ONE copy into pinned host memory behind `stream`
        (ambi_batch_runs_to_host); slots 0 / 1 alternate so that one step's copy travels while the next step computes."""
        self._ck(self.lib.ambi_batch_runs_to_host(self.h, which, slot, C.c_void_p(stream or 0)), "runs_to_host")

    def runs_wait(self, slot=0):
        """Waits for the slot's copy; returns {n_runs, n_cells, bytes, copied_bytes, lengths, run_counts, run_start, run_len} with
        numpy views of the engine's pinned block (valid until the slot is queued again)."""
        v = RunsView()
        self._ck(self.lib.ambi_batch_runs_wait(self.h, slot, C.byref(v)), "runs_wait")
        U = self.size()
        view = lambda p, n: np.ctypeslib.as_array(p, shape=(max(int(n), 0),)) if n > 0 else np.zeros(0, np.int32)
        return {"n_runs": v.n_runs, "n_cells": v.n_cells, "bytes": v.bytes, "copied_bytes": v.copied_bytes,
                "lengths": view(v.lengths, U), "run_counts": view(v.run_counts, U),
                "run_off": np.ctypeslib.as_array(v.run_off, shape=(U + 1,)) if U > 0 else np.zeros(1, np.int64),
                "_run_start": v.run_start, "_run_len": v.run_len}

    def runs_unit_path(self, slot, u):
        """One unit's path expanded on the host from the runs that arrived in `slot` (== unit_path after a download)."""
        n = self._ck(self.lib.ambi_batch_runs_unit_path(self.h, slot, u, None, 0), "runs_unit_path")
        buf = np.empty(max(n, 1), np.int32)
        self.lib.ambi_batch_runs_unit_path(self.h, slot, u, buf.ctypes.data_as(_P(C.c_int32)), n)
        return buf[:n]

    def pack_paths(self, which, dev_lengths_ptr, dev_cells_ptr, cell_cap, dev_total_ptr, stream=None):
        self._ck(self.lib.ambi_batch_pack_paths(self.h, which, C.c_void_p(dev_lengths_ptr), C.c_void_p(dev_cells_ptr), cell_cap,
                                                 C.c_void_p(dev_total_ptr), C.c_void_p(stream or 0)), "pack_paths")

    def unit_result(self, u):
        r = UnitResult()
        self._ck(self.lib.ambi_batch_unit_result(self.h, u, C.byref(r)), "unit_result")
        return {k: getattr(r, k) for k, _ in UnitResult._fields_}

    def unit_path(self, u, which=0):
        # one call into a scratch buffer (the function returns the length and copies at most `cap` cells); a second one only
        # when the path is longer than the scratch
        buf = getattr(self, "_path_buf", None)
        if buf is None:
            buf = self._path_buf = np.empty(1 << 15, np.int32)
            self._path_ptr = buf.ctypes.data_as(_P(C.c_int32))
        n = self._ck(self.lib.ambi_batch_unit_path(self.h, u, which, self._path_ptr, len(buf)), "unit_path")
        if n > len(buf):
            buf = self._path_buf = np.empty(n + (n >> 2), np.int32)
            self._path_ptr = buf.ctypes.data_as(_P(C.c_int32))
            self.lib.ambi_batch_unit_path(self.h, u, which, self._path_ptr, len(buf))
        return buf[:n].copy()

    def unit_bkp(self, u):
        n = self._ck(self.lib.ambi_batch_unit_bkp(self.h, u, None, 0), "unit_bkp")
        out = np.zeros(max(n, 1), np.int32)
        self.lib.ambi_batch_unit_bkp(self.h, u, out.ctypes.data_as(_P(C.c_int32)), n)
        return out[:n]

    def unit_prepare(self, u, n_seg):
        jc = np.zeros(2 * (n_seg + 1)); sc = np.zeros(n_seg + 1)
        tc = np.zeros(n_seg + 1, np.int32); ij = np.zeros(n_seg + 1, np.int32)
        self._ck(self.lib.ambi_batch_unit_prepare(self.h, u, jc.ctypes.data_as(_P(C.c_double)), sc.ctypes.data_as(_P(C.c_double)),
                                                   tc.ctypes.data_as(_P(C.c_int32)), ij.ctypes.data_as(_P(C.c_int32))), "unit_prepare")
        return dict(junc_cn=jc.reshape(-1, 2), seg_cn=sc, target_cn=tc, inv_junc=ij)

    def unit_dag(self, u, K):
        pat = np.zeros((max(K, 1), 3), np.int32); loop = np.zeros((max(K, 1), 3), np.int32); succ = np.zeros(max(K, 1), np.uint64)
        self._ck(self.lib.ambi_batch_unit_dag(self.h, u, pat.ctypes.data_as(_P(C.c_int32)), loop.ctypes.data_as(_P(C.c_int32)),
                                               succ.ctypes.data_as(_P(C.c_uint64))), "unit_dag")
        if K > 63:      # a wide unit: successor sets as Python integers over all four words
            w = np.zeros(4 * K, np.uint64)
            self._ck(self.lib.ambi_batch_unit_dag_nwords(self.h, u, 4, w.ctypes.data_as(_P(C.c_uint64))), "unit_dag_nwords")
            return pat[:K], loop[:K], [sum(int(w[4 * i + k]) << (64 * k) for k in range(4)) for i in range(K)]
        return pat[:K], loop[:K], succ[:K]

    def unit_out_juncs(self, u):
        n = self._ck(self.lib.ambi_batch_unit_out_juncs(self.h, u, None, None, None, 0), "unit_out_juncs")
        a, b, c = (np.zeros(max(n, 1), np.int32) for _ in range(3))
        p = lambda x: x.ctypes.data_as(_P(C.c_int32))
        self.lib.ambi_batch_unit_out_juncs(self.h, u, p(a), p(b), p(c), n)
        return [(int(a[i]), int(b[i]), int(c[i])) for i in range(n)]

    def unit_orders(self, u, first, count, K):
        out = np.zeros((max(count, 1), max(K, 1)), np.uint8)
        self._ck(self.lib.ambi_batch_unit_orders(self.h, u, first, count, out.ctypes.data_as(_P(C.c_uint8))), "unit_orders")
        return out[:count, :K]

    def set_timing(self, on=True):
        self.lib.ambi_batch_set_timing(self.h, 1 if on else 0)

    KERNEL_INDEX = {"ambi_prepare_kernel": 0, "ambi_plan_kernel": 1, "ambi_blocks_build_kernel": 2, "ambi_enumerate_kernel": 3,
                    "ambi_first_kernel": 4, "ambi_finish_kernel": 5, "ambi_finish_ext_kernel": 6}

    def set_timing_only(self, kernel_names):
        """HIP events around the named kernels only (every event pair is a marker in the stream)."""
        mask = 0
        for k in kernel_names:
            mask |= 1 << self.KERNEL_INDEX[k]
        self.lib.ambi_batch_set_timing_mask(self.h, mask)

    def all_orders(self, u, pass_):
        """--all: indices of the valid orders of pass 0 (first orientation) / 1 (flipped), in print order."""
        n = C.c_int64()
        self._ck(self.lib.ambi_batch_all_count(self.h, u, pass_, C.byref(n)), "all_count")
        out = np.zeros(max(n.value, 1), np.int64)
        if n.value:
            self._ck(self.lib.ambi_batch_all_orders(self.h, u, pass_, 0, n.value, out.ctypes.data_as(_P(C.c_int64))), "all_orders")
        return out[:n.value]

    def all_set_shard(self, rank, world):
        """--all with the 64-order chunks dealt over `world` ranks (include/ambigram_hip.h)."""
        self._ck(self.lib.ambi_batch_all_set_shard(self.h, rank, world), "all_set_shard")

    def all_device(self):
        """(address, bytes) of the pool the ranks merge after a sharded --all run: bitmaps + undefined-order flags."""
        p, n = C.c_void_p(), C.c_int64()
        self._ck(self.lib.ambi_batch_all_device(self.h, C.byref(p), C.byref(n)), "all_device")
        return p.value or 0, n.value

    def all_finish(self):
        self._ck(self.lib.ambi_batch_all_finish(self.h), "all_finish")

    def all_paths(self, u, pass_, first, count, stride):
        """--all: the paths of valid orders [first, first+count) of the pass (list of int32 arrays)."""
        lengths = np.zeros(max(count, 1), np.int32)
        cells = np.zeros(max(count * stride, 1), np.int32)
        self._ck(self.lib.ambi_batch_all_paths(self.h, u, pass_, first, count, lengths.ctypes.data_as(_P(C.c_int32)),
                                               cells.ctypes.data_as(_P(C.c_int32)), stride), "all_paths")
        out = []
        for j in range(count):
            if lengths[j] < 0:
                raise AmbiError(self.lib, int(lengths[j]), "all_paths")
            out.append(cells[j * stride: j * stride + lengths[j]].copy())
        return out

    def slices(self):
        return int(self.lib.ambi_batch_slices(self.h))

    def kernel_times(self):
        out = {}
        for i in range(self.lib.ambi_batch_kernel_count(self.h)):
            name, ms = C.c_char_p(), C.c_float()
            self.lib.ambi_batch_kernel_time(self.h, i, C.byref(name), C.byref(ms))
            out[name.value.decode()] = ms.value
        return out

    def kernel_spans(self):
        """{kernel: (start_ms, end_ms)} from the start of the run's first kernel (ambi_batch_kernel_span)."""
        out = {}
        for i in range(self.lib.ambi_batch_kernel_count(self.h)):
            name, ms, a, b = C.c_char_p(), C.c_float(), C.c_float(), C.c_float()
            self.lib.ambi_batch_kernel_time(self.h, i, C.byref(name), C.byref(ms))
            self.lib.ambi_batch_kernel_span(self.h, i, C.byref(a), C.byref(b))
            out[name.value.decode()] = (a.value, b.value)
        return out

    def traffic(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self.lib.ambi_batch_traffic(self.h, C.byref(a), C.byref(b), C.byref(c))
        return dict(input_bytes=a.value, order_bytes=b.value, result_bytes=c.value)


def read_sol(path):
    """Token scan of a CBC .sol (localhap.cpp:192-212) -> (infeasible, objective, cols, vals)."""
    cols, vals, infeasible, obj = [], [], False, 0.0
    with open(path) as f:
        toks = f.read().split()
    i = 0
    while i < len(toks):
        t = toks[i]
        if t == "Infeasible":
            infeasible = True
            break
        if t == "value" and i + 1 < len(toks):
            try:
                obj += float(toks[i + 1])
            except ValueError:
                pass
            i += 1
        elif t[0] == 'x' and i + 1 < len(toks):
            cols.append(int(t[1:]))
            vals.append(int(float(toks[i + 1])) if toks[i + 1].lstrip('-').replace('.', '', 1).isdigit() else 0)
            i += 1
        i += 1
    return infeasible, obj, cols, vals


def merge_out_juncs(acc, unit_list, increase=True):
    """localhap.cpp:267-289: merge one path's junction steps (already aggregated per unit) in first-appearance order."""
    for (u, v, c) in unit_list:
        for j in acc:
            if (j[0] == u and j[1] == v) or (j[0] == -v and j[1] == -u):
                if increase:
                    j[2] += c
                break
        else:
            acc.append([u, v, c])


def reconstruct_sample(lib, lh, sols, juncs="", reversed_=False, all_=False, first_budget=0, order_arena_bytes=-1,
                       target_lanes=0, keep_orders=False):
    """Host-side mirror of `Ambigram --op bfb` (localhap.cpp:49-388) with the external `cbc` call replaced by the given
    .sol files (one per chromosome that reaches the ILP, in order).  Returns a dict shaped like the oracle's dump."""
    g = Graph(lib, lh)
    trx_before = g.trx_before() is not None
    if trx_before:   # PROP I1 / C1: the reference leaves the rebuilt graph in ./new.lh (and says "write seg"); here: a scratch file
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            g.write_lh(os.path.join(td, "new.lh"))
    log = ["bfb"] + g.log()
    n_log0 = len(g.log())
    if juncs:
        g.read_juncs(juncs)
        log += g.log()[n_log0:]
    b = Batch(lib)
    b.configure(order_arena_bytes=order_arena_bytes, first_budget=first_budget, target_lanes=target_lanes)
    # a chromosome reaches the ILP unless it has no fold-back inversion; the engine decides (status SHORTCUT), so the
    # caller hands a .sol to every chromosome for which one exists, consuming them in order for non-shortcut units.
    # To know which chromosomes are shortcuts before assigning .sol files, run a solution-less probe batch first.
    probe = Batch(lib)
    for c in range(g.n_chr):
        probe.add_chromosome(g, c, [], [])
    probe.upload(); probe.run(0); probe.download()
    shortcut = [probe.unit_result(c)["status"] == ST_SHORTCUT for c in range(g.n_chr)]
    probe.close()
    cursor = 0
    for c in range(g.n_chr):
        if shortcut[c]:
            b.add_chromosome(g, c, [], [])
        else:
            if cursor >= len(sols):
                raise RuntimeError("missing .sol for chromosome %d" % c)
            b.add_chromosome_sol(g, c, sols[cursor])
            cursor += 1
    flags = (FLAG_REVERSED if reversed_ else 0) | (FLAG_ALL if all_ else 0)
    b.upload(); b.run(flags); b.download()
    # the final paths a second way: in run-length form as the finish kernels leave them for the host (ambi_batch_runs_to_host)
    b.runs_to_host(1, 0); b.runs_wait(0)
    res = dict(ok=True, err="", log=log, chr=[], paths=[], out_juncs=[], trx_run=False, trx_path=[])
    out_acc = []
    for c in range(g.n_chr):
        r = b.unit_result(c)
        s, e = g.chromosome(c)
        st = dict(start=s, end=e, status=r["status"], bias=r["bias"], num_orders=r["num_orders"], first_valid=r["first_valid"],
                  first_forward=r["first_forward"], evaluated=r["evaluated"], K=r["n_nodes"])
        if r["status"] < 0 or r["status"] == ST_NO_VALID_ORDER:
            res["ok"] = False
            res["err"] = lib.ambi_error_string(r["status"]).decode()
            res["chr"].append(st)
            res["paths"].append([])
            continue
        path = b.unit_path(c, 0)
        path_ind = b.unit_path(c, 1)
        prep = b.unit_prepare(c, e - s + 1)
        st.update(path=path.tolist(), path_indel=path_ind.tolist(), path_indel_from_runs=b.runs_unit_path(0, c).tolist(), bkp=b.unit_bkp(c).tolist(),
                  junc_cn=prep["junc_cn"], seg_cn=prep["seg_cn"], target_cn=prep["target_cn"], inv_junc=prep["inv_junc"],
                  indel_printed=bool(r["indel_printed"]), shortcut=r["status"] == ST_SHORTCUT,
                  infeasible=r["status"] == ST_INFEASIBLE)
        if r["status"] == ST_OK:
            pat, loop, succ = b.unit_dag(c, r["n_nodes"])
            st.update(node2pat=pat.tolist(), node2loop=loop.tolist(), succ=[int(x) for x in succ])
            if keep_orders:
                st["orders"] = b.unit_orders(c, 0, r["num_orders"], r["n_nodes"]).tolist()
        if r["status"] == ST_SHORTCUT:
            log.append(g.format_path(path))
        else:
            log += ["Declare done", "ILP formula done", "Variable constrains done"]   # BFB_ILP progress lines
            if all_ and r["status"] == ST_OK:
                # --all: one line per valid order, first orientation then (if it was run) the flipped one (LGM.cpp:3672-3695)
                st["all_paths"] = []
                for pass_ in (0, 1):
                    idx = b.all_orders(c, pass_)
                    for lo in range(0, len(idx), 64):
                        for p_all in b.all_paths(c, pass_, lo, min(64, len(idx) - lo), 2 * len(path) + 64):
                            st["all_paths"].append(p_all.tolist())
                            log.append(g.format_path(p_all))
            else:
                log.append(g.format_path(path))
            if r["status"] == ST_INFEASIBLE:
                log.append("ILP is unsolvable.")
            elif r["indel_printed"]:
                log.append("BFB path with insertion, deletion, or duplication:")
                log.append(g.format_path(path_ind))
        res["chr"].append(st)
        if trx_before and r["status"] == ST_OK:   # virusBFB (localhap.cpp:263): back to the segments of the file; junction steps counted there
            back, lines = g.trx_restore(path_ind)
            log += lines
            res["paths"].append(back.tolist())
            steps = []
            for i in range(len(back) - 1):
                u, v = int(back[i]), int(back[i + 1])
                if not (abs(abs(u) - abs(v)) == 1 and (u > 0) == (v > 0)):
                    steps.append((u, v, 1))
            merge_out_juncs(out_acc, steps)
            continue
        res["paths"].append(path_ind.tolist())
        merge_out_juncs(out_acc, b.unit_out_juncs(c))
    res["trx_before"] = trx_before
    ins_mode, con_mode, main_chr = g.props()
    if res["ok"] and (ins_mode == 2 or con_mode == 2):
        if not main_chr:
            res["ok"] = False
            res["err"] = "BFB-TRX without M:<chr> (reference segfaults)"
        else:
            trx, new_paths = g.translocation_bfb([np.asarray(p, np.int32) for p in res["paths"]])
            res["trx_run"] = True
            res["trx_path"] = trx.tolist()
            res["paths"] = [p.tolist() for p in new_paths]
            log.append("BFB with translocation:")
            log.append(g.format_path(trx))
            steps = []
            for i in range(len(trx) - 1):
                u, v = int(trx[i]), int(trx[i + 1])
                if not (abs(abs(u) - abs(v)) == 1 and (u > 0) == (v > 0)):
                    steps.append((u, v, 1))
            merge_out_juncs(out_acc, steps, increase=False)
    res["out_juncs"] = [tuple(j) for j in out_acc]
    b.close()
    g.close()
    return res


def reconstruct_sc(lib, lhs, sols, reversed_=False, all_=False, first_budget=0):
    """Host-side mirror of `Ambigram --op sc_bfb` (localhap.cpp:390-679) with the external `cbc` call replaced by the
    given joint .sol files (one per chromosome of the first graph that has a fold-back inversion, in order).
    All graphs x chromosomes run as ONE batch of units.  Returns a dict shaped like the oracle's run_sc_bfb dump."""
    graphs = [Graph(lib, p) for p in lhs]
    G = len(graphs)
    log = ["sc_bfb"]
    for g in graphs:
        log += g.log()
    n0 = len(graphs[0].log())
    rc = lib.ambi_graph_recalculate(graphs[0].h)                  # localhap.cpp:438-439
    if rc != 0:
        raise AmbiError(lib, rc, "recalculate")
    log += graphs[0].log()[n0:]
    g0 = graphs[0]
    probe = Batch(lib)                                            # which chromosomes reach the ILP: the FIRST graph decides (:505)
    for c in range(g0.n_chr):
        probe.add_chromosome(g0, c, [], [])
    probe.upload(); probe.run(0); probe.download()
    shortcut = [probe.unit_result(c)["status"] == ST_SHORTCUT for c in range(g0.n_chr)]
    probe.close()
    b = Batch(lib)
    if first_budget:
        b.configure(first_budget=first_budget)
    unit_of, cursor, sol_of = {}, 0, {}
    for c in range(g0.n_chr):
        if shortcut[c]:
            continue
        if cursor >= len(sols):
            raise RuntimeError("missing .sol for chromosome %d" % c)
        sol_of[c] = sols[cursor]
        cursor += 1
        for k, g in enumerate(graphs):
            rc = lib.ambi_batch_add_chromosome_sol_block(b.h, g.h, c, sol_of[c].encode(), k, G)
            if rc < 0:
                raise AmbiError(lib, rc, "add_chromosome_sol_block")
            unit_of[(c, k)] = rc
    res = dict(ok=True, err="", log=log, paths=[[] for _ in range(G)], trx_paths=[[] for _ in range(G)], chr=[])
    if unit_of:
        b.upload(); b.run((FLAG_REVERSED if reversed_ else 0) | (FLAG_ALL if all_ else 0)); b.download()
    for c in range(g0.n_chr):
        s, e = g0.chromosome(c)
        stages = []
        if shortcut[c]:                                           # reference path for every graph, nothing printed (:505-512)
            for k in range(G):
                res["paths"][k].append(list(range(s, e + 1)))
                stages.append(dict(shortcut=True, infeasible=False))
            res["chr"].append(stages)
            continue
        log += ["Declare done"] + ["ILP formula done"] * G + ["Variable constrains done"]
        first = b.unit_result(unit_of[(c, 0)])
        if first["status"] == ST_INFEASIBLE:                      # :543-551
            log.append("ILP is unsolvable.")
            for k in range(G):
                res["paths"][k].append(list(range(s, e + 1)))
                stages.append(dict(shortcut=False, infeasible=True))
            res["chr"].append(stages)
            continue
        for k, g in enumerate(graphs):
            u = unit_of[(c, k)]
            r = b.unit_result(u)
            if r["status"] != ST_OK:
                res["ok"] = False
                res["err"] = "graph %d chromosome %d: %s" % (k, c, lib.ambi_error_string(r["status"]).decode())
                res["paths"][k].append([])
                stages.append(dict(status=r["status"]))
                continue
            path, path_ind = b.unit_path(u, 0), b.unit_path(u, 1)
            pat, loop, _ = b.unit_dag(u, r["n_nodes"])
            stages.append(dict(shortcut=False, infeasible=False, num_orders=r["num_orders"], first_valid=r["first_valid"],
                               first_forward=r["first_forward"], evaluated=r["evaluated"], bkp=b.unit_bkp(u).tolist(),
                               node2pat=[[] if x[0] == 0 else x for x in pat.tolist()], node2loop=[[] if x[0] == 0 else x for x in loop.tolist()],
                               path=path.tolist(), path_indel=path_ind.tolist(), indel_printed=bool(r["indel_printed"])))
            if all_:
                for pass_ in (0, 1):
                    idx = b.all_orders(u, pass_)
                    for lo in range(0, len(idx), 64):
                        for p_all in b.all_paths(u, pass_, lo, min(64, len(idx) - lo), 2 * len(path) + 64):
                            log.append(g.format_path(p_all))
            else:
                log.append(g.format_path(path))
            if r["indel_printed"]:
                log += ["BFB path with insertion, deletion, or duplication:", g.format_path(path_ind)]
            res["paths"][k].append(path_ind.tolist())
        res["chr"].append(stages)
    ins, con, main_chr = g0.props()
    if res["ok"] and (ins == 2 or con == 2):                      # :661-664, every graph
        for k, g in enumerate(graphs):
            log.append("BFB with translocation:")
            trx, new_paths = g.translocation_bfb([np.array(p, np.int32) for p in res["paths"][k]])
            res["paths"][k] = [p.tolist() for p in new_paths]
            res["trx_paths"][k] = trx.tolist()
            log.append(g.format_path(trx))
    b.close()
    for g in graphs:
        g.close()
    return res


class IlpModel:
    """ILP of one chromosome (BFB_ILP, LGM.cpp:4397-4752) built on the host in closed form."""

    def __init__(self, lib, graph, chr_, seg_cn, junc_cn, bias, max_cn_total, juncs_info=False, device=False):
        """device=True: the entries are written by ambi_ilp_fill_kernel on the GPU (kernel_ms = its mean device time)."""
        self.lib = lib
        self.h = C.c_void_p()
        self.kernel_ms = None
        sc = np.ascontiguousarray(seg_cn, np.float64)
        jc = np.ascontiguousarray(np.asarray(junc_cn, np.float64).reshape(-1))
        if device:
            ms = C.c_float()
            rc = lib.ambi_ilp_build_device(graph.h, chr_, sc.ctypes.data_as(_P(C.c_double)), jc.ctypes.data_as(_P(C.c_double)), int(bias),
                                           float(max_cn_total), 1 if juncs_info else 0, C.byref(ms), C.byref(self.h))
            self.kernel_ms = ms.value
        else:
            rc = lib.ambi_ilp_build(graph.h, chr_, sc.ctypes.data_as(_P(C.c_double)), jc.ctypes.data_as(_P(C.c_double)), int(bias),
                                    float(max_cn_total), 1 if juncs_info else 0, C.byref(self.h))
        if rc != 0:
            raise AmbiError(lib, rc, "ilp_build")
        self._sizes()

    @classmethod
    def joint(cls, lib, graph0, chr_, seg_cn, fold_cn):
        """Joint model of `--op sc_bfb` (BFB_ILP_SC, LGM.cpp:4754-5093): seg_cn / fold_cn are G x n arrays (graph-major)."""
        self = cls.__new__(cls)
        self.lib, self.h, self.kernel_ms = lib, C.c_void_p(), None
        sc = np.ascontiguousarray(seg_cn, np.float64)
        fc = np.ascontiguousarray(fold_cn, np.float64)
        rc = lib.ambi_ilp_build_sc(graph0.h, chr_, sc.shape[0], sc.ctypes.data_as(_P(C.c_double)), fc.ctypes.data_as(_P(C.c_double)), C.byref(self.h))
        if rc != 0:
            raise AmbiError(lib, rc, "ilp_build_sc")
        self._sizes()
        return self

    def _sizes(self):
        r, z, c, i = C.c_int64(), C.c_int64(), C.c_int32(), C.c_int32()
        self.lib.ambi_ilp_sizes(self.h, C.byref(r), C.byref(z), C.byref(c), C.byref(i))
        self.n_rows, self.nnz, self.n_cols, self.n_int = r.value, z.value, c.value, i.value

    def arrays(self):
        rp = np.zeros(self.n_rows + 1, np.int64); col = np.zeros(self.nnz, np.int32); val = np.zeros(self.nnz)
        rlo, rup = np.zeros(self.n_rows), np.zeros(self.n_rows)
        clo, cup, obj = np.zeros(self.n_cols), np.zeros(self.n_cols), np.zeros(self.n_cols)
        p = lambda a, t: a.ctypes.data_as(_P(t))
        self.lib.ambi_ilp_copy(self.h, p(rp, C.c_int64), p(col, C.c_int32), p(val, C.c_double), p(rlo, C.c_double), p(rup, C.c_double),
                               p(clo, C.c_double), p(cup, C.c_double), p(obj, C.c_double))
        return dict(row_ptr=rp, col=col, val=val, row_lo=rlo, row_up=rup, col_lo=clo, col_up=cup, obj=obj)

    def write_mps(self, path):
        rc = self.lib.ambi_ilp_write_mps(self.h, path.encode())
        if rc != 0:
            raise AmbiError(self.lib, rc, "write_mps")

    def write_lp(self, path):
        rc = self.lib.ambi_ilp_write_lp(self.h, path.encode())
        if rc != 0:
            raise AmbiError(self.lib, rc, "write_lp")

    def close(self):
        if self.h:
            self.lib.ambi_ilp_destroy(self.h)
            self.h = C.c_void_p()
