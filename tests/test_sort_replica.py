"""ambi_sort.hpp must replay libstdc++'s std::sort operation for operation: the reference's comparator
(LocalGenomicMap.cpp:3267-3274) is not a strict weak order, so the permutation depends on the algorithm itself."""
import ctypes as C
import random

import numpy as np


def _both(lib, recs):
    n = len(recs)
    a = np.ascontiguousarray(np.array(recs, np.int32).reshape(-1, 3)) if n else np.zeros((0, 3), np.int32)
    e = np.zeros((max(n, 1), 3), np.int32)
    s = np.zeros((max(n, 1), 3), np.int32)
    p = lambda x: x.ctypes.data_as(C.POINTER(C.c_int))
    lib.hostsim_sort_both.restype = C.c_int
    ub = lib.hostsim_sort_both(p(a), n, p(e), p(s))
    return ub, e[:n].tolist(), s[:n].tolist()


def test_sort_matches_libstdcxx(hostsim_lib):
    rng = random.Random(1234)
    for trial in range(3000):
        n = rng.choice([0, 1, 2, 5, 9, 15, 16, 17, 18, 24, 31, 32, 33, 40, 48, 63, 64, 65, 80, 100, 127, 128, 129, 160, 200, 254, 255])   # (above 63: the memory form, units with 64..255 nodes)
        recs = []
        for _ in range(n):
            if rng.random() < 0.35:
                recs.append([0, 0, 0])                       # empty slot (pattern position)
            else:
                a = rng.randint(1, 40 if n < 64 else 200)
                b = rng.randint(a, min(60 if n < 64 else 600, a + rng.choice([0, 1, 2, 3, 5, 8, 20, 150])))
                recs.append([a, b, rng.randint(1, 3)])
        ub, eng, std = _both(hostsim_lib, recs)
        assert ub == 0
        assert eng == std, (n, recs)


def test_sort_reference_layout_loops_then_patterns(hostsim_lib):
    # std::map order puts every "l:" key before every "p:" key, i.e. loops first, empties last
    rng = random.Random(7)
    for n in range(1, 256):
        k = rng.randint(0, n)
        recs = [[rng.randint(1, 30), 0, 1] for _ in range(k)] + [[0, 0, 0]] * (n - k)
        for r in recs:
            if r[0]:
                r[1] = r[0] + rng.randint(0, 25)
        ub, eng, std = _both(hostsim_lib, recs)
        assert ub == 0 and eng == std


def test_key_order_matches_std_string(hostsim_lib):
    hostsim_lib.hostsim_key_less.restype = C.c_int
    rng = random.Random(5)
    for _ in range(5000):
        a = (rng.randint(0, 1), rng.choice([1, 2, 9, 10, 11, 19, 99, 100, 101, 255, 1000]), rng.choice([1, 2, 10, 20, 100, 256, 1024]))
        b = (rng.randint(0, 1), rng.choice([1, 2, 9, 10, 11, 19, 99, 100, 101, 255, 1000]), rng.choice([1, 2, 10, 20, 100, 256, 1024]))
        ka = ("l:" if a[0] else "p:") + "%d,%d" % (a[1], a[2])
        kb = ("l:" if b[0] else "p:") + "%d,%d" % (b[1], b[2])
        assert bool(hostsim_lib.hostsim_key_less(*a, *b)) == (ka < kb), (ka, kb)
