"""Exhaustive search (CPU oracle only) for inputs on which the topological orders of one unit differ in validity.

Question: does `getBFB` (LGM.cpp:3514-3697) ever meet an INVALID order before a VALID one, i.e. is
first_valid > 0 reachable?  Every element set over n segments with up to K selected patterns p(a,b) / loops l(a,b,cn) is
run through the oracle with --all in both orientations; a unit is "mixed" when, within one pass, some orders are valid
and some are not.  Not a pytest: run by hand, results quoted in DESIGN.md and frozen into tests/golden/mixed_validity.json.

    python tests/tools/search_mixed_validity.py --n 4 --kmax 4 [--cn2] [--limit N] [--out file.json]
"""
import argparse
import itertools
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_py  # noqa: E402


def rank_ab(a, b, n):
    return (a - 1) * n - (a - 1) * (a - 2) // 2 + (b - a)


def write_lh(path, n):
    L = ["SAMPLE_NAME mix", "AVG_CHR_SEG_DP 30", "AVG_WHOLE_HOST_DP 30", "AVG_JUNC_DP 30", "PURITY 1", "AVG_TUMOR_PLOIDY 2",
         "PLOIDY 2m1", "VIRUS_START %d" % (n + 1), "SOURCE 1", "SINK %d" % n]
    for i in range(1, n + 1):
        L.append("SEG H:%d:chr1:%d:%d 60.0 4.0" % (i, i * 1000, i * 1000 + 999))
    for i in range(1, n):
        L.append("JUNC H:%d:+ H:%d:+ 30.0 1.0 U B" % (i, i + 1))
    for i in range(1, n + 1):                      # perfect fold-backs on both sides of every segment: no imperfectFBI rewrite
        L.append("JUNC H:%d:+ H:%d:- 30.0 1.0 U B" % (i, i))
        L.append("JUNC H:%d:- H:%d:+ 30.0 1.0 U B" % (i, i))
    with open(path, "w") as f:
        f.write("\n".join(L) + "\n")


def write_sol(path, els, n):
    num_pat = n * (n + 1) // 2
    rows = sorted((rank_ab(a, b, n) + (num_pat if lp else 0), cn) for (lp, a, b, cn) in els)
    with open(path, "w") as f:
        f.write("Optimal - objective value 0.00000000\n")
        for col, cn in rows:
            f.write("%7d x%-7d %15d %15d\n" % (col, col, cn, 0))


def classify(oc):
    """(mixed, detail) from an --all run of the oracle."""
    R = oc["num_orders"]
    idx = oc["all_eval_idx"]
    p0 = [i for i in idx if i < R]
    p1 = [i - R for i in idx if i >= R]
    mixed = (0 < len(p0) < R) or (0 < len(p1) < R)
    return mixed, dict(R=R, pass0=p0, pass1=p1, first_valid=oc["first_valid"], evaluated=oc["evaluated"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4)
    ap.add_argument("--kmin", type=int, default=2)
    ap.add_argument("--kmax", type=int, default=4)
    ap.add_argument("--cn2", action="store_true", help="loops also with copy number 2")
    ap.add_argument("--limit", type=int, default=0, help="stop after this many mixed units")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    n = a.n
    pool = []
    for x in range(1, n + 1):
        for y in range(x, n + 1):
            pool.append((0, x, y, 1))
            pool.append((1, x, y, 1))
            if a.cn2:
                pool.append((1, x, y, 2))
    tmp = tempfile.mkdtemp(prefix="mixsearch")
    lh, sol = os.path.join(tmp, "m.lh"), os.path.join(tmp, "m.sol")
    write_lh(lh, n)
    tried = mixed_n = valid_n = ub_n = 0
    found = []
    for K in range(a.kmin, a.kmax + 1):
        for els in itertools.combinations(pool, K):
            keys = set((lp, x, y) for (lp, x, y, cn) in els)
            if len(keys) < K:
                continue            # the same loop twice with different cn is one .sol column
            write_sol(sol, els, n)
            for rev in (False, True):
                o = oracle_py.run_bfb(lh, [sol], reversed_=rev, all_=True)
                oc = o["chr"][0]
                tried += 1
                if oc["ub"]:
                    ub_n += 1
                    continue
                if oc["first_valid"] >= 0:
                    valid_n += 1
                m, d = classify(oc)
                if m:
                    mixed_n += 1
                    if len(found) < 200:
                        found.append(dict(n=n, elements=[list(e) for e in els], reversed=rev, **d))
                    if mixed_n <= 5:
                        print("MIXED", els, "rev" if rev else "fwd", d, flush=True)
            if a.limit and mixed_n >= a.limit:
                break
        print("K=%d done: tried %d, with a valid order %d, mixed %d, ub %d" % (K, tried, valid_n, mixed_n, ub_n), flush=True)
        if a.limit and mixed_n >= a.limit:
            break
    if a.out:
        with open(a.out, "w") as f:
            json.dump(dict(n=n, tried=tried, valid=valid_n, mixed=mixed_n, ub=ub_n, found=found), f)


if __name__ == "__main__":
    main()
