"""Expectations derived BY HAND from the reference source (the walk is in each docstring), so that they do not depend on
either transcription: both the product's host code (lh_graph.cpp, ambi_pack.cpp; run here through the host-simulation
build of the same sources) and the oracle are checked against them.  Covers the host twins named in VERDICT r1:
translocationBFB incl. its insertion branch (LGM.cpp:4052-4193), the .sol scan (localhap.cpp:196-211)."""
import os

import numpy as np
import pytest

from ambigram_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _two_chr_lh(path, juncs, prop="PROP I2:chr1:chr2 M:chr1"):
    """chr1 = segments 1..4, chr2 = segments 5..7; only the inter-chromosome junctions matter to translocationBFB."""
    L = ["SAMPLE_NAME trx", "AVG_CHR_SEG_DP 30", "AVG_WHOLE_HOST_DP 30", "AVG_JUNC_DP 30", "PURITY 1", "AVG_TUMOR_PLOIDY 2",
         "PLOIDY 2m1", "VIRUS_START 8", "SOURCE 1,5", "SINK 4,7"]
    for i in range(1, 8):
        c = 1 if i <= 4 else 2
        L.append("SEG H:%d:chr%d:%d:%d 30.0 2.0" % (i, c, i * 100, i * 100 + 99))
    for (a, ad, b, bd) in juncs:
        L.append("JUNC H:%d:%s H:%d:%s 30.0 1.0 U B" % (a, ad, b, bd))
    L.append(prop)
    with open(path, "w") as f:
        f.write("\n".join(L) + "\n")
    return path


P0 = [1, 2, 3, 4, -4, -3, -2, -1]        # a BFB path of chr1 (loop l(1,4) once, forward seed)
P1_LOOP = [5, 6, 7, -7, -6, -5]          # a BFB path of chr2 (loop l(5,7))
P1_PLAIN = [5, 6, 7]                     # the reference path of a chromosome without fold-backs (localhap.cpp:164-170)

# name -> (inter-chromosome junctions in file order, per-chromosome paths, expected result, expected paths[1] afterwards,
#          expected printed line)
TRX_CASES = {
    # T1.  sv = [J1: 2+ -> 5+, J2: 7+ -> 3+] (LGM.cpp:4057-4060).  res = P0 (:4062-4065), startPos = 0.
    #  group: J1's source is on the main chromosome -> [edgeA.src, edgeA.tgt] = [2+, 5+] (:4070-4075).  Chain (:4085-4097):
    #  J2's edge A is 7+ -> 3+, its source lies on the chromosome of group.back() = 5+ -> push 7+, 3+; back is on the main
    #  chromosome -> stop.  group = [2+, 5+, 7+, 3+]: size 4 -> insertion branch (:4120).
    #  front id 2 <= back id 3: no reversal (:4121).  flag = find(res, 2+) = index 1 (:4126).  i = 1: paths[chr2],
    #  pos1 = first 5+ = index 0 (:4131); pos2 = last 7+ = index 2, pos2.base() = 3, pos1 > base? no (:4138-4144) -> push 2.
    #  last = find(from flag+1, 3+) = index 2 (:4147).  4 positions for 4 group entries: ok.
    #  temp = paths[chr2][0..2] = 5+ 6+ 7+ (:4180).  erase res[2, 2) = nothing (:4182), insert at 2 (:4183).
    "T1 insertion, forward": ([(2, '+', 5, '+'), (7, '+', 3, '+')], [P0, P1_LOOP],
                              [1, 2, 5, 6, 7, 3, 4, -4, -3, -2, -1], P1_LOOP,
                              "1+2+||5+6+7+||3+4+|4-3-2-1-"),
    # T2.  J1: 2+ -> 7-, J2: 5- -> 3+.  group = [2+, 7-] then J2's edge A source 5- is on chr2 -> [2+, 7-, 5-, 3+].
    #  paths[chr2] = 5+ 6+ 7+ has no 7-: reversed and complemented in place -> 7- 6- 5- (:4132-4136), pos1 = 0;
    #  last 5- = index 2, base 3 -> push 2.  temp = 7- 6- 5-.
    "T2 insertion, other chromosome reverse-complemented": ([(2, '+', 7, '-'), (5, '-', 3, '+')], [P0, P1_PLAIN],
                              [1, 2, -7, -6, -5, 3, 4, -4, -3, -2, -1], [-7, -6, -5],
                              "1+2+||7-6-5-||3+4+|4-3-2-1-"),
    # T3.  J1: 3- -> 5+, J2: 7+ -> 2-.  group = [3-, 5+] + [7+, 2-].  front id 3 > back id 2 -> the group is reversed and
    #  complemented (:4121-4124): [2+, 7-, 5-, 3+] -- from here on T2.
    "T3 insertion, group reversed": ([(3, '-', 5, '+'), (7, '+', 2, '-')], [P0, P1_PLAIN],
                              [1, 2, -7, -6, -5, 3, 4, -4, -3, -2, -1], [-7, -6, -5],
                              "1+2+||7-6-5-||3+4+|4-3-2-1-"),
    # T4.  Two insertions; the second needs the retry with the reverse-complemented group (:4148-4176).
    #  sv = [J1: 2+ -> 5+, J2: 7+ -> 3+, J3: 1+ -> 6+, J4: 6+ -> 2+].  Group 1 = [2+,5+,7+,3+] as in T1 -> res =
    #  1+ 2+ 5+ 6+ 7+ 3+ 4+ 4- 3- 2- 1-, startPos = find(res, temp.back() = 7+) = index 4 (:4184).
    #  Group 2: J3's source is on the main chromosome -> [1+, 6+]; J4's edge A 6+ -> 2+ starts on chr2 -> [1+,6+,6+,2+].
    #  Attempt 1: flag = find(from 4, 1+) = end (the only 1+ is at index 0) -> no path positions; last = find(end+1, end)
    #  = end (libstdc++: empty loop) -> 2 positions < 4 -> retry with [2-, 6-, 6-, 1-]: flag = find(from 4, 2-) = 9;
    #  paths[chr2] = 5+ 6+ 7+ 7- 6- 5-: first 6- = 4, last 6- = 4 (base 5) -> push 4, 4; last = find(from 10, 1-) = 10.
    #  temp = [6-]; erase res[10, 10) nothing; insert at 10.
    "T4 two insertions, second one by retry": ([(2, '+', 5, '+'), (7, '+', 3, '+'), (1, '+', 6, '+'), (6, '+', 2, '+')], [P0, P1_LOOP],
                              [1, 2, 5, 6, 7, 3, 4, -4, -3, -2, -6, -1], P1_LOOP,
                              "1+2+||5+6+7+||3+4+|4-3-2-||6-||1-"),
    # T5.  Insertion that finds no place.  Main path = the plain 1+ 2+ 3+ 4+; J1: 2- -> 5+, J2: 7+ -> 3-.
    #  group = [2-, 5+, 7+, 3-], front id 2 <= back id 3.  res has no 2-: flag = end, 2 positions < 4 -> retry with
    #  [3+, 7-, 5-, 2+]: flag = find(3+) = 2; paths[chr2] = 5+ 6+ 7+ has no 7- -> complemented in place to 7- 6- 5-,
    #  pos1 = 0, pos2 = 2; last = find(from 3, 2+) = end -> `continue` (:4177): res unchanged, but paths[chr2] stays
    #  reverse-complemented.
    "T5 insertion skipped": ([(2, '-', 5, '+'), (7, '+', 3, '-')], [[1, 2, 3, 4], P1_PLAIN],
                              [1, 2, 3, 4], [-7, -6, -5],
                              "1+2+3+4+"),
    # T6.  Concatenation (:4099-4119): one junction 3+ -> 6+.  group = [3+, 6+], size 2.  pos1 = last 3+ in res = index 2
    #  -> res cut to 1+ 2+ 3+ (:4107); paths[chr2]: first 6+ = index 1 -> append 6+ 7+ 7- 6- 5- (:4117).
    "T6 concatenation": ([(3, '+', 6, '+')], [P0, P1_LOOP],
                              [1, 2, 3, 6, 7, -7, -6, -5], P1_LOOP,
                              "1+2+3+||6+7+|7-6-5-"),
    # T7.  Concatenation whose junction is written from the other chromosome: 6- -> 3-.  Its TARGET is on the main
    #  chromosome -> group = [edgeB.src, edgeB.tgt] = complement edge 3+ -> 6+ (:4076-4081): same as T6.
    "T7 concatenation through the complement edge": ([(6, '-', 3, '-')], [P0, P1_LOOP],
                              [1, 2, 3, 6, 7, -7, -6, -5], P1_LOOP,
                              "1+2+3+||6+7+|7-6-5-"),
}


@pytest.mark.parametrize("name", sorted(TRX_CASES))
def test_translocation_bfb_hand_derived(hostsim_lib, oracle, workdir, name):
    juncs, paths, want, want_p1, line = TRX_CASES[name]
    lh = _two_chr_lh(os.path.join(workdir, "trx_%s.lh" % name.split()[0]), juncs)
    g = api.Graph(hostsim_lib, lh)
    res, new_paths = g.translocation_bfb([np.array(p, np.int32) for p in paths])
    assert res.tolist() == want, name
    assert new_paths[1].tolist() == want_p1, name
    assert g.format_path(res) == line, name
    g.close()
    o = oracle.translocation(lh, paths)
    assert o["path"] == want and o["paths"][1] == want_p1 and o["line"] == line, (name, o)


def test_translocation_branches_all_seen(oracle, workdir):
    """the hand-derived cases reach every branch of the function (trace of the oracle's restatement)"""
    seen = set()
    for name, (juncs, paths, _, _, _) in TRX_CASES.items():
        lh = _two_chr_lh(os.path.join(workdir, "trxb_%s.lh" % name.split()[0]), juncs)
        seen.update(oracle.translocation(lh, paths)["trace"])
    assert {"concat", "insert", "insert-retry", "insert-skip"} <= seen, seen


def test_sol_scan_hand_derived(hostsim_lib, oracle, workdir):
    """The .sol token scan, localhap.cpp:196-211, on the README graph (6 segments: numPat = 21, numComp = 42):
      * a token starting with 'x' is a column NAME; x = stoi(name.substr(1)); only x < numComp is read further (:205-206),
        then the NEXT token is the copy number, through stoi (:207-208): "1.9" -> 1, the reduced-cost column is skipped
        as an ordinary token;
      * the leading index token ("26") starts with a digit: ignored;
      * the same column twice: the later line overwrites (:209);
      * x60 >= numComp (an epsilon column): its value is NOT consumed;
      * a column with value 0 selects nothing (constructDAG takes elementCN > 0, LGM.cpp:3281).
    Columns (localhap.cpp:122-133, rank(a,b) over a <= b in lexicographic order, loops offset by numPat = 21):
      26 = l(1,6), 29 = l(2,4), 31 = l(2,6), 33 = l(3,4), 2 = p(1,3).
    Expected selection: l(1,6) cn 1, l(2,4) cn 2 (second line wins), l(2,6) cn 1, l(3,4) cn 1; p(1,3) has value 0."""
    sol = os.path.join(workdir, "scan.sol")
    with open(sol, "w") as f:
        f.write("Optimal - objective value 3.50000000\n"
                "      2 x2                      0                       0\n"
                "     26 x26                   1.9                       0\n"
                "     29 x29                     1                       0\n"
                "     60 x60                     5                       0\n"
                "     31 x31                     1                       0\n"
                "     33 x33                     1                       0\n"
                "     29 x29                     2                       0\n")
    lh = os.path.join(ROOT, "tests/data/readme6.lh")
    want = sorted([(1, 6, 1), (2, 4, 2), (2, 6, 1), (3, 4, 1)])
    g = api.Graph(hostsim_lib, lh)
    b = api.Batch(hostsim_lib)
    b.add_chromosome_sol(g, 0, sol)
    b.upload(); b.run(0); b.download()
    r = b.unit_result(0)
    assert r["n_nodes"] == 4
    pat, loop, _ = b.unit_dag(0, 4)
    assert all(p[0] == 0 for p in pat)
    assert sorted(tuple(int(v) for v in l) for l in loop) == want
    b.close(); g.close()
    oc = oracle.run_bfb(lh, [sol])["chr"][0]
    assert sorted(tuple(l) for l in oc["node2loop"] if l) == want and not any(oc["node2pat"])
