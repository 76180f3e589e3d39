"""Expectations derived BY HAND from the reference source for the stages that no reference-held vector touches
(missing #3): `imperfectFBI` (LGM.cpp:3431-3512: the plain rewrite :3442-3471 incl. the adjustment :3468, the
rewrite of a palindrome from its middle :3473-3508 incl. the `p1 > pos-1` guard), every branch of `indelBFB`
(LGM.cpp:3746-3837: deletion :3781-3793, duplication :3794-3805, inversion :3807-3818, insertion :3820-3832, the second attempt
with the reversed + complemented group, the distance limits) and `getIndelBias` (LGM.cpp:3699-3744, :3732-3742).

Every case is a complete input (a .lh and the .sol of its chromosome) small enough to walk the reference by hand; the walk
is written above the case.  The PRODUCT (stage code through the C ABI: on the CPU the host simulation of the same sources, on
the GPU libambigram_hip.so) and the ORACLE are both compared with the hand-derived values, so the two transcriptions are not
only compared with each other.  Notation: a breakpoint path is a list of signed segment ids, cell pairs (first, last);
`inv[i]` = the fold-back junction recorded for segment i by getJuncCN (LGM.cpp:4012-4049: the first free of (source, target)
in file order, then every end still unrecorded)."""
import os

import pytest

from ambigram_amd import api


def _lh(path, n, juncs, cn=2.0):
    L = ["SAMPLE_NAME hand", "AVG_CHR_SEG_DP 30", "AVG_WHOLE_HOST_DP 30", "AVG_JUNC_DP 30", "PURITY 1", "AVG_TUMOR_PLOIDY 2", "PLOIDY 2m1",
         "VIRUS_START %d" % (n + 1), "SOURCE 1", "SINK %d" % n]
    for i in range(1, n + 1):
        L.append("SEG H:%d:chr1:%d:%d 60.0 %.1f" % (i, i * 1000 + 1, i * 1000 + 1000, cn))
    for i in range(1, n):
        L.append("JUNC H:%d:+ H:%d:+ 30.0 1.0 U B" % (i, i + 1))
    for (a, ad, b, bd) in juncs:
        L.append("JUNC H:%d:%s H:%d:%s 30.0 1.0 U B" % (a, ad, b, bd))
    with open(path, "w") as f:
        f.write("\n".join(L) + "\n")
    return path


def _sol(path, n, elems):
    """elems: (kind 'p' | 'l', a, b, cn); column numbering of localhap.cpp:117-133: rank(a,b) over a <= b in lexicographic order,
    patterns first, loops offset by numPat = n(n+1)/2."""
    num_pat = n * (n + 1) // 2
    rank = lambda a, b: (a - 1) * n - (a - 1) * (a - 2) // 2 + (b - a)
    with open(path, "w") as f:
        f.write("Optimal - objective value 0.00000000\n")
        for (k, a, b, c) in elems:
            col = rank(a, b) + (num_pat if k == "l" else 0)
            f.write("%7d x%-6d %22d %22d\n" % (col, col, c, 0))
    return path


def expand(bkp):
    out = []
    for j in range(0, len(bkp), 2):
        a, b = bkp[j], bkp[j + 1]
        out += list(range(a, b + 1)) if a > 0 else [-k for k in range(-a, -b - 1, -1)]   # LGM.cpp:3661-3670
    return out


P = [1, 2, 3, 4, 5, 6, -6, -5, -4, -3, -2, -1]            # loop l(1,6) once, forward seed, perfect fold-backs at both ends
TELO6 = [(6, '+', 6, '-'), (1, '-', 1, '+')]               # those two fold-backs
Q = [1, 2, 3, 4, 5, 6, -6, -5, -4, -3, -2]                 # p(1,6) then p(2,6): an asymmetric path (no 1-)
TELOQ = [(6, '+', 6, '-'), (2, '-', 2, '+')]

# name -> dict(n, juncs, elems, bkp, path, path_indel, printed, cn_delta {segment: delta after getIndelBias}, bias)
CASES = {
    # ---------------------------------------------------------------- getBFB, loop placement (:3586-3644): a candidate that fails the nesting test
    # G1.  6 segments; loops l(1,6), l(2,6), l(3,6), all cn 1, perfect fold-backs at 6 and 1.  Keys "l:1,6" < "l:2,6" < "l:3,6", lengths 5, 4, 3
    #  (the sort leaves them), edges by the shared end 6: 0->1, 0->2, 1->2: one order 0 1 2.  Forward: seed 1+ 6+ 6- 1-  (:3548-3566).
    #  l(2,6): v1 = 2- is not in the path -> v2 = 6+, searched from the back (:3599-3604): cell 1 (odd; 1 < L-2 = 2 is false: no test)
    #   -> 6- 2- 2+ 6+ goes in behind it:  1+ 6+ 6- 2- 2+ 6+ 6- 1-.
    #  l(3,6): no 3-; 6+ from the back: cell 5 is odd, 5 < L-2 = 6: id(cell 4) = 2 > id(cell 7) = 1 -> SKIPPED (:3601); next 6+ is cell 1:
    #   id(cell 0) = 1 > id(cell 3) = 2 is false -> taken; 6- 3- 3+ 6+ behind it:  1+ 6+ 6- 3- 3+ 6+ 6- 2- 2+ 6+ 6- 1-.
    #   (in-place fix-ups :3639-3641: cell 1 := 6+, the cell behind it := 6-: the values they hold.)
    #  imperfectFBI: pos 0: 1- is found at cell 11 and cell 10 = 6- = -cell 1: one palindrome over everything, mid = 5; iterations
    #   p1 = 5 (6+: perfect fold-back, cells 5, 6 keep 6+ 6-), p1 = 3 (3-: NO fold-back recorded at 3: nothing), p1 = 1 (6+: cells 1, 2
    #   and the mirrored 10, 9 keep their values).  Nothing changes.
    "G1 loop placement: the last candidate fails the nesting test": dict(n=6, juncs=TELO6,
        elems=[("l", 1, 6, 1), ("l", 2, 6, 1), ("l", 3, 6, 1)], bkp=[1, 6, -6, -3, 3, 6, -6, -2, 2, 6, -6, -1], bias=1),
    # G2.  The same with perfect fold-backs at 2 and 3 as well: the placement is the same, but now iteration p1 = 3 of the palindrome
    #  rewrite finds inv[3] (3- -> 3+): w0 = 3-, w1 = 3+ into cells 3, 4 (unchanged) AND, mirrored about mid (:3493-3500: p2 = 8),
    #  cell 8 = -w0 = 3+, cell 7 = -w1 = 3-: the right half, which held 2- 2+, now repeats the left one -- imperfectFBI takes the two
    #  halves of whatever lies between a vertex and its complement for mirror images.
    "G2 palindrome rewrite mirrors the left half over a different right half": dict(n=6, juncs=[(6, '+', 6, '-'), (1, '-', 1, '+'), (2, '-', 2, '+'), (3, '-', 3, '+')],
        elems=[("l", 1, 6, 1), ("l", 2, 6, 1), ("l", 3, 6, 1)], bkp=[1, 6, -6, -3, 3, 6, -6, -3, 3, 6, -6, -1], bias=1),
    # ---------------------------------------------------------------- imperfectFBI, plain rewrite (:3442-3471)
    # F1.  7 segments; elements p(1,6), p(3,6), p(3,5).  Nodes in string-key order (LGM.cpp:3279): 0 = p:1,6  1 = p:3,5  2 = p:3,6;
    #  edges (:3314-3324) 0->2 (same end, longer), 2->1 (same start, longer): the only order is 0,2,1.  getBFB forward (:3527-3585):
    #  seed 1+ 6+; p(3,6): the path ends with 6+ = its end -> 6- 3-; p(3,5): ends with 3- = its start -> 3+ 5+.
    #  bkp = 1+ 6+ | 6- 3- | 3+ 5+.   Fold-backs: A = 6+ -> 7- (inv[6] = inv[7] = A), B = 3- -> 3+ (inv[3] = B), C = 5+ -> 4- (inv[5] = inv[4] = C).
    #  imperfectFBI: pos 0: find(pos+3, 1-) fails -> plain branch.  Cell 1 = 6+: A.source 6 < A.target 7 -> source+ = 6+ (:3445-3447).
    #   pos 2 (6- 3-): find(cell 5.., 6+) fails.  Cell 3 = 3-: B is perfect -> 3-.  Cell 2 = 6-: inv[6] = A and cell 1 still has id 6
    #   (:3460): A.source is 6 -> the TARGET on the same strand = 7- (:3461-3462).  No adjustment (7 > 3).
    #   pos 4 (3+ 5+): find(pos+3 > end) returns end (libstdc++).  Cell 5 = 5+: C.source 5 > C.target 4 -> target+ = 4+ (:3448-3449).
    #   Cell 4 = 3+: inv[3] = B, cell 3 has id 3: B.source is 3 -> target+ = 3+.  3 < 4: no adjustment.
    #  bias (localhap.cpp:141-146) = 1 + A (cn 1, ends differ) + C = 3.
    "F1 plain rewrite: both cells of a turn": dict(n=7, juncs=[(6, '+', 7, '-'), (3, '-', 3, '+'), (5, '+', 4, '-')],
        elems=[("p", 1, 6, 1), ("p", 3, 6, 1), ("p", 3, 5, 1)], bkp=[1, 6, -7, -3, 3, 4], bias=3),
    # F2.  The adjustment (:3468).  7 segments; p(1,7), p(5,7), p(5,6): nodes 0 = p:1,7  1 = p:5,6  2 = p:5,7, edges 0->2, 2->1,
    #  order 0,2,1; bkp = 1+ 7+ | 7- 5- | 5+ 6+.  Fold-backs: T = 7+ -> 7- (perfect), D = 6+ -> 4- (two apart: still a fold-back,
    #  LGM.cpp:4013; inv[6] = inv[4] = D).  pos 0, pos 2: nothing changes (T is perfect; no fold-back at 5).
    #  pos 4 (5+ 6+): cell 5 = 6+: D.source 6 > D.target 4 -> target+ = 4+.  Cell 4 = 5+: no inv[5].  Now the pair reads 5+ 4+:
    #  '+' and 5 > 4 -> cell 5 = cell 4 = 5+ (:3468).   bias = 1 + D = 2.
    "F2 plain rewrite: adjustment of an overshooting end": dict(n=7, juncs=[(7, '+', 7, '-'), (6, '+', 4, '-')],
        elems=[("p", 1, 7, 1), ("p", 5, 7, 1), ("p", 5, 6, 1)], bkp=[1, 7, -7, -5, 5, 5], bias=2),
    # ---------------------------------------------------------------- imperfectFBI, palindrome from the middle (:3473-3508)
    # F3.  6 segments; loops l(1,6), l(2,6).  node2loop after the sort by length (:3303): 0 = l(1,6), 1 = l(2,6); edge 0->1; order 0,1.
    #  Seed 1+ 6+ 6- 1-.  l(2,6): no 2- in the path; searching 6+ from the back finds cell 1 (even offset, :3598-3603): the loop
    #  6- 2- 2+ 6+ goes in behind it: bkp = 1+ 6+ | 6- 2- | 2+ 6+ | 6- 1-.
    #  Fold-backs: L = 3- -> 2+ (inv[3] = inv[2] = L), R = 6+ -> 5- (inv[6] = inv[5] = R).
    #  imperfectFBI pos 0: r = the closing 1- (cell 7), cell 6 = 6- = complement of cell 1: palindrome.  Middle p1 = cell 3, p2 = cell 4.
    #   p1 = 3 (2-, '-'): L.source 3 > L.target 2 -> cell 3 = source- = 3-, cell 4 = target+ = 2+ (:3493-3496); p2 = p1+1: no mirror.
    #   p1 = 1 (6+, '+'): R.source 6 > R.target 5 -> cell 1 = target+ = 5+, cell 2 = source- = 6- (:3485-3488); mirror (:3498-3501):
    #    cell 6 = complement(cell 1) = 5-, cell 5 = complement(cell 2) = 6+.   p1 = -1: stop.
    #  bias = 1 + L + R = 3.
    "F3 palindrome rewrite: two layers, mirrored cells": dict(n=6, juncs=[(3, '-', 2, '+'), (6, '+', 5, '-')],
        elems=[("l", 1, 6, 1), ("l", 2, 6, 1)], bkp=[1, 5, -6, -3, 2, 6, -5, -1], bias=3),
    # F4.  The guard `p1 > pos-1` (:3499).  Loops l(1,4), l(1,6): after the sort 0 = l(1,6), 1 = l(1,4); edge 0->1.  Seed 1+ 6+ 6- 1-;
    #  l(1,4): its 1- is the LAST cell (even offset): loop 1+ 4+ 4- 1- appended: bkp = 1+ 6+ 6- 1- | 1+ 4+ 4- 1-, two palindromes.
    #  Fold-backs: 6+ -> 6-, 4+ -> 4- (perfect), E = 1- -> 2+ (inv[1] = inv[2] = E).
    #  pos 0: palindrome, middle cell 1: perfect, nothing changes; p1 = -1 ends the loop (p1 > begin fails).
    #  pos 4: palindrome (cells 4..7), middle p1 = 5 (4+): perfect.  p1 = 3 = pos-1 is still inside the loop condition (:3475):
    #   cell 3 = 1-, '-': E.source 1 < E.target 2 -> cell 3 = target- = 2-, cell 4 = source+ = 1+ (:3489-3492).  p2 = 8 is one past
    #   the palindrome: the guard skips `*p2` (it would be the cell BEHIND the vector) and only cell 7 = complement(cell 4) = 1- is
    #   written (:3500).   bias = 1 + E = 2.
    "F4 palindrome rewrite: the cell in front of the palindrome (guard)": dict(n=6, juncs=[(6, '+', 6, '-'), (4, '+', 4, '-'), (1, '-', 2, '+')],
        elems=[("l", 1, 6, 1), ("l", 1, 4, 1)], bkp=[1, 6, -6, -2, 1, 4, -4, -1], bias=2),
    # ---------------------------------------------------------------- indelBFB (:3746-3837) and getIndelBias (:3699-3744) on P
    # D1.  SV 2+ -> 4+.  getIndelBias: group [2, 4], 2 < 4: deletion, CN of the segments strictly between +1 (:3733-3735): seg 3.
    #  indelBFB: group [2+, 4+], same strand, '+' and 2 < 4: deletion.  pos1 = first 2+ = 1, pos2 = first 4+ behind it = 3; 3 - 1 <= 3:
    #  erase (pos1, pos2) = cell 2 (3+).
    "D1 deletion": dict(n=6, juncs=TELO6 + [(2, '+', 4, '+')], elems=[("l", 1, 6, 1)], path=P,
        path_indel=[1, 2, 4, 5, 6, -6, -5, -4, -3, -2, -1], printed=True, cn_delta={3: +1}, bias=1),
    # D2.  SV 1+ -> 5+: pos1 = 0, pos2 = 4, 4 > 3 apart: skipped (:3791), the caption is still printed (:3835).  getIndelBias: 2, 3, 4 +1.
    "D2 deletion too far apart": dict(n=6, juncs=TELO6 + [(1, '+', 5, '+')], elems=[("l", 1, 6, 1)], path=P,
        path_indel=P, printed=True, cn_delta={2: +1, 3: +1, 4: +1}, bias=1),
    # D3.  Second attempt with the reversed + complemented group (:3785-3790).  Elements p(1,6), p(2,6): path Q (no 1-).
    #  SV 3- -> 1-: group [3-, 1-], '-' and 3 > 1: deletion.  pos1 = the 3- (cell 9), no 1- behind it -> group becomes [1+, 3+]:
    #  pos1 = 0, pos2 = 2: erase cell 1 (2+).   getIndelBias: ids -3, -1: -3 < -1 -> "deletion", j = -2: seg 2 +1.
    "D3 deletion, second attempt": dict(n=6, juncs=TELOQ + [(3, '-', 1, '-')], elems=[("p", 1, 6, 1), ("p", 2, 6, 1)], path=Q,
        path_indel=[1, 3, 4, 5, 6, -6, -5, -4, -3, -2], printed=True, cn_delta={2: +1}, bias=1),
    # U1.  SV 4+ -> 2+: '+' and 4 > 2: duplication (:3794).  pos1 = first 4+ = 3, pos2 = first 2+ in front of it = 1:
    #  insert(pos1+1, [pos2, pos1+1)) = 2+ 3+ 4+ again behind cell 3.  getIndelBias: group [4, 2], 4 > 2: CN of 2..4 -1 (:3736-3738).
    "U1 duplication": dict(n=6, juncs=TELO6 + [(4, '+', 2, '+')], elems=[("l", 1, 6, 1)], path=P,
        path_indel=[1, 2, 3, 4, 2, 3, 4, 5, 6, -6, -5, -4, -3, -2, -1], printed=True, cn_delta={2: -1, 3: -1, 4: -1}, bias=1),
    # U2.  SV 2- -> 4- on the reverse strand: '-' and 2 < 4: duplication.  pos1 = the 2- (cell 10), pos2 = the 4- in front (cell 8):
    #  -4 -3 -2 again behind cell 10.  getIndelBias: ids -2, -4: -2 > -4: j = -4..-2: segments 4, 3, 2 -1.
    "U2 duplication on the reverse strand": dict(n=6, juncs=TELO6 + [(2, '-', 4, '-')], elems=[("l", 1, 6, 1)], path=P,
        path_indel=[1, 2, 3, 4, 5, 6, -6, -5, -4, -3, -2, -4, -3, -2, -1], printed=True, cn_delta={2: -1, 3: -1, 4: -1}, bias=1),
    # U3.  Second attempt (:3798-3803).  Path Q; SV 1- -> 3-: '-' and 1 < 3: duplication; there is no 1- -> group becomes [3+, 1+]:
    #  pos1 = 2, pos2 = 0: 1+ 2+ 3+ again behind cell 2.  getIndelBias: ids -1, -3: j = -3..-1: segments 3, 2, 1 -1.
    "U3 duplication, second attempt": dict(n=6, juncs=TELOQ + [(1, '-', 3, '-')], elems=[("p", 1, 6, 1), ("p", 2, 6, 1)], path=Q,
        path_indel=[1, 2, 3, 1, 2, 3, 4, 5, 6, -6, -5, -4, -3, -2], printed=True, cn_delta={1: -1, 2: -1, 3: -1}, bias=1),
    # V1.  SV 3+ -> 6- (three apart: not a fold-back, LGM.cpp:3758): strands differ -> inversion (:3807).  pos1 = 2, pos2 = the 6-
    #  (cell 6); 4 <= 5: erase cells 3..5.  getIndelBias skips it (:3709): no CN edit.
    "V1 inversion": dict(n=6, juncs=TELO6 + [(3, '+', 6, '-')], elems=[("l", 1, 6, 1)], path=P,
        path_indel=[1, 2, 3, -6, -5, -4, -3, -2, -1], printed=True, cn_delta={}, bias=1),
    # V2.  SV 1+ -> 6-: pos1 = 0, pos2 = 6, 6 > 5: skipped (:3816).
    "V2 inversion too far apart": dict(n=6, juncs=TELO6 + [(1, '+', 6, '-')], elems=[("l", 1, 6, 1)], path=P,
        path_indel=P, printed=True, cn_delta={}, bias=1),
    # I1.  SVs 2+ -> 5+ and 5+ -> 3+ chain (:3768): group [2+, 5+, 3+], three vertices: insertion (:3820).  pos1 = first 2+ = 1,
    #  pos2 = first 3+ behind it = 2: nothing to erase, the inner vertex 5+ goes in between.  getIndelBias: [2, 5, 3]: inner seg 5 -1 (:3740-3742).
    "I1 insertion": dict(n=6, juncs=TELO6 + [(2, '+', 5, '+'), (5, '+', 3, '+')], elems=[("l", 1, 6, 1)], path=P,
        path_indel=[1, 2, 5, 3, 4, 5, 6, -6, -5, -4, -3, -2, -1], printed=True, cn_delta={5: -1}, bias=1),
    # I2.  The same chain with the second SV written from its other side, 3- -> 5-: indelBFB chains it through edge B = 5+ -> 3+
    #  (:3769); getIndelBias through `group.back() == -targetID` (:3726): same group, same edit.
    "I2 insertion through the complement edge": dict(n=6, juncs=TELO6 + [(2, '+', 5, '+'), (3, '-', 5, '-')], elems=[("l", 1, 6, 1)], path=P,
        path_indel=[1, 2, 5, 3, 4, 5, 6, -6, -5, -4, -3, -2, -1], printed=True, cn_delta={5: -1}, bias=1),
    # I3.  Chain grown at the FRONT (:3766): SVs in file order 5+ -> 3+, then 2+ -> 5+: group starts [5+, 3+], the second SV's
    #  target is its front -> [2+, 5+, 3+]: as I1.  getIndelBias: `targetID == group.front()` (:3723).
    "I3 insertion, chain grown at the front": dict(n=6, juncs=TELO6 + [(5, '+', 3, '+'), (2, '+', 5, '+')], elems=[("l", 1, 6, 1)], path=P,
        path_indel=[1, 2, 5, 3, 4, 5, 6, -6, -5, -4, -3, -2, -1], printed=True, cn_delta={5: -1}, bias=1),
}


def _expected(c):
    bkp = c.get("bkp")
    path = c.get("path") or expand(bkp)
    return bkp, path, c.get("path_indel", path), c.get("printed", False)


def _run(lib, oracle, workdir, name, tag):
    c = CASES[name]
    key = name.split()[0]
    lh = _lh(os.path.join(workdir, "hand_%s_%s.lh" % (tag, key)), c["n"], c["juncs"])
    sol = _sol(os.path.join(workdir, "hand_%s_%s.sol" % (tag, key)), c["n"], c["elems"])
    bkp, path, path_indel, printed = _expected(c)
    want_cn = [2.0 + c.get("cn_delta", {}).get(i, 0) for i in range(1, c["n"] + 1)]
    e = api.reconstruct_sample(lib, lh, [sol])
    o = oracle.run_bfb(lh, [sol])
    assert e["ok"] and o["ok"], (name, e.get("err"), o.get("err"))
    for who, r in (("engine", e["chr"][0]), ("oracle", o["chr"][0])):
        if bkp is not None:
            assert list(r["bkp"]) == bkp, (name, who, "bkp", list(r["bkp"]))
        assert list(r["path"]) == path, (name, who, "path", list(r["path"]))
        assert list(r["path_indel"]) == path_indel, (name, who, "path after indelBFB", list(r["path_indel"]))
        assert bool(r["indel_printed"]) == printed, (name, who)
        assert r["bias"] == c["bias"], (name, who, "bias", r["bias"])
    assert [float(x) for x in e["chr"][0]["seg_cn"][1:]] == want_cn, (name, "engine seg CN")
    assert [float(x) for x in o["chr"][0]["seg_cn"][:c["n"]]] == want_cn, (name, "oracle seg CN")
    if printed:
        assert "BFB path with insertion, deletion, or duplication:" in e["log"] and e["log"] == o["log"]


@pytest.mark.parametrize("name", sorted(CASES))
def test_hand_derived_stage_cases_host(hostsim_lib, oracle, workdir, name):
    _run(hostsim_lib, oracle, workdir, name, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_hand_derived_stage_cases_gpu(hip_lib, oracle, workdir, name):
    _run(hip_lib, oracle, workdir, name, "gpu")


def test_expand_helper():
    assert expand([1, 3, -3, -2]) == [1, 2, 3, -3, -2]


# ---- what the reference itself holds: README.md:122 (6-segment example), :146 (I2 insertion), :166 (C2 concatenation) ----
def _known(lib):
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ka = json.load(open(os.path.join(root, "tests", "golden", "known_answers.json")))
    for case in ("readme6", "readme_c2", "readme_i2"):
        k = ka[case]
        e = api.reconstruct_sample(lib, os.path.join(root, k["lh"]), [os.path.join(root, s) for s in k["sols"]])
        assert e["ok"], (case, e["err"])
        lines = [l for l in e["log"] if l and (l[0].isdigit() or l.startswith("BFB"))]
        assert lines == k["forward"], (case, lines)
        assert e["log"][-1] == k.get("reference_held_last_line", k["forward"][-1]), case


def test_product_prints_the_reference_held_lines_host(hostsim_lib):
    _known(hostsim_lib)


@pytest.mark.gpu
def test_product_prints_the_reference_held_lines_gpu(hip_lib):
    _known(hip_lib)


def _check_dag_hand_derived(lib, oracle, workdir, tag):
    """constructDAG (LGM.cpp:3276-3378) and allTopologicalOrders (:3380-3409) walked by hand on a chromosome of 12 segments, where
    the std::map's STRING order of the keys differs from their numeric order.

    Elements (all cn 1): p(1,12), l(1,10), l(1,2), l(3,10).  Keys in map order (:3279; byte-wise): "l:1,10" < "l:1,2" (after the
    common "l:1," comes '1' < '2') < "l:3,10" < "p:1,12"  ->  nodes 0 = l(1,10), 1 = l(1,2), 2 = l(3,10), 3 = p(1,12);
    node2pat = [-, -, -, (1,12)], node2loop = [(1,10), (1,2), (3,10), -].
    std::sort(node2loop, compareLoops) (:3303; 4 elements: libstdc++ runs a plain insertion sort; compareLoops is "longer first"
    and says false whenever a side is empty, :3267-3274), lengths 9, 1, 7, -:
      i=1 (1,2):   not before (1,10) (1 > 9 false); unguarded insert: (1,2) vs (1,10) false -> stays           [(1,10) (1,2) (3,10) -]
      i=2 (3,10):  7 > 9 false; unguarded: vs (1,2) 7 > 1 true -> (1,2) moves up; vs (1,10) false -> placed     [(1,10) (3,10) (1,2) -]
      i=3 empty:   false against everything -> stays
    so node2loop = [(1,10), (3,10), (1,2), -]: node 1 now stands for l(3,10) and node 2 for l(1,2).
    Edges.  p -> l (:3326-3336) from node 3 = p(1,12), length 11: node 0 l(1,10) same start, 11 > 9 -> 3->0; node 1 l(3,10) shares
    nothing; node 2 l(1,2) same start, 11 > 1 -> 3->2.  Loops in index order (:3339-3377): node 0 l(1,10): l -> p: node 3 is its
    parent -> skipped (:3343); l -> l: node 1 l(3,10) same end, 9 > 7 -> 0->1; node 2 l(1,2) same start, 9 > 1 -> 0->2.
    node 1 l(3,10): pattern (1,12) shares neither end; loops: (1,10) same end but 7 > 9 false; (1,2) shares nothing.  node 2 l(1,2):
    node 3 is a parent -> skipped; (1,10) same start but 1 > 9 false.
    adj = {0: [1,2], 1: [], 2: [], 3: [0,2]}; in-degrees 1, 1, 2, 0 -> orders, lowest free node first: 3 0 1 2, then 3 0 2 1.

    getBFB on them (fold-backs at 12 and 1 only).  FORWARD: both orders seed p(1,12) = 1+ 12+ and then meet l(1,10): neither 1- nor 10+
    is in the path -> break (:3610): both invalid, the last one too -> the orientation flips (:3691-3695).  REVERSED, order 3 0 1 2:
      seed p(1,12) reversed        12- 1-
      l(1,10): v1 = 1- at cell 1 (odd, the last cell: no nesting test) -> behind it 1+ 10+ 10- 1-        12- 1- 1+ 10+ 10- 1-
      l(3,10): no 3-; 10+ from the back: cell 3 (odd; 3 < L-2 = 4: id(cell 2) = 1 > id(cell 5) = 1 false) -> 10- 3- 3+ 10+ behind it
                                   12- 1- 1+ 10+ 10- 3- 3+ 10+ 10- 1-
      l(1,2): 1- from the back: cell 9 (odd, the last cell) -> 1+ 2+ 2- 1- at the end
                                   12- 1- 1+ 10+ 10- 3- 3+ 10+ 10- 1- 1+ 2+ 2- 1-
    All placed.  imperfectFBI: pos 0 (12- 1-): no 12+ behind -> plain, the perfect fold-back at 1 rewrites 1- as 1-.  pos 2: 1- found at
    cell 9, cell 8 = 10- = -cell 3: palindrome, mid 5; p1 = 5 (3-), 3 (10+): no fold-back there; p1 = 1 = pos-1 (1-): cells 1, 2 and the
    mirrored cell 9 get the values they hold.  pos 10: 1- at cell 13, cell 12 = 2- = -cell 11: palindrome, mid 11; p1 = 11 (2+): none;
    p1 = 9 = pos-1 (1-): cells 9, 10, 13 unchanged.  -> first valid order 0 of the REVERSED pass, three evaluations."""
    n = 12
    lh = _lh(os.path.join(workdir, "hand_%s_dag.lh" % tag), n, [(12, '+', 12, '-'), (1, '-', 1, '+')])
    sol = _sol(os.path.join(workdir, "hand_%s_dag.sol" % tag), n, [("p", 1, 12, 1), ("l", 1, 10, 1), ("l", 1, 2, 1), ("l", 3, 10, 1)])
    want_pat = [[], [], [], [1, 12, 1]]
    want_loop = [[1, 10, 1], [3, 10, 1], [1, 2, 1], []]
    want_adj = [[1, 2], [], [], [0, 2]]
    want_orders = [[3, 0, 1, 2], [3, 0, 2, 1]]
    oc = oracle.run_bfb(lh, [sol], keep_orders=True)["chr"][0]
    assert oc["node2pat"] == want_pat and oc["node2loop"] == want_loop and [sorted(a) for a in oc["adj"]] == want_adj
    assert oc["num_orders"] == 2 and oc["orders"] == want_orders
    g = api.Graph(lib, lh)
    b = api.Batch(lib)
    b.add_chromosome_sol(g, 0, sol)
    b.upload(); b.run(0); b.download()
    r = b.unit_result(0)
    assert r["n_nodes"] == 4 and r["num_orders"] == 2
    pat, loop, succ = b.unit_dag(0, 4)
    assert [[] if p[0] == 0 else p for p in pat.tolist()] == want_pat
    assert [[] if p[0] == 0 else p for p in loop.tolist()] == want_loop
    assert [int(x) for x in succ] == [sum(1 << j for j in a) for a in want_adj]
    assert b.unit_orders(0, 0, 2, 4).tolist() == want_orders
    want_bkp = [-12, -1, 1, 10, -10, -3, 3, 10, -10, -1, 1, 2, -2, -1]
    assert (oc["first_valid"], oc["first_forward"], oc["evaluated"], list(oc["bkp"])) == (0, 0, 3, want_bkp) and list(oc["path"]) == expand(want_bkp)
    assert (r["status"], r["first_valid"], r["first_forward"], r["evaluated"]) == (0, 0, 0, 3)
    assert b.unit_bkp(0).tolist() == want_bkp and b.unit_path(0, 0).tolist() == expand(want_bkp)
    # --all (LGM.cpp:3519-3696 with printAll): every order of a pass is evaluated and every valid one printed; the reversed pass runs
    # because the LAST forward order was invalid.  Reversed order 3 0 2 1: 12- 1- | l(1,10) as above | l(1,2): 1- from the back is cell 5,
    # the last cell -> 1+ 2+ 2- 1- at the end | l(3,10): 10+ from the back is cell 3 (3 < L-2 = 8: 1 > 1 false) -> 10- 3- 3+ 10+ behind it:
    # the SAME cells as order 3 0 1 2.  So: four evaluations, the valid ones are evaluations 2 and 3, the path printed twice.
    line = "12-11-10-9-8-7-6-5-4-3-2-1-|1+2+3+4+5+6+7+8+9+10+|10-9-8-7-6-5-4-3-|3+4+5+6+7+8+9+10+|10-9-8-7-6-5-4-3-2-1-|1+2+|2-1-"
    oa = oracle.run_bfb(lh, [sol], all_=True)
    ea = api.reconstruct_sample(lib, lh, [sol], all_=True)
    for who, res in (("oracle", oa), ("engine", ea)):
        c = res["chr"][0]
        assert [list(x) for x in c["all_paths"]] == [expand(want_bkp)] * 2 and c["evaluated"] == 4, who
        assert [l for l in res["log"] if l and l[0].isdigit()] == [line, line], who
    b.close(); g.close()


def test_dag_hand_derived_host(hostsim_lib, oracle, workdir):
    _check_dag_hand_derived(hostsim_lib, oracle, workdir, "cpu")


@pytest.mark.gpu
def test_dag_hand_derived_gpu(hip_lib, oracle, workdir):
    _check_dag_hand_derived(hip_lib, oracle, workdir, "gpu")


def _check_junc_cn_hand_derived(lib, oracle, workdir, tag):
    """getJuncCN (LGM.cpp:3989-4050) and the bias (localhap.cpp:141-146) walked by hand.  Six segments, junctions in file order
    (index = position in the graph's junction list; none is a duplicate or the complement of an earlier one, Graph.cpp:592-597):

      0  1+ -> 2+  cn 2.0   same strand, source+1 == target            juncCN[1][0] += 2.0                         (:4003-4005)
      1  2+ -> 3+  cn 0.7   0.5 < cn < 1 counts as 1 (:4000-4001)      juncCN[2][0] += 1.0
      2  4- -> 3-  cn 1.5   same strand, source-1 == target            juncCN[3][0] += 1.5  (slot of the TARGET)   (:4007-4009)
      3  4+ -> 5+  cn 1.0                                              juncCN[4][0] += 1.0
      4  5+ -> 6+  cn 1.0                                              juncCN[5][0] += 1.0
      5  6+ -> 6-  cn 3.0   opposite strands, |6-6| <= 2: fold-back; inversions[6] free -> junction 5, juncCN[6][1] += 3.0   (:4012-4016)
      6  5+ -> 6-  cn 0.6   fold-back (|5-6| = 1); inversions[5] free -> junction 6, juncCN[5][1] += 1.0 (rounded)
      7  6+ -> 4-  cn 2.0   fold-back (|6-4| = 2); inversions[6] taken, inversions[4] free -> junction 7, juncCN[4][1] += 2.0   (:4027-4029)
      8  5- -> 6+  cn 1.0   fold-back; inversions[5] and [6] both taken: nothing recorded, its CN counted nowhere
      9  2- -> 1+  cn 1.0   fold-back; inversions[2] free -> junction 9, juncCN[2][1] += 1.0
    second pass (:4042-4049): every end of a fold-back still without an entry: only segment 1 (target of junction 9) -> junction 9.
    bias = 1 + sum over segments with juncCN[i][1] > 0 whose recorded fold-back joins two DIFFERENT segments of int(juncCN[i][1]) % 2:
      i=2 (junction 9: 2 vs 1) 1 % 2 = 1;  i=4 (junction 7: 6 vs 4) 2 % 2 = 0;  i=5 (junction 6: 5 vs 6) 1;  i=6 (junction 5: 6 vs 6) skipped  ->  3.
    No same-strand junction skips a segment: getIndelBias leaves the segment CNs alone."""
    lh = os.path.join(workdir, "hand_%s_jcn.lh" % tag)
    J = [(1, '+', 2, '+', 2.0), (2, '+', 3, '+', 0.7), (4, '-', 3, '-', 1.5), (4, '+', 5, '+', 1.0), (5, '+', 6, '+', 1.0),
         (6, '+', 6, '-', 3.0), (5, '+', 6, '-', 0.6), (6, '+', 4, '-', 2.0), (5, '-', 6, '+', 1.0), (2, '-', 1, '+', 1.0)]
    with open(lh, "w") as f:
        f.write("SAMPLE_NAME hand\nAVG_CHR_SEG_DP 30\nAVG_WHOLE_HOST_DP 30\nAVG_JUNC_DP 30\nPURITY 1\nAVG_TUMOR_PLOIDY 2\nPLOIDY 2m1\nVIRUS_START 7\nSOURCE 1\nSINK 6\n")
        for i in range(1, 7):
            f.write("SEG H:%d:chr1:%d:%d 60.0 4.0\n" % (i, i * 1000 + 1, i * 1000 + 1000))
        for (a, ad, b, bd, cn) in J:
            f.write("JUNC H:%d:%s H:%d:%s %.1f %.1f U B\n" % (a, ad, b, bd, 30 * cn, cn))
    sol = _sol(os.path.join(workdir, "hand_%s_jcn.sol" % tag), 6, [("l", 1, 6, 1)])
    want_jcn = [[2.0, 0.0], [1.0, 1.0], [1.5, 0.0], [1.0, 2.0], [1.0, 1.0], [0.0, 3.0]]        # segments 1..6: [normal, fold-back]
    want_inv = {1: 9, 2: 9, 4: 7, 5: 6, 6: 5}
    oc = oracle.run_bfb(lh, [sol])["chr"][0]
    import numpy as np
    assert np.array(oc["junc_cn"], dtype=float).reshape(-1, 2)[1:7].tolist() == want_jcn      # (rows by segment id, row 0 unused)
    assert dict(zip(oc["inv_seg"], oc["inv_junc"])) == want_inv and oc["bias"] == 3
    g = api.Graph(lib, lh)
    assert g.n_junc == 10
    b = api.Batch(lib)
    b.add_chromosome_sol(g, 0, sol)
    b.upload(); b.run(0); b.download()
    prep = b.unit_prepare(0, 6)
    assert prep["junc_cn"][1:].tolist() == want_jcn
    assert {i: int(j) for i, j in enumerate(prep["inv_junc"]) if i >= 1 and j >= 0} == want_inv
    assert b.unit_result(0)["bias"] == 3
    assert prep["seg_cn"][1:].tolist() == [4.0] * 6
    b.close(); g.close()


def test_junc_cn_hand_derived_host(hostsim_lib, oracle, workdir):
    _check_junc_cn_hand_derived(hostsim_lib, oracle, workdir, "cpu")


@pytest.mark.gpu
def test_junc_cn_hand_derived_gpu(hip_lib, oracle, workdir):
    _check_junc_cn_hand_derived(hip_lib, oracle, workdir, "gpu")


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check_readme_out_junctions(lib, oracle):
    """Output-junction synthesis (localhap.cpp:267-289) on the README example, whose path the reference holds (README.md:122):
      1+2+3+4+5+6+|6-5-4-3-2-|2+3+4+|4-3-|3+4+|4-3-2-|2+3+4+5+6+|6-5-4-3-2-1-
    A step u -> v is a junction unless the ids are one apart on the same strand; a repeat of the same edge (or its complement)
    raises the count of the first entry, which starts at 1.  Steps at the seven '|': 6+ -> 6- (new), 2- -> 2+ (new), 4+ -> 4- (new),
    3- -> 3+ (new), 4+ -> 4- (2nd), 2- -> 2+ (2nd), 6+ -> 6- (2nd)  ->  in first-appearance order (6+,6-) x2, (2-,2+) x2, (4+,4-) x2, (3-,3+) x1."""
    lh, sol = os.path.join(ROOT, "tests", "data", "readme6.lh"), os.path.join(ROOT, "tests", "data", "readme6.sol")
    want = [(6, -6, 2), (-2, 2, 2), (4, -4, 2), (-3, 3, 1)]
    o = oracle.run_bfb(lh, [sol])
    assert [tuple(x) for x in o["out_juncs"]] == want
    g = api.Graph(lib, lh)
    b = api.Batch(lib)
    b.add_chromosome_sol(g, 0, sol)
    b.upload(); b.run(0); b.download()
    assert b.unit_path(0, 1).tolist() == [1, 2, 3, 4, 5, 6, -6, -5, -4, -3, -2, 2, 3, 4, -4, -3, 3, 4, -4, -3, -2, 2, 3, 4, 5, 6, -6, -5, -4, -3, -2, -1]
    assert b.unit_out_juncs(0) == want
    b.close(); g.close()


def test_readme_out_junctions_host(hostsim_lib, oracle):
    _check_readme_out_junctions(hostsim_lib, oracle)


@pytest.mark.gpu
def test_readme_out_junctions_gpu(hip_lib, oracle):
    _check_readme_out_junctions(hip_lib, oracle)


def _check_readme_reversed_walk(lib, oracle):
    """getBFB with --reversed (LGM.cpp:3519-3658, seed :3524-3568, loop placement :3586-3644) walked by hand on the README example.
    The .sol selects columns 26, 29, 31, 33 = loops l(1,6), l(2,4), l(2,6), l(3,4) (numPat 21; rank(a,b) of localhap.cpp:117-133).
    constructDAG: keys "l:1,6" < "l:2,4" < "l:2,6" < "l:3,4"; lengths 5, 2, 4, 1 -> after the sort node2loop = (1,6) (2,6) (2,4) (3,4);
    edges (1,6)->(2,6) (same end), (2,6)->(2,4) (same start), (2,4)->(3,4) (same end): a chain, ONE order 0 1 2 3.
    Reversed orientation: the seed of a loop (s,e) is  e- s- s+ e+  (:3548-3566), an insert behind a slot holding s- is
    s+ e+ e- s-, behind a slot holding e+ it is e- s- s+ e+ (:3605-3643); the slot is the LAST odd cell holding s- or e+ that
    passes the nesting test |cell[q-1]| vs |cell[q+2]| (only for q < L-2):
      seed (1,6)        6- 1- 1+ 6+
      (2,6): odd cells 1- 6+ : cell 3 = 6+ = e+ (q = 3 = L-1: no test)            -> behind it  6- 2- 2+ 6+
                        6- 1- 1+ 6+ 6- 2- 2+ 6+
      (2,4): odd cells 1- 6+ 2- 6+ : cell 5 = 2- = s- (q = 5 < 6: |6-| vs |6+| equal: kept)  -> behind it  2+ 4+ 4- 2-
                        6- 1- 1+ 6+ 6- 2- 2+ 4+ 4- 2- 2+ 6+
      (3,4): odd cells .. cell 7 = 4+ = e+ (q = 7 < 10: |2+| vs |2-| equal: kept)  -> behind it  4- 3- 3+ 4+
                        6- 1- 1+ 6+ 6- 2- 2+ 4+ 4- 3- 3+ 4+ 4- 2- 2+ 6+
    (the in-place fix-ups :3626-3628 / :3639-3641 rewrite cells with the values they hold).  All four placed; every fold-back of
    the file is perfect: imperfectFBI changes nothing.  Pairs -> 6-..1- | 1+..6+ | 6-..2- | 2+3+4+ | 4-3- | 3+4+ | 4-3-2- | 2+..6+
    (the same line the survey's run of the reference printed for --reversed, tests/golden/known_answers.json)."""
    lh, sol = os.path.join(ROOT, "tests", "data", "readme6.lh"), os.path.join(ROOT, "tests", "data", "readme6.sol")
    bkp = [-6, -1, 1, 6, -6, -2, 2, 4, -4, -3, 3, 4, -4, -2, 2, 6]
    path = expand(bkp)
    o = oracle.run_bfb(lh, [sol], reversed_=True, keep_orders=True)["chr"][0]
    assert o["node2loop"] == [[1, 6, 1], [2, 6, 1], [2, 4, 1], [3, 4, 1]] and o["orders"] == [[0, 1, 2, 3]]
    assert list(o["bkp"]) == bkp and list(o["path"]) == path and (o["first_valid"], o["first_forward"]) == (0, 0)
    g = api.Graph(lib, lh)
    b = api.Batch(lib)
    b.add_chromosome_sol(g, 0, sol)
    b.upload(); b.run(api.FLAG_REVERSED); b.download()
    r = b.unit_result(0)
    assert (r["status"], r["first_valid"], r["first_forward"], r["evaluated"]) == (0, 0, 0, 1)
    assert b.unit_bkp(0).tolist() == bkp and b.unit_path(0, 0).tolist() == path and b.unit_path(0, 1).tolist() == path
    b.close(); g.close()


def test_readme_reversed_walk_host(hostsim_lib, oracle):
    _check_readme_reversed_walk(hostsim_lib, oracle)


@pytest.mark.gpu
def test_readme_reversed_walk_gpu(hip_lib, oracle):
    _check_readme_reversed_walk(hip_lib, oracle)
