"""TRX-BFB, `PROP I1` / `PROP C1`: insertBeforeBFB, concatBeforeBFB (LocalGenomicMap.cpp:4195-4395) and
virusBFB (:3839-3939).

Pins:
  * README.md:134 (I1) and README.md:154-157 (C1, two captioned stages) -- the OUTPUT lines are reference-held, the inputs are
    reconstructed from them (tests/data/readme_i1.lh, readme_c1.lh; the reference's own files live in another repository);
  * one small case per mode derived BY HAND from the reference source, the walk written above each case.
Both the product (host code lh_graph.cpp behind the C ABI; batch through the host simulation on the CPU and through
libambigram_hip.so with -m gpu) and the oracle are compared with them.

`new Graph(mSegs, mJuncs, mSources, mSinks)` (LGM.cpp:4293 / :4393) assigns through uninitialised pointers (Graph.cpp:25-34); its
evident meaning -- a graph made of copies of the four vectors -- is what both transcriptions implement.
The order of the "Seg conversion" lines is the iteration order of a std::unordered_map<int,int>: libstdc++ links a node whose bucket
is empty in FRONT of the element list, so for the small distinct ids of these cases (no two in one of the 13 buckets, no rehash)
the lines come out in REVERSE insertion order."""
import json
import os

import pytest

from ambigram_amd import api
import parity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KNOWN = json.load(open(os.path.join(ROOT, "tests", "golden", "known_answers.json")))


def _lh(path, chroms, juncs, prop):
    """chroms: [(name, [cn of every segment])] in file order; juncs: [(a, adir, b, bdir, cn)]"""
    L = ["SAMPLE_NAME trx", "AVG_CHR_SEG_DP 30", "AVG_WHOLE_HOST_DP 30", "AVG_JUNC_DP 30", "PURITY 1", "AVG_TUMOR_PLOIDY 2", "PLOIDY 2m1", "VIRUS_START 99"]
    src, snk, at = [], [], 1
    for _, cns in chroms:
        src.append(at); at += len(cns); snk.append(at - 1)
    L += ["SOURCE " + ",".join(map(str, src)), "SINK " + ",".join(map(str, snk))]
    i = 1
    for name, cns in chroms:
        for cn in cns:
            L.append("SEG H:%d:%s:%d:%d %.1f %.1f" % (i, name, i * 1000 + 1, i * 1000 + 1000, 30.0 * cn, cn)); i += 1
    for (a, ad, b, bd, cn) in juncs:
        L.append("JUNC H:%d:%s H:%d:%s %.1f %.1f U B" % (a, ad, b, bd, 30.0 * cn, cn))
    L.append(prop)
    with open(path, "w") as f:
        f.write("\n".join(L) + "\n")
    return path


def _sol(path, n, elems):
    """planted .sol over a chromosome of n segments starting at id 1: elems = [('p'|'l', a, b, cn)]"""
    num_pat = n * (n + 1) // 2
    with open(path, "w") as f:
        f.write("Optimal - objective value 0.00000000\n")
        for (k, a, b, cn) in elems:
            col = (a - 1) * n - (a - 1) * (a - 2) // 2 + (b - a) + (num_pat if k == "l" else 0)
            f.write("%7d x%-7d %d 0\n" % (col, col, cn))
    return path


# ---- hand-derived cases --------------------------------------------------------------------------------------------
# H-I1.  chrA = 1 2 3, virus V = 4; junctions in file order: 1+ -> 2+, 2+ -> 4+, 4+ -> 3+, 3+ -> 3-.  PROP I1:chrA:V:chrA M:chrA.
#  insertBeforeBFB: insChr = [chrA, V, chrA].  i = 1: the first junction between chrA and V is 2+ -> 4+ (chr1 = chrA = insChr[0]:
#  no swap): insertionIDs = [2, 4] (:4210-4224).  i = 2 (V, chrA): 4+ -> 3+, id1 = 4 = back(): nothing in between (:4217), push 4, 3
#  -> [2, 4, 4, 3]; unique -> [2, 4, 3] (:4228); front 2 < back 3: no reversal; sID = 2, eID = 3, insertionIDs = [4] (:4230-4232);
#  deletedChrIDs = [chr of V].  mSegs (:4237-4253): i = 1 < sID -> 1 -> new 1; i = 2 in [sID, eID]: 2 -> new 2, nothing between 2 and 3 to
#  delete, inserted 4 -> new 3 (on chrA), eID 3 -> new 4, i = eID; i = 4: its chromosome is deleted -> skipped.  Inserts into
#  segConversion: 1, 2, 4, 3.  Junctions (:4263-4283): 1+ -> 2+: "1-2 1-2"; 2+ -> 4+ touches the inserted 4: ids (2,3), dirs ++:
#  "2-4 2-3"; 4+ -> 3+: ids conv[4] = 3, conv[3] = 4, touches: "4-3 3-4"; 3+ -> 3-: edge A runs from 3+ to 3-, two vertices, kept (:4264):
#  ids (4,4): "3-3 4-4".  "Seg conversion:" then the map in reverse insertion order: 3-4, 4-3, 2-2, 1-1.  writeGraph: "write seg".
#  The rebuilt chromosome 1..4 with the planted loop l(1,4) and the fold-back 4+ -> 4-: getBFB prints 1+2+3+4+|4-3-2-1-.
#  virusBFB (:3839-3901) on [1,2,3,4,-4,-3,-2,-1], originalSegs new -> file = 1,2,4,3: first vertex: file segments 1 and 2 on one
#  chromosome -> 1+ (:3869-3874).  2: same chromosome as 1+, no turn -> 2+.  new 3 = file 4 (V), previous 2+ on chrA: the edges that
#  LEAVE 2+ in registration order: 2+ -> 4+ leads to segment 4: 4+ (:3881-3888).  new 4 = file 3 (chrA) after 4+ (V): 4+ -> 3+: 3+.
#  -4 = file 3, same chromosome as 3+, a turn, previous '+': 3- (:3890-3895).  -3 = file 4 (V) after 3- (chrA): edge B of 4+ -> 3+ is
#  3- -> 4-: 4-.  -2 = file 2 after 4- (V): edge B of 2+ -> 4+ is 4- -> 2-: 2-.  -1 = file 1, same chromosome, previous '-': 1-.
#  printBFB with the chromosomes of the FILE: 1+2+||4+||3+|3-||4-||2-1-.  No unused junction: no second stage.
H_I1 = dict(chroms=[("chrA", [2, 2, 2]), ("V", [2])],
            juncs=[(1, '+', 2, '+', 1), (2, '+', 4, '+', 1), (4, '+', 3, '+', 1), (3, '+', 3, '-', 1)], prop="PROP I1:chrA:V:chrA M:chrA",
            n=4, elems=[("l", 1, 4, 1)],
            log=["1-2 1-2", "2-4 2-3", "4-3 3-4", "3-3 4-4", "Seg conversion:", "3-4", "4-3", "2-2", "1-1", "write seg",
                 "Declare done", "ILP formula done", "Variable constrains done", "1+2+3+4+|4-3-2-1-",
                 "TRX-BFB mode: BFB path in the first stage:", "1+2+||4+||3+|3-||4-||2-1-"],
            original_of=[0, 1, 2, 4, 3], paths=[[1, 2, 4, 3, -3, -4, -2, -1]],
            out_juncs=[(2, 4, 2), (3, -3, 1)])
# (output junctions, localhap.cpp:267-289, on the restored path: 2+ -> 4+ new; 4+ -> 3+ is NOT one -- the test is on the ids alone,
#  |4 - 3| == 1 on one strand, whatever the chromosomes; 3+ -> 3- new; 3- -> 4- again ids one apart; 4- -> 2- the complement of 2+ -> 4+:
#  count 2; 1+ -> 2+ and 2- -> 1- are adjacencies.)

# H-C1.  chrA = 1 2 3, V = 4 5; junctions: 1+ -> 2+, 2+ -> 3+, 2+ -> 4+, 4+ -> 5+, 5+ -> 5-.  PROP C1:chrA:V.
#  concatBeforeBFB: the first junction between chrA and V is 2+ -> 4+: sID = 2 '+', eID = 4 '+': "Concat segs: 2+ 4+" (:4305-4324).
#  chrID1 = chrA, '+': segments source..sID = 1, 2 -> new 1, 2; 3 -> 0 (:4327-4333).  chrID2 = V, '+': eID..sink = 4, 5 -> new 3, 4 (all on
#  chrA's id); nothing in front of 4 (:4342-4348).  Inserts: 1, 2, 3, 4, 5.  Junctions (:4365-4383), every one printed first:
#  "1+ - 2+ 1-2"; "2+ - 3+ 2-0": 3 has no place -> unused; "2+ - 4+ 2-3": the joint -> ids (2,3), ++; "4+ - 5+ 3-4"; "5+ - 5- 4-4".
#  "Seg conversion:" in reverse insertion order: 5-4, 4-3, 3-0, 2-2, 1-1; "write seg".
#  l(1,4) on the rebuilt chromosome with the fold-back 4+ -> 4-: 1+2+3+4+|4-3-2-1-.
#  virusBFB: 1+ 2+ as above; new 3 = file 4 (V) after 2+ (chrA): the edges that leave 2+: 2+ -> 3+ (segment 3, not it), 2+ -> 4+: 4+.
#  new 4 = file 5, same chromosome as 4+, no turn: 5+; then the turn: 5-; -3 = file 4 after 5- (same chromosome, previous '-'): 4-;
#  -2 = file 2 (chrA) after 4- (V): edge B of 2+ -> 4+ is 4- -> 2-: 2-; then 1-.  First stage: 1+2+||4+5+|5-4-||2-1-.
#  Second stage (:3904-3938), unused = [2+ -> 3+]: edge A's source 2+ is in the path, LAST at index 1, i.e. 6 steps from the back
#  (pos1 - rbegin); pos2 = first occurrence of edge B's target = 2-: index 6; 6 < 6 is false -> the tail behind that 2+ goes
#  (:3917-3918) and edge A's target 3+ is appended: 1+2+3+.
H_C1 = dict(chroms=[("chrA", [2, 2, 1]), ("V", [2, 2])],
            juncs=[(1, '+', 2, '+', 1), (2, '+', 3, '+', 1), (2, '+', 4, '+', 1), (4, '+', 5, '+', 1), (5, '+', 5, '-', 1)], prop="PROP C1:chrA:V",
            n=4, elems=[("l", 1, 4, 1)],
            log=["Concat segs: 2+ 4+", "1+ - 2+ 1-2", "2+ - 3+ 2-0", "2+ - 4+ 2-3", "4+ - 5+ 3-4", "5+ - 5- 4-4", "Seg conversion:", "5-4", "4-3", "3-0",
                 "2-2", "1-1", "write seg", "Declare done", "ILP formula done", "Variable constrains done", "1+2+3+4+|4-3-2-1-",
                 "TRX-BFB mode: BFB path in the first stage:", "1+2+||4+5+|5-4-||2-1-", "TRX-BFB mode: BFB path in the second stage:", "1+2+3+"],
            original_of=[0, 1, 2, 4, 5], paths=[[1, 2, 3]], out_juncs=[])
HAND = {"H-I1": H_I1, "H-C1": H_C1}


def _write_case(workdir, name, case):
    lh = _lh(os.path.join(workdir, name + ".lh"), case["chroms"], case["juncs"], case["prop"])
    sol = _sol(os.path.join(workdir, name + ".sol"), case["n"], case["elems"])
    return lh, [sol]


def _tail(log):
    """the lines behind the loader's own ("bfb", "Reading graph...")"""
    return [l for l in log if l not in ("bfb", "Reading graph...")]


def _check_hand(lib, oracle, workdir, name):
    case = HAND[name]
    lh, sols = _write_case(workdir, name, case)
    o = oracle.run_bfb(lh, sols)
    assert o["ok"], o["err"]
    assert _tail(o["log"]) == case["log"], (name, "oracle", o["log"])
    assert o["trx_before"] and o["original_of"] == case["original_of"]
    assert o["paths"] == case["paths"]
    assert [tuple(j) for j in o["out_juncs"]] == case["out_juncs"]
    e = api.reconstruct_sample(lib, lh, sols)
    assert e["ok"], e["err"]
    assert _tail(e["log"]) == case["log"], (name, "engine", e["log"])
    assert e["trx_before"] and e["paths"] == case["paths"]
    assert e["out_juncs"] == case["out_juncs"]
    g = api.Graph(lib, lh)
    assert g.trx_before().tolist() == case["original_of"]
    g.close()
    assert parity.compare(lib, oracle, lh, sols) == []


def _check_readme(lib, oracle, key):
    d = KNOWN[key]
    lh, sols = os.path.join(ROOT, d["lh"]), [os.path.join(ROOT, s) for s in d["sols"]]
    o = oracle.run_bfb(lh, sols)
    assert o["ok"], o["err"]
    assert o["log"][-len(d["forward_tail"]):] == d["forward_tail"]
    for line in d["reference_held_lines"]:
        assert line in o["log"], (key, line)
    e = api.reconstruct_sample(lib, lh, sols)
    assert e["ok"], e["err"]
    assert e["log"] == o["log"]
    assert parity.compare(lib, oracle, lh, sols) == []


@pytest.mark.parametrize("key", ["readme_i1", "readme_c1"])
def test_readme_trx_before_hostsim(hostsim_lib, oracle, key):
    _check_readme(hostsim_lib, oracle, key)


@pytest.mark.parametrize("name", sorted(HAND))
def test_hand_derived_trx_before_hostsim(hostsim_lib, oracle, workdir, name):
    _check_hand(hostsim_lib, oracle, workdir, name)


@pytest.mark.gpu
@pytest.mark.parametrize("key", ["readme_i1", "readme_c1"])
def test_readme_trx_before_gpu(hip_lib, oracle, key):
    _check_readme(hip_lib, oracle, key)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(HAND))
def test_hand_derived_trx_before_gpu(hip_lib, oracle, workdir, name):
    _check_hand(hip_lib, oracle, workdir, name)


def test_trx_before_refusals(hostsim_lib, oracle, workdir):
    """Where the reference reads what nothing has set, both sides say so instead of printing something."""
    # no junction between the chromosomes of the I1 list: insertionIDs stays empty and front() / back() are read (LGM.cpp:4229-4230)
    lh = _lh(os.path.join(workdir, "no_junc_i1.lh"), [("chrA", [2, 2]), ("V", [2])], [(1, '+', 2, '+', 1)], "PROP I1:chrA:V:chrA M:chrA")
    o = oracle.run_bfb(lh, [])
    assert not o["ok"] and "insertBeforeBFB" in o["err"]
    with pytest.raises(api.AmbiError) as ei:
        api.Graph(hostsim_lib, lh)
    assert ei.value.code == -9   # AMBI_ERR_UNSUPPORTED
    # ... of the C1 list: sID / eID / sDir / eDir are read without ever being set (:4304-4324)
    lh = _lh(os.path.join(workdir, "no_junc_c1.lh"), [("chrA", [2, 2]), ("V", [2])], [(1, '+', 2, '+', 1)], "PROP C1:chrA:V")
    assert not oracle.run_bfb(lh, [])["ok"]
    with pytest.raises(api.AmbiError):
        api.Graph(hostsim_lib, lh)
    # a .juncs file with these modes: readComponents reads the rebuilt graph's mean coverage, which nothing sets (Graph.cpp:25-34, LGM.cpp:5133)
    lh, sols = _write_case(workdir, "juncs_with_i1", H_I1)
    jf = os.path.join(workdir, "x.juncs")
    open(jf, "w").write("1+ 2+\n")
    assert not oracle.run_bfb(lh, sols, juncs=jf)["ok"]
    g = api.Graph(hostsim_lib, lh)
    with pytest.raises(api.AmbiError):
        g.read_juncs(jf)
    g.close()
