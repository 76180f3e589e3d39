"""The oracle is pinned before it is trusted: reference known answers + the real reference graph model."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
KA = json.load(open(os.path.join(GOLD, "known_answers.json")))


def _paths(log):
    return [l for l in log if l and (l[0].isdigit() or l.startswith("BFB"))]


@pytest.mark.parametrize("case", ["readme6", "trx_c2", "readme_c2", "readme_i2"])
def test_known_answer_forward(oracle, case):
    k = KA[case]
    r = oracle.run_bfb(os.path.join(ROOT, k["lh"]), [os.path.join(ROOT, s) for s in k["sols"]])
    assert r["ok"], r["err"]
    assert _paths(r["log"]) == k["forward"]
    if "reference_held_last_line" in k:   # README.md:146 / :166 -- what the reference itself holds for these two
        assert r["log"][-1] == k["reference_held_last_line"]


def test_known_answer_reversed(oracle):
    k = KA["readme6"]
    r = oracle.run_bfb(os.path.join(ROOT, k["lh"]), [os.path.join(ROOT, s) for s in k["sols"]], reversed_=True)
    assert _paths(r["log"]) == k["reversed"]


def test_readme_time_csv_fields(oracle):
    # time.csv row of the README example recorded by the survey: readme6,6,4,0,32,32,8,<sec> (localhap.cpp:386-388)
    k = KA["readme6"]
    r = oracle.run_bfb(os.path.join(ROOT, k["lh"]), [os.path.join(ROOT, s) for s in k["sols"]])
    t = k["time_csv"]
    assert (r["num_inv"], r["cn_sum"], r["path_len"], r["max_cn"]) == (t["n_inv"], t["cn_sum"], t["path_len"], t["max_cn"])


@pytest.mark.parametrize("name,lh", [("readme6", "tests/data/readme6.lh"), ("trx_c2", "tests/data/trx_c2.lh"),
                                     ("quirks", "tests/data/quirks.lh"), ("syn24", "tests/golden/syn24.lh")])
def test_reader_against_reference_fixture(oracle, name, lh):
    """graph_*.json were produced by the REAL reference reader (tests/golden/make_golden.py)."""
    gold = json.load(open(os.path.join(GOLD, "graph_%s.json" % name)))
    assert oracle.graph_dump(os.path.join(ROOT, lh)) == gold


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference sources only exist in the build container")
def test_reader_against_live_reference(oracle):
    for lh in ["tests/data/readme6.lh", "tests/data/quirks.lh", "tests/golden/syn24.lh"]:
        p = os.path.join(ROOT, lh)
        ref = oracle.ref_graph_dump(p)
        assert ref is not None and ref["ok"]
        assert oracle.graph_dump(p) == ref


def test_all_mode_prints_every_valid_order(oracle, workdir):
    from ambigram_amd import synth
    s = synth.make_sample(32, 64, "wide", 5, seed=3)
    lh, sols = s.write(workdir, "all5")
    r = oracle.run_bfb(lh, sols, all_=True)
    c = r["chr"][0]
    assert c["num_orders"] == 6 and len(c["all_paths"]) >= 1
    assert c["all_paths"][0] == c["path"]


def test_synthetic_cases_reach_the_insertion_branch(oracle, workdir):
    """the multi-chromosome parity cases really run translocationBFB's insertion branch (LGM.cpp:4120-4190), not only the
    concatenation one"""
    import cases
    from collections import Counter
    seen = Counter()
    for name, lh, sols in cases.synthetic_cases(workdir):
        if name.startswith("trxins"):
            seen.update(oracle.run_bfb(lh, sols)["trx_trace"])
    assert seen["insert"] >= 10 and seen["concat"] >= 2, seen
