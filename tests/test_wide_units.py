"""Units with 64..255 DAG nodes ("wide units", csrc/ambi_wide.hpp; up to 127 in round 3).  The reference has no bound on the number of selected
patterns / loops of a chromosome (constructDAG, LocalGenomicMap.cpp:3276-3301); rounds 1-2 refused more than 63.  Wide units
take the plain four-word form of the DAG / lattice stages and then the ordinary machinery (order table, parallel search for
the first valid order, finish stages, --all).  Everything against the oracle, on the host simulation and on the GPU."""
import numpy as np
import pytest

import parity
from ambigram_amd import api, synth


def _check_unit(lib, oracle, lh, sols, rev=False, orders=True):
    """one sample, one chromosome: the engine's results against the oracle's, refusals included"""
    o = oracle.run_bfb(lh, sols, reversed_=rev, keep_orders=orders)
    assert o["ok"], o["err"]
    oc = o["chr"][0]
    g = api.Graph(lib, lh)
    b = api.Batch(lib)
    b.add_chromosome_sol(g, 0, sols[0])
    b.upload(); b.run(api.FLAG_REVERSED if rev else 0); b.download()
    r = b.unit_result(0)
    assert r["n_nodes"] == len(oc["node2pat"]) and r["num_orders"] == oc["num_orders"]
    K = r["n_nodes"]
    if oc["ub_valid"]:      # the reference dereferences a null vector / reads out of bounds on this input: refused, at the same evaluation
        assert r["status"] == -12 and r["evaluated"] == oc["evaluated"], (r, oc["evaluated"])
        b.close(); g.close()
        return "refused"
    assert r["status"] == 0, r
    assert (r["first_valid"], r["first_forward"], r["evaluated"], r["bias"]) == (oc["first_valid"], oc["first_forward"], oc["evaluated"], oc["bias"])
    pat, loop, succ = b.unit_dag(0, K)
    assert oc["node2pat"] == [[] if p[0] == 0 else p for p in pat.tolist()]
    assert oc["node2loop"] == [[] if p[0] == 0 else p for p in loop.tolist()]
    assert [sum(1 << j for j in set(a)) for a in oc["adj"]] == [int(x) for x in succ]
    if orders:
        assert b.unit_orders(0, 0, r["num_orders"], K).tolist() == oc["orders"]
    assert b.unit_bkp(0).tolist() == oc["bkp"] and b.unit_path(0, 0).tolist() == oc["path"] and b.unit_path(0, 1).tolist() == oc["path_indel"]
    assert b.unit_out_juncs(0) == [tuple(x) for x in o["out_juncs"]]
    b.close(); g.close()
    return "ok"


def _cases(workdir, tag):
    out = []
    for (n, m, tier, K, seed) in [(256, 512, "chain", 64, 2), (256, 512, "chain", 80, 1), (256, 512, "chain", 127, 3), (140, 300, "mixed", 65, 1),
                                  (200, 420, "chain", 100, 4), (330, 700, "chain", 128, 5), (400, 840, "chain", 160, 6), (600, 1260, "chain", 255, 7),
                                  (330, 700, "mixed", 129, 2)]:
        s = synth.make_sample(n, m, tier, K, seed=seed, imperfect=seed % 2, n_del=seed % 3, n_dup=seed % 2, name="%s_%s%d" % (tag, tier, K))
        out.append((tier, K) + tuple(s.write(workdir)))
    return out


def check_wide(lib, oracle, workdir, tag):
    seen = set()
    for tier, K, lh, sols in _cases(workdir, tag):
        for rev in (False, True):
            seen.add(_check_unit(lib, oracle, lh, sols, rev))
        assert parity.compare(lib, oracle, lh, sols, keep_orders=(tier == "chain")) == []     # the whole stage-by-stage comparison, stdout lines included
    assert "ok" in seen
    return seen


def test_wide_units_on_the_host_simulation(hostsim_lib, oracle, workdir):
    check_wide(hostsim_lib, oracle, workdir, "wh")


def test_more_than_255_nodes_is_refused(hostsim_lib, workdir):
    s = synth.make_sample(620, 1300, "chain", 256, seed=9, name="w256")
    lh, sols = s.write(workdir)
    g = api.Graph(hostsim_lib, lh)
    b = api.Batch(hostsim_lib)
    with pytest.raises(api.AmbiError) as e:
        b.add_chromosome_sol(g, 0, sols[0])
    assert e.value.code == -10
    b.close(); g.close()


def check_mixed_batch(lib, oracle, workdir, tag, sharded=None):
    """wide and ordinary units in one batch (the express path is off for such a batch), and through the sharded driver"""
    specs = [(64, 128, "wide", 9, 11), (256, 512, "chain", 80, 12), (96, 200, "mixed", 11, 13), (140, 300, "mixed", 65, 14), (64, 128, "chain", 9, 15),
             (140, 300, "mixed", 65, 16)]      # (seed 14 with its two deletions selects a cyclic relation: R = 0)
    gs, samples, b = [], [], api.Batch(lib)
    for (n, m, tier, K, seed) in specs:
        s = synth.make_sample(n, m, tier, K, seed=seed, n_del=seed % 3, n_dup=seed % 2, name="%s_mb%d" % (tag, seed))
        lh, sols = s.write(workdir)
        gs.append(api.Graph(lib, lh)); samples.append((lh, sols))
        b.add_chromosome_sol(gs[-1], 0, sols[0])
    if sharded:
        b.run_sharded(0, devices=sharded)
    else:
        b.upload(); b.run(0); b.download()
    for u, (lh, sols) in enumerate(samples):
        oc = oracle.run_bfb(lh, sols, keep_orders=False)["chr"][0]
        r = b.unit_result(u)
        assert r["num_orders"] == oc["num_orders"] and r["evaluated"] == oc["evaluated"], (u, r)
        if oc["first_valid"] < 0:      # e.g. a cyclic relation among the selected elements: no order at all (R = 0)
            assert r["status"] == api.ST_NO_VALID_ORDER, (u, r)
            continue
        assert r["status"] == 0, (u, r)
        assert b.unit_path(u, 1).tolist() == oc["path_indel"] and b.unit_bkp(u).tolist() == oc["bkp"], u
    b.close()


def test_wide_and_ordinary_units_in_one_batch(hostsim_lib, oracle, workdir):
    check_mixed_batch(hostsim_lib, oracle, workdir, "wmb")
    check_mixed_batch(hostsim_lib, oracle, workdir, "wms", sharded=[0, 0])


def test_wide_units_all_mode_on_the_host_simulation(hostsim_lib, oracle, workdir):
    check_wide_all_mode(hostsim_lib, oracle, workdir, "wha", 80, 170)


def check_wide_all_mode(lib, oracle, workdir, tag, n=140, m=300):
    seen = set()
    for tier, K, seed in (("mixed", 65, 1), ("chain", 70, 2)):      # (few thousand orders at most: --all prints every valid one)
        s = synth.make_sample(n, m, tier, K, seed=seed, name="%s_all%s%d" % (tag, tier, K))
        lh, sols = s.write(workdir)
        o = oracle.run_bfb(lh, sols, all_=True, keep_orders=False)
        if o["chr"][0]["ub_valid"]:     # the reference's own behaviour is undefined on one of the orders --all evaluates: refused, by both
            e = api.reconstruct_sample(lib, lh, sols, all_=True)
            assert not e["ok"] and e["chr"][0]["status"] == -12
            seen.add("refused")
            continue
        if o["chr"][0]["first_valid"] < 0:   # no order assembles in either orientation (e.g. a cyclic relation: R = 0)
            e = api.reconstruct_sample(lib, lh, sols, all_=True)
            assert not e["ok"] and e["chr"][0]["status"] == api.ST_NO_VALID_ORDER and e["chr"][0]["num_orders"] == o["chr"][0]["num_orders"]
            seen.add("none")
            continue
        assert parity.compare(lib, oracle, lh, sols, all_=True, keep_orders=False) == []
        assert len(o["chr"][0]["all_paths"]) >= 1
        seen.add("ok")
    assert "ok" in seen


@pytest.mark.gpu
def test_wide_units_on_the_gpu(hip_lib, oracle, workdir):
    seen = check_wide(hip_lib, oracle, workdir, "wg")
    # larger order counts (tens of thousands of 128-byte rows) and both orientations
    for (tier, K, seed) in (("skew", 65, 3), ("mixed", 81, 2)):
        s = synth.make_sample(256, 512, tier, K, seed=seed, imperfect=seed % 2, n_del=seed % 3, n_dup=seed % 2, name="wg_%s%d" % (tier, K))
        lh, sols = s.write(workdir)
        for rev in (False, True):
            seen.add(_check_unit(hip_lib, oracle, lh, sols, rev, orders=False))
    assert seen == {"ok", "refused"}


@pytest.mark.gpu
def test_wide_units_mixed_batch_and_all_mode_on_the_gpu(hip_lib, oracle, workdir):
    check_mixed_batch(hip_lib, oracle, workdir, "wgb")
    check_mixed_batch(hip_lib, oracle, workdir, "wgs", sharded=[0, 0])
    check_wide_all_mode(hip_lib, oracle, workdir, "wga", 256, 512)
