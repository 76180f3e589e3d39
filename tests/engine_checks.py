"""Parity checks shared by the CPU host-simulation tests and the GPU tests (same assertions, different library)."""
import os
import numpy as np

import cases
import parity
from ambigram_amd import api


def check_fixed_and_synthetic(lib, oracle, workdir, small_only):
    bad = {}
    for name, lh, sols in cases.fixed_cases() + cases.synthetic_cases(workdir, small_only=small_only):
        d = parity.compare(lib, oracle, lh, sols)
        if d:
            bad[name] = d
        d = parity.compare(lib, oracle, lh, sols, reversed_=True, keep_orders=False)
        if d:
            bad[name + "/reversed"] = d
    assert not bad, bad


def check_search_budget(lib, oracle, workdir):
    """first_budget=1 forces the parallel search path whenever the first order is not the valid one."""
    bad = {}
    for name, lh, sols in cases.synthetic_cases(workdir, small_only=True)[:6]:
        d = parity.compare(lib, oracle, lh, sols, first_budget=1, keep_orders=False)
        if d:
            bad[name] = d
    assert not bad, bad


def check_random_decompositions(lib, oracle, workdir, seeds, budget=0):
    stats = dict(valid=0, none=0, deep=0, reversed_pass=0, refused=0)
    for seed in seeds:
        lh, sols = cases.random_decomposition(workdir, seed)
        for rev in (False, True):
            o = oracle.run_bfb(lh, sols, reversed_=rev, keep_orders=True)
            assert o["ok"], o["err"]
            oc = o["chr"][0]
            if oc["shortcut"]:
                continue
            g = api.Graph(lib, lh)
            b = api.Batch(lib)
            if budget:
                b.configure(first_budget=budget)
            b.add_chromosome_sol(g, 0, sols[0])
            b.upload(); b.run(api.FLAG_REVERSED if rev else 0); b.download()
            r = b.unit_result(0)
            if oc["ub_valid"]:
                # the reference reads out of bounds while rewriting a VALID order: its printed path is not defined by its
                # source; the engine refuses.  (The same read on an order that is invalid anyway changes nothing.)
                assert r["status"] == -12, (seed, r)
                stats["refused"] += 1
                continue
            assert r["num_orders"] == oc["num_orders"], (seed, rev)
            K = r["n_nodes"]
            if oc["num_orders"] and oc["num_orders"] <= 20000:
                assert b.unit_orders(0, 0, r["num_orders"], K).tolist() == oc["orders"], (seed, rev)
            if oc["first_valid"] < 0:
                assert r["status"] == api.ST_NO_VALID_ORDER, (seed, rev, r)
                assert r["evaluated"] == oc["evaluated"], (seed, rev, r["evaluated"], oc["evaluated"])
                stats["none"] += 1
            else:
                assert r["status"] == 0, (seed, rev, r)
                assert (r["first_valid"], r["first_forward"], r["evaluated"]) == (oc["first_valid"], oc["first_forward"], oc["evaluated"]), (seed, rev)
                assert b.unit_bkp(0).tolist() == oc["bkp"], (seed, rev)
                assert b.unit_path(0, 0).tolist() == oc["path"], (seed, rev)
                assert b.unit_path(0, 1).tolist() == oc["path_indel"], (seed, rev)
                stats["valid"] += 1
                if oc["first_valid"] > 0:
                    stats["deep"] += 1
                if oc["first_forward"] == (1 if rev else 0):
                    stats["reversed_pass"] += 1
            b.close(); g.close()
        for f in [lh] + list(sols):    # long runs (tests/soak.py) must not fill the temporary directory
            try:
                os.remove(f)
            except OSError:
                pass
    return stats


def check_edge_cases(lib, oracle, workdir):
    import os
    # (1) chromosome without any fold-back inversion -> shortcut path 1+..n+ (localhap.cpp:164-170)
    lh = os.path.join(workdir, "nofbi.lh")
    with open(lh, "w") as f:
        f.write("SAMPLE_NAME nofbi\nAVG_CHR_SEG_DP 30\nAVG_WHOLE_HOST_DP 30\nAVG_JUNC_DP 30\nPURITY 1\nAVG_TUMOR_PLOIDY 2\n"
                "PLOIDY 2m1\nVIRUS_START 5\nSOURCE 1\nSINK 4\n"
                "SEG H:1:chr1:1:10 30.0 1.0\nSEG H:2:chr1:11:20 30.0 1.0\nSEG H:3:chr1:21:30 30.0 1.0\nSEG H:4:chr1:31:40 30.0 1.0\n"
                "JUNC H:1:+ H:2:+ 30.0 1.0 U B\nJUNC H:2:+ H:3:+ 30.0 1.0 U B\nJUNC H:1:+ H:4:+ 30.0 1.0 U B\n")
    assert parity.compare(lib, oracle, lh, []) == []
    # (2) infeasible ILP -> reference path + "ILP is unsolvable." (localhap.cpp:213-220)
    sol = os.path.join(workdir, "infeasible.sol")
    with open(sol, "w") as f:
        f.write("Infeasible - objective value 0.00000000\n")
    assert parity.compare(lib, oracle, os.path.join(cases.DATA, "readme6.lh"), [sol]) == []
    # (3) a .sol that selects nothing -> the reference indexes an empty order; the engine refuses
    sol0 = os.path.join(workdir, "empty.sol")
    with open(sol0, "w") as f:
        f.write("Optimal - objective value 0.00000000\n")
    g = api.Graph(lib, os.path.join(cases.DATA, "readme6.lh"))
    b = api.Batch(lib)
    b.add_chromosome_sol(g, 0, sol0)
    b.upload(); b.run(0); b.download()
    assert b.unit_result(0)["status"] == -11
    # (4) missing .sol / missing .lh error codes (localhap.cpp:187-190, Graph.cpp:111-114)
    import pytest
    with pytest.raises(api.AmbiError) as e:
        b2 = api.Batch(lib); b2.add_chromosome_sol(g, 0, os.path.join(workdir, "nope.sol"))
    assert e.value.code == -7
    with pytest.raises(api.AmbiError) as e:
        api.Graph(lib, os.path.join(workdir, "nope.lh"))
    assert e.value.code == -1


def check_juncs_file(lib, oracle, workdir):
    """.juncs components (readComponents, LGM.cpp:5096-5156): adds junctions / bumps CN before getJuncCN."""
    import os
    j = os.path.join(workdir, "readme6.juncs")
    with open(j, "w") as f:
        f.write("6+ 6- 5- 4- 3- 2- 2+\n2- 2+ 3+ 4+ 5+ 6+ 6-\n6+ 6- 5- 4- 3-\n")   # README.md:173-177
    lh = os.path.join(cases.DATA, "readme6.lh")
    assert parity.compare(lib, oracle, lh, [os.path.join(cases.DATA, "readme6.sol")], juncs=j) == []


def check_batch_many_units(lib, oracle, workdir, n_samples):
    """Many independent samples in ONE batch == each sample alone (units do not interact)."""
    from ambigram_amd import synth
    graphs, b = [], api.Batch(lib)
    expect = []
    for i in range(n_samples):
        tier, K = [("chain", 7), ("wide", 7), ("mixed", 7)][i % 3]
        s = synth.make_sample(48, 100, tier, K, seed=5000 + i, imperfect=i % 2, n_del=i % 2)
        lh, sols = s.write(workdir, "b%d" % i)
        g = api.Graph(lib, lh)
        graphs.append(g)
        b.add_chromosome_sol(g, 0, sols[0])
        expect.append(oracle.run_bfb(lh, sols)["chr"][0])
    b.upload(); b.run(0); b.download()
    for i, oc in enumerate(expect):
        r = b.unit_result(i)
        assert r["status"] == 0 and r["num_orders"] == oc["num_orders"], i
        assert b.unit_path(i, 0).tolist() == oc["path"], i
        assert b.unit_path(i, 1).tolist() == oc["path_indel"], i
    # a second run over the resident batch gives the same answer (idempotence)
    b.run(0); b.download()
    for i, oc in enumerate(expect):
        assert b.unit_path(i, 1).tolist() == oc["path_indel"], i
    return b, graphs, expect


def check_enumerate_variants(lib, oracle, workdir, big=False):
    """The order table (allTopologicalOrders, LGM.cpp:3380-3409) through every enumeration path of the engine:
    block emission with several block sizes (AMBI_BLOCK_MAX: 1 = one row per block ... 1024), with an LDS budget that
    forces the general per-lane DFS (AMBI_BLOCK_LDS=64), and with few / many rows per lane -- all byte-identical to the
    oracle's orders."""
    import os
    from ambigram_amd import synth
    specs = [("wide", 11, 40, 90), ("mixed", 12, 48, 100), ("chain", 9, 40, 80), ("wide", 13, 64, 128)]
    if big:
        specs += [("wide", 15, 96, 200), ("mixed", 22, 128, 256), ("chain", 35, 256, 512)]
    specs += [("skew", 23, 64, 128), ("skew", 27, 64, 128), ("skew", 34, 96, 200), ("skew", 45, 128, 256), ("skew", 50, 128, 256)]
    samples = []
    for i, (tier, K, nseg, njunc) in enumerate(specs):
        s = synth.make_sample(nseg, njunc, tier, K, seed=7100 + i)
        lh, sols = s.write(workdir, "ev%d" % i)
        oc = oracle.run_bfb(lh, sols, keep_orders=True)["chr"][0]
        samples.append((lh, sols, oc))
    variants = [({}, 0), ({"AMBI_BLOCK_MAX": "1"}, 0), ({"AMBI_BLOCK_MAX": "6"}, 64), ({"AMBI_BLOCK_MAX": "1024", "AMBI_BLOCK_LDS": "150000"}, 0),
                ({"AMBI_BLOCK_LDS": "64"}, 0), ({"AMBI_BLOCK_LDS": "64"}, 100000), ({"AMBI_BLOCK_MAX": "37"}, 1 << 20),
                # budgets in which the directory of most units does not fit but tables + suffix rows do: the
                # directory-free block walk, started at the top of the table and (few rows per lane) in the middle of it
                ({"AMBI_BLOCK_LDS": "12288"}, 0), ({"AMBI_BLOCK_LDS": "8192", "AMBI_BLOCK_MAX": "24"}, 256), ({"AMBI_BLOCK_LDS": "6144"}, 64)]
    saved = {k: os.environ.get(k) for k in ("AMBI_BLOCK_MAX", "AMBI_BLOCK_LDS")}
    try:
        for env, lanes in variants:
            for k in saved:
                os.environ.pop(k, None)
            os.environ.update(env)
            graphs, b = [], api.Batch(lib)
            if lanes:
                b.configure(target_lanes=lanes)
            for lh, sols, oc in samples:
                g = api.Graph(lib, lh)
                graphs.append(g)
                b.add_chromosome_sol(g, 0, sols[0])
            b.upload(); b.run(0); b.download()
            for u, (lh, sols, oc) in enumerate(samples):
                r = b.unit_result(u)
                assert r["status"] == 0 and r["num_orders"] == oc["num_orders"], (env, lanes, u, r)
                got = b.unit_orders(u, 0, r["num_orders"], r["n_nodes"])
                if oc["orders"]:
                    assert got.tolist() == oc["orders"], (env, lanes, u)
                assert b.unit_path(u, 1).tolist() == oc["path_indel"], (env, lanes, u)
            b.close()
            for g in graphs:
                g.close()
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def check_large_lattice(lib, oracle, workdir, K=50, k2=5):
    """A lattice with more ideals than the group-local search holds (256): (K-k2)(k2+1)+1 ideals take the HBM path of
    the lattice search.  R = C(K-1,k2) rows; the table is checked against the oracle's order count, by sampling rows
    (permutation, DAG-respecting, lexicographically increasing) and through the reconstructed path."""
    from ambigram_amd import synth
    s = synth.make_sample(128, 256, "skew%d" % k2, K, seed=7300 + K)
    lh, sols = s.write(workdir, "ll%d" % K)
    oc = oracle.run_bfb(lh, sols)["chr"][0]
    g = api.Graph(lib, lh)
    b = api.Batch(lib)
    b.add_chromosome_sol(g, 0, sols[0])
    b.upload(); b.run(0); b.download()
    r = b.unit_result(0)
    assert r["status"] == 0 and r["num_orders"] == oc["num_orders"], (r, oc["num_orders"])
    assert (K - k2) * (k2 + 1) + 1 > 256
    R = r["num_orders"]
    for first in (0, R // 3, R - 5000):
        rows = b.unit_orders(0, first, 5000, K)
        assert np.array_equal(np.sort(rows, axis=1), np.tile(np.arange(K, dtype=np.uint8), (len(rows), 1)))
        a, bb = rows[:-1].astype(np.int16), rows[1:].astype(np.int16)
        neq = a != bb
        fi = neq.argmax(axis=1)
        ii = np.arange(len(a))
        assert neq.any(axis=1).all() and (a[ii, fi] < bb[ii, fi]).all()
    assert b.unit_path(0, 0).tolist() == oc["path"] and b.unit_path(0, 1).tolist() == oc["path_indel"]
    b.close(); g.close()


def check_all_mode(lib, oracle, workdir, seeds=range(40)):
    """--all (LGM.cpp:3672-3695): every valid order is reported, in the reference's print order, with the orientation
    flip only when the last order of the first pass is invalid; `evaluated` counts every order of the executed passes."""
    bad = {}
    for name, lh, sols in cases.fixed_cases() + cases.synthetic_cases(workdir, small_only=True)[:8]:
        for rev in (False, True):
            d = parity.compare(lib, oracle, lh, sols, reversed_=rev, all_=True, keep_orders=False)
            if d:
                bad[name + ("/reversed" if rev else "")] = d
    stats = dict(flipped=0, multi=0, none=0)
    for seed in seeds:
        lh, sols = cases.random_decomposition(workdir, seed)
        for rev in (False, True):
            o = oracle.run_bfb(lh, sols, reversed_=rev, all_=True)
            assert o["ok"], o["err"]
            oc = o["chr"][0]
            if oc["shortcut"] or oc["ub_valid"]:
                continue
            d = parity.compare(lib, oracle, lh, sols, reversed_=rev, all_=True, keep_orders=False)
            if oc["first_valid"] < 0:
                # no valid order: the engine reports the status, the oracle an empty path
                stats["none"] += 1
                continue
            if d:
                bad["random%d%s" % (seed, "/reversed" if rev else "")] = d
            stats["multi"] += len(oc["all_paths"]) > 1
            stats["flipped"] += oc["evaluated"] > oc["num_orders"]
    assert not bad, bad
    return stats


def check_mixed_batch(lib, oracle, workdir, big=False):
    """ONE batch with every kind of unit side by side: all three row-width classes of the enumerate kernels (K <= 20,
    <= 32, <= 63), units whose block image does not fit next to units whose image does (small AMBI_BLOCK_LDS budget),
    a chromosome without fold-backs (shortcut), an infeasible .sol, units without any valid order and units whose first
    valid order lies behind the scan budget.  Every unit equals its own oracle run."""
    import os
    from ambigram_amd import synth
    specs = [("chain", 9, 48, 100), ("wide", 11, 48, 100), ("mixed", 12, 48, 100), ("skew", 23, 64, 128), ("skew", 27, 64, 128),
             ("skew", 34, 96, 200), ("skew", 45, 128, 256), ("wide", 13, 64, 128), ("chain", 25, 128, 256)]
    if big:
        specs += [("wide", 15, 96, 200), ("skew", 50, 128, 256), ("wide", 19, 256, 512)]
    items = []
    for i, (tier, K, nseg, njunc) in enumerate(specs):
        s = synth.make_sample(nseg, njunc, tier, K, seed=7600 + i, imperfect=i % 2, n_del=i % 3)
        lh, sols = s.write(workdir, "mb%d" % i)
        items.append((lh, sols[0]))
    for seed in (3, 5, 8, 13, 21, 34):
        lh, sols = cases.random_decomposition(workdir, 900 + seed)
        items.append((lh, sols[0]))
    # shortcut (no fold-back) and infeasible
    nofbi = os.path.join(workdir, "mb_nofbi.lh")
    with open(nofbi, "w") as f:
        f.write("SAMPLE_NAME nofbi\nAVG_CHR_SEG_DP 30\nAVG_WHOLE_HOST_DP 30\nAVG_JUNC_DP 30\nPURITY 1\nAVG_TUMOR_PLOIDY 2\n"
                "PLOIDY 2m1\nVIRUS_START 5\nSOURCE 1\nSINK 4\n"
                "SEG H:1:chr1:1:10 30.0 1.0\nSEG H:2:chr1:11:20 30.0 1.0\nSEG H:3:chr1:21:30 30.0 1.0\nSEG H:4:chr1:31:40 30.0 1.0\n"
                "JUNC H:1:+ H:2:+ 30.0 1.0 U B\nJUNC H:2:+ H:3:+ 30.0 1.0 U B\nJUNC H:1:+ H:4:+ 30.0 1.0 U B\n")
    infeasible = os.path.join(workdir, "mb_infeasible.sol")
    with open(infeasible, "w") as f:
        f.write("Infeasible - objective value 0.00000000\n")
    items.append((nofbi, None))
    items.append((os.path.join(cases.DATA, "readme6.lh"), infeasible))
    expect = []
    for lh, sol in items:
        o = oracle.run_bfb(lh, [sol] if sol else [], keep_orders=True)
        assert o["ok"], o["err"]
        expect.append(o["chr"][0])
    saved = {k: os.environ.get(k) for k in ("AMBI_BLOCK_LDS",)}
    try:
        for lds in (None, "6000"):          # 6000 bytes: only the smallest images fit, the rest take the general path
            os.environ.pop("AMBI_BLOCK_LDS", None)
            if lds:
                os.environ["AMBI_BLOCK_LDS"] = lds
            graphs, b = [], api.Batch(lib)
            b.configure(first_budget=3)
            for lh, sol in items:
                g = api.Graph(lib, lh)
                graphs.append(g)
                if sol:
                    b.add_chromosome_sol(g, 0, sol)
                else:
                    b.add_chromosome(g, 0, [], [])
            b.upload(); b.run(0); b.download()
            for u, oc in enumerate(expect):
                r = b.unit_result(u)
                tag = (lds, u, items[u][0])
                if oc["shortcut"] or oc["infeasible"]:
                    assert r["status"] == (api.ST_SHORTCUT if oc["shortcut"] else api.ST_INFEASIBLE), tag
                    want = oc["path_indel"] or oc["path"]      # the reference path 1+ .. n+ (no indelBFB on this branch)
                    assert b.unit_path(u, 1).tolist() == want or not want, tag
                    continue
                if oc["ub_valid"]:
                    assert r["status"] == -12, tag
                    continue
                assert r["num_orders"] == oc["num_orders"], tag
                if oc["orders"]:
                    assert b.unit_orders(u, 0, r["num_orders"], r["n_nodes"]).tolist() == oc["orders"], tag
                if oc["first_valid"] < 0:
                    assert r["status"] == api.ST_NO_VALID_ORDER and r["evaluated"] == oc["evaluated"], tag
                    continue
                assert r["status"] == 0, (tag, r)
                assert (r["first_valid"], r["first_forward"], r["evaluated"]) == (oc["first_valid"], oc["first_forward"], oc["evaluated"]), tag
                assert b.unit_bkp(u).tolist() == oc["bkp"], tag
                assert b.unit_path(u, 0).tolist() == oc["path"] and b.unit_path(u, 1).tolist() == oc["path_indel"], tag
            b.close()
            for g in graphs:
                g.close()
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def check_max_sizes(lib, oracle, workdir):
    """The engine's documented limits: 63 DAG nodes per unit on the fast path, 64..255 on the wide path
    (tests/test_wide_units.py), 256 are refused when the unit is added;
    a path longer than the 65 536 cells the full finish stage can hold in group memory is served by the lean stage (runs
    only); if its SVs edit the path, by the direct full-stage launch with the cells in device memory -- ST_ERR_PATH_CAPACITY
    only where the group-memory form runs."""
    import pytest
    from ambigram_amd import synth
    for tier, K in (("chain", 63), ("skew", 63)):
        s = synth.make_sample(512, 1024, tier, K, seed=4242)
        lh, sols = s.write(workdir, "max_%s%d" % (tier, K))
        o = oracle.run_bfb(lh, sols)["chr"][0]
        assert len(o["path"]) > 65536                       # really beyond the group-memory limit
        g = api.Graph(lib, lh)
        b = api.Batch(lib)
        b.add_chromosome_sol(g, 0, sols[0])
        b.upload(); b.run(0); b.download()
        r = b.unit_result(0)
        assert r["status"] == 0 and r["n_nodes"] == K and r["num_orders"] == o["num_orders"], r
        assert b.unit_path(0, 0).tolist() == o["path"] and b.unit_path(0, 1).tolist() == o["path_indel"]
        b.close(); g.close()
    s = synth.make_sample(620, 1300, "chain", 256, seed=4242)
    lh, sols = s.write(workdir, "max_chain256")
    g = api.Graph(lib, lh)
    with pytest.raises(api.AmbiError) as e:
        api.Batch(lib).add_chromosome_sol(g, 0, sols[0])
    assert e.value.code == -10
    g.close()
    s = synth.make_sample(512, 1024, "chain", 64, seed=4242)            # one node more than the fast path takes: the wide path, same long path
    lh, sols = s.write(workdir, "max_chain64")
    o = oracle.run_bfb(lh, sols)["chr"][0]
    g = api.Graph(lib, lh)
    b = api.Batch(lib)
    b.add_chromosome_sol(g, 0, sols[0])
    b.upload(); b.run(0); b.download()
    r = b.unit_result(0)
    assert r["status"] == 0 and r["n_nodes"] == 64 and r["num_orders"] == o["num_orders"], r
    assert b.unit_path(0, 0).tolist() == o["path"] and b.unit_path(0, 1).tolist() == o["path_indel"]
    b.close(); g.close()
    s = synth.make_sample(512, 1024, "chain", 63, seed=4242, n_del=2)      # long path AND an SV that edits it
    lh, sols = s.write(workdir, "max_chain63_del")
    o = oracle.run_bfb(lh, sols)["chr"][0]
    assert len(o["path"]) > 65536 and o["path_indel"] != o["path"]
    g = api.Graph(lib, lh)
    b = api.Batch(lib)
    b.add_chromosome_sol(g, 0, sols[0])
    b.upload(); b.run(0); b.download()
    r = b.unit_result(0)
    if r["status"] == 0:    # the direct full-stage launch keeps the path cells in device memory: no limit but the unit's own capacity
        assert b.unit_path(0, 0).tolist() == o["path"] and b.unit_path(0, 1).tolist() == o["path_indel"]
    else:                   # the form with the cells in group memory (AMBI_DIRECT_EXT=0, or a unit the lean stage handed over)
        assert r["status"] == -14
    b.close(); g.close()


def check_injected_validity(lib, oracle, workdir):
    """Control flow around the per-order evaluation of getBFB (LGM.cpp:3519-3696) at places no known input reaches.

    An exhaustive search (tests/tools/search_mixed_validity.py: every element set of up to 5 patterns/loops over 4 and 5
    segments, both orientations, 360 000+ runs of the oracle) found NO unit whose orders differ in validity, so on real
    inputs the first valid order is always order 0 of one of the two passes.  The scan budget, the parallel search for
    the minimum index, the error-before-valid rule, the orientation flip and --all are therefore driven here through the
    diagnostics hook ambi_batch_debug_inject_validity: verdicts are injected, everything else (orders, breakpoints,
    paths) is the real thing and is compared with the oracle's --all output for the same order."""
    from ambigram_amd import synth
    s = synth.make_sample(48, 100, "wide", 9, seed=9100)
    lh, sols = s.write(workdir, "inj")
    of = oracle.run_bfb(lh, sols, all_=True)["chr"][0]
    orv = oracle.run_bfb(lh, sols, all_=True, reversed_=True)["chr"][0]
    R = of["num_orders"]
    assert R == 70 and len(of["all_paths"]) == R and len(orv["all_paths"]) == R      # every order valid in both orientations
    E = -12    # AMBI_ERR_REF_UB

    def run(fwd, rev, flags=0, budget=4):
        g = api.Graph(lib, lh)
        b = api.Batch(lib)
        b.configure(first_budget=budget)
        b.add_chromosome_sol(g, 0, sols[0])
        b.debug_inject_validity(0, list(fwd) + list(rev))
        b.upload(); b.run(flags); b.download()
        return g, b, b.unit_result(0)

    zeros = [0] * R
    mid = R // 2
    cases = [
        # (forward verdicts, reversed verdicts, expected (status, first_valid, first_forward, evaluated))
        ("hit at 1 (inside the scan budget)", [0, 1] + [127] * (R - 2), zeros, (0, 1, 1, 2)),
        ("hit mid-table, later orders valid too", [0] * mid + [1] * (R - mid), zeros, (0, mid, 1, mid + 1)),
        ("hit at the last order", [0] * (R - 1) + [1], zeros, (0, R - 1, 1, R)),
        ("flip, hit at 3", zeros, [0, 0, 0, 1] + [1] * (R - 4), (0, 3, 0, R + 4)),
        ("flip, hit at the last order", zeros, [0] * (R - 1) + [1], (0, R - 1, 0, 2 * R)),
        ("no valid order", zeros, zeros, (api.ST_NO_VALID_ORDER, -1, -1, 2 * R)),
        ("valid at 5 before undefined at 10", [0] * 5 + [1] + [0] * 4 + [E] + [1] * (R - 11), zeros, (0, 5, 1, 6)),
        ("undefined at 5 before valid at 10", [0] * 5 + [E] + [0] * 4 + [1] * (R - 10), zeros, (E, -1, -1, 6)),
        ("undefined in the flipped pass before its first valid order", zeros, [0] * 20 + [E] + [0] * 9 + [1] * (R - 30), (E, -1, -1, R + 21)),
        ("valid and undefined in the same chunk, valid first", [0] * 33 + [1, E] + [0] * (R - 35), zeros, (0, 33, 1, 34)),
    ]
    for name, fwd, rev, want in cases:
        for budget in (4, 64, 1):
            g, b, r = run(fwd, rev, budget=budget)
            got = (r["status"], r["first_valid"], r["first_forward"], r["evaluated"])
            if want[0] < 0:
                assert got[0] == want[0] and got[3] == want[3], (name, budget, got, want)
            else:
                assert got == want, (name, budget, got, want)
            if want[0] == 0:      # the winner's path is the oracle's path of that very order and orientation
                ref = (of if want[2] == 1 else orv)["all_paths"][want[1]]
                assert b.unit_path(0, 0).tolist() == ref, (name, budget)
            b.close(); g.close()
    # --all: bitmaps, counts, flip rule and `evaluated` with mixed verdicts
    import random
    rng = random.Random(5)
    for last_valid in (True, False):
        fwd = [rng.randint(0, 1) for _ in range(R)]
        rev = [rng.randint(0, 1) for _ in range(R)]
        fwd[0] = 1                       # the default-mode scan finds order 0 (the unit's ordinary results)
        fwd[-1] = 1 if last_valid else 0
        g, b, r = run(fwd, rev, flags=api.FLAG_ALL, budget=64)
        assert r["status"] == 0 and r["evaluated"] == (R if last_valid else 2 * R), (last_valid, r)
        i0 = b.all_orders(0, 0).tolist()
        i1 = b.all_orders(0, 1).tolist()
        assert i0 == [i for i in range(R) if fwd[i]], last_valid
        assert i1 == ([] if last_valid else [i for i in range(R) if rev[i]]), last_valid
        got = b.all_paths(0, 0, 0, len(i0), 4096)
        assert [p.tolist() for p in got] == [of["all_paths"][i] for i in i0]
        if i1:
            got = b.all_paths(0, 1, 0, len(i1), 4096)
            assert [p.tolist() for p in got] == [orv["all_paths"][i] for i in i1]
        b.close(); g.close()
    # --all with an undefined order: the unit is refused
    fwd = [1] * R
    fwd[40] = E
    g, b, r = run(fwd, [1] * R, flags=api.FLAG_ALL, budget=64)
    assert r["status"] == E, r
    b.close(); g.close()


def check_all_two_forms(lib, workdir, seeds=range(300, 420)):
    """--all evaluates an order with one THREAD (ambi_eval_lane.hpp, units with a short breakpoint path) or with one
    WAVEFRONT (ambi_eval.hpp): the same valid-order lists, counts, flip decisions and statuses from both, on random
    decompositions (most orders invalid, both orientations) and on wide synthetic samples (every order valid)."""
    import os
    from ambigram_amd import synth
    items = []
    for seed in seeds:
        lh, sols = cases.random_decomposition(workdir, seed)
        items.append((lh, sols[0]))
    for i, (tier, K) in enumerate([("wide", 9), ("mixed", 9), ("wide", 13), ("chain", 7), ("mixed", 17), ("mixed", 19)]):
        s = synth.make_sample(64, 128, tier, K, seed=8800 + i, imperfect=i % 2)
        lh, sols = s.write(workdir, "tf%d" % i)
        items.append((lh, sols[0]))
    saved = os.environ.get("AMBI_ALL_LANES")
    got = {}
    try:
        for form in ("1", "0"):
            os.environ["AMBI_ALL_LANES"] = form
            for rev in (0, api.FLAG_REVERSED):
                graphs, b = [], api.Batch(lib)
                for lh, sol in items:
                    g = api.Graph(lib, lh); graphs.append(g)
                    b.add_chromosome_sol(g, 0, sol)
                b.upload(); b.run(api.FLAG_ALL | rev); b.download()
                rec = []
                for u in range(len(items)):
                    r = b.unit_result(u)
                    rec.append((r["status"], r["evaluated"], b.all_orders(u, 0).tolist(), b.all_orders(u, 1).tolist()))
                got[(form, rev)] = rec
                b.close()
                for g in graphs:
                    g.close()
    finally:
        os.environ.pop("AMBI_ALL_LANES", None)
        if saved is not None:
            os.environ["AMBI_ALL_LANES"] = saved
    for rev in (0, api.FLAG_REVERSED):
        a, c = got[("1", rev)], got[("0", rev)]
        for u, (x, y) in enumerate(zip(a, c)):
            assert x == y, (rev, u, items[u][0], x[:2], y[:2])
    assert sum(1 for x in got[("1", 0)] if x[0] == 0 and len(x[2]) > 1) >= 2


def check_arena_limit(lib, workdir, n_units=40, budget=3, seeds=range(9300, 9340)):
    """The order-table arena has an upper limit (AMBI_ARENA_MAX_BYTES): the plan stage gives rows to the units that fit, in
    unit order, and the others end with ORDERS_CAPACITY (-15) -- whatever the scan for the first valid order, which may run
    BESIDE the plan stage, made of them.  Units that got their rows are exactly what they are without the limit; runs are
    repeatable; the same batch without the limit reconstructs every unit."""
    import os
    from ambigram_amd import synth
    items = []
    for i, seed in zip(range(n_units), seeds):
        s = synth.make_sample(48, 100, ("chain", "wide", "mixed")[i % 3], 9 + 2 * (i % 3), seed=seed, imperfect=i % 2, n_del=(i % 5 == 4) * 2, n_dup=(i % 7 == 6) * 1)
        lh, sols = s.write(workdir, "al%d_%d" % (n_units, i))
        items.append((lh, sols[0]))

    def run(cap):
        os.environ.pop("AMBI_ARENA_MAX_BYTES", None)
        if cap:
            os.environ["AMBI_ARENA_MAX_BYTES"] = str(cap)
        try:
            graphs, b = [], api.Batch(lib)
            b.configure(order_arena_bytes=4096, first_budget=budget)
            for lh, sol in items:
                g = api.Graph(lib, lh); graphs.append(g)
                b.add_chromosome_sol(g, 0, sol)
            b.upload()
            outs = []
            for flags in (0, 0, api.FLAG_REVERSED):
                b.run(flags); b.wait(); b.download()
                out = []
                for u in range(len(items)):
                    r = dict(b.unit_result(u))
                    if r["status"] == 0:
                        r["path"] = b.unit_path(u, 0).tolist(); r["path_indel"] = b.unit_path(u, 1).tolist(); r["out"] = b.unit_out_juncs(u)
                        r["orders"] = b.unit_orders(u, 0, min(r["num_orders"], 200), r["n_nodes"]).tolist()
                    out.append(r)
                outs.append(out)
            b.close()
            for g in graphs:
                g.close()
            return outs
        finally:
            os.environ.pop("AMBI_ARENA_MAX_BYTES", None)

    free = run(None)
    assert all(r["status"] != -15 for out in free for r in out)
    with_rows = [u for u, r in enumerate(free[0]) if r["status"] == 0 and r["num_orders"] > 0]
    assert len(with_rows) >= n_units // 2
    cap = 4096 * (len(with_rows) // 2)                      # every table of these small units is one 4 KB granule
    lim = run(cap)
    for u, (x, y) in enumerate(zip(lim[0], lim[1])):
        assert x == y, ("two runs of the limited batch differ", u, {k: (x.get(k), y.get(k)) for k in set(x) | set(y) if x.get(k) != y.get(k)})
    n_refused = 0
    for flags_i in (0, 2):
        refused = [u for u, r in enumerate(lim[flags_i]) if r["status"] == -15]
        wanted = [u for u, r in enumerate(free[flags_i]) if r["num_orders"] > 0 and free[flags_i][u]["status"] not in (1, 2)]
        assert refused and len(refused) < len(wanted), (cap, refused)
        assert refused == [u for u in wanted if u >= refused[0]], (refused, wanted)   # unit order: the tail of the units with rows
        for u, (x, y) in enumerate(zip(lim[flags_i], free[flags_i])):
            if u in refused:
                continue
            assert x == y, (flags_i, u, items[u][0])
        n_refused = len(refused)
    return n_refused


def check_large_batch_of_pending_units(lib, oracle, workdir, seeds=range(20000, 20600), budget=1, runs=2):
    """ONE batch of many random decompositions with a scan budget of one order: most of them have no valid order at all and
    more orders than the budget, so their scan ends PENDING and the parallel search resolves them -- while, in the same
    launch, the plan stage hands out rows beside the scan (they run on two streams and must not share a word: round 1 let
    the plan stage read the status the scan rewrites).  Every unit equals its own oracle run: status, counts, the order
    table, paths.  The batch is run twice (first run: arena sizing; second: the resident path)."""
    units = []
    for seed in seeds:
        lh, sols = cases.random_decomposition(workdir, seed)
        o = oracle.run_bfb(lh, sols, keep_orders=True)
        assert o["ok"], o["err"]
        oc = o["chr"][0]
        if oc["shortcut"]:
            continue
        units.append((seed, lh, sols[0], oc))
    graphs, b = [], api.Batch(lib)
    b.configure(first_budget=budget)
    for _, lh, sol, _ in units:
        g = api.Graph(lib, lh); graphs.append(g)
        b.add_chromosome_sol(g, 0, sol)
    b.upload()
    stats = dict(units=len(units), none_pending=0, valid=0, refused=0)
    for run in range(runs):
        b.run(0); b.wait(); b.download()
        for u, (seed, lh, sol, oc) in enumerate(units):
            r = b.unit_result(u)
            if oc["ub_valid"]:
                assert r["status"] == -12, (run, seed, r)
                stats["refused"] += run == 0
                continue
            assert r["num_orders"] == oc["num_orders"], (run, seed)
            if 0 < oc["num_orders"] <= 5000:
                assert b.unit_orders(u, 0, r["num_orders"], r["n_nodes"]).tolist() == oc["orders"], (run, seed)
            if oc["first_valid"] < 0:
                assert r["status"] == api.ST_NO_VALID_ORDER, (run, seed, r)
                assert r["evaluated"] == oc["evaluated"], (run, seed, r["evaluated"], oc["evaluated"])
                stats["none_pending"] += run == 0 and oc["num_orders"] > budget
            else:
                assert r["status"] == 0, (run, seed, r)
                assert (r["first_valid"], r["first_forward"], r["evaluated"]) == (oc["first_valid"], oc["first_forward"], oc["evaluated"]), (run, seed)
                assert b.unit_bkp(u).tolist() == oc["bkp"] and b.unit_path(u, 0).tolist() == oc["path"] and b.unit_path(u, 1).tolist() == oc["path_indel"], (run, seed)
                stats["valid"] += run == 0
    b.close()
    for g in graphs:
        g.close()
    assert stats["none_pending"] >= len(units) // 3, stats
    return stats
