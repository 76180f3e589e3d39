"""`--op sc_bfb` (localhap.cpp:390-679): several graphs with the same segmentation, one joint ILP per
chromosome, per-graph reconstruction downstream of the joint .sol -- all graphs x chromosomes as ONE batch of units."""
import os

import numpy as np
import pytest

from ambigram_amd import api, synth


def make_cells(workdir, tag, G, n_seg=48, n_junc=100, n_chr=1, K=5, prop=None, **kw):
    """G samples over the same segmentation with different planted decompositions + the joint .sol per chromosome"""
    samples = [synth.make_sample(n_seg, n_junc, ("chain", "wide", "mixed")[k % 3], K, seed=9300 + 10 * k + len(tag), n_chr=n_chr,
                                 prop=prop, name="%s_cell%d" % (tag, k), **kw) for k in range(G)]
    lhs = [s.write(workdir)[0] for s in samples]
    sols = []
    for c, text in enumerate(synth.joint_sol_texts(samples)):
        p = os.path.join(workdir, "%s.joint.chr%d.sol" % (tag, c))
        with open(p, "w") as f:
            f.write(text)
        sols.append(p)
    return samples, lhs, sols


def compare_sc(lib, oracle, lhs, sols, **kw):
    o = oracle.run_sc_bfb(lhs, sols, **kw)
    e = api.reconstruct_sc(lib, lhs, sols, **kw)
    assert o["ok"], o["err"]
    assert e["ok"], e["err"]
    assert e["log"] == o["log"]
    assert e["paths"] == o["paths"]
    assert e["trx_paths"] == o["trx_paths"]
    for c, (es, os_) in enumerate(zip(e["chr"], o["chr"])):
        for k, (a, b) in enumerate(zip(es, os_)):
            assert a["shortcut"] == b["shortcut"] and a["infeasible"] == b["infeasible"], (c, k)
            if a["shortcut"] or a["infeasible"]:
                continue
            for key in ("num_orders", "first_valid", "first_forward", "evaluated", "node2pat", "node2loop", "bkp", "path", "path_indel", "indel_printed"):
                assert a[key] == b[key], (c, k, key)
    return o


def check_sc(lib, oracle, workdir):
    # three cells, one chromosome; both orientations; --all
    _, lhs, sols = make_cells(workdir, "a", 3, imperfect=1, n_del=1)
    compare_sc(lib, oracle, lhs, sols)
    compare_sc(lib, oracle, lhs, sols, reversed_=True)
    compare_sc(lib, oracle, lhs, sols, all_=True)
    # a single graph (G = 1)
    _, lhs1, sols1 = make_cells(workdir, "b", 1)
    compare_sc(lib, oracle, lhs1, sols1)
    # three chromosomes with BFB-TRX (translocationBFB per graph, localhap.cpp:661-664) -- 2 cells x 3 chromosomes = 6 units
    _, lhs3, sols3 = make_cells(workdir, "c", 2, n_seg=72, n_junc=150, n_chr=3, translocations=1, trx_insertions=1, prop="PROP C2:chr1:chr2 M:chr1")
    o = compare_sc(lib, oracle, lhs3, sols3)
    assert o["log"].count("BFB with translocation:") == 2
    # infeasible joint solution: one "ILP is unsolvable." line, reference paths for every graph, nothing else printed
    inf = os.path.join(workdir, "sc_inf.sol")
    with open(inf, "w") as f:
        f.write("Infeasible - objective value 0.00000000\n")
    o = compare_sc(lib, oracle, lhs, [inf])
    assert o["log"][-1] == "ILP is unsolvable." and o["paths"][2][0] == list(range(1, 49))
    # the first graph has no fold-back on its chromosome: every graph gets the reference path silently (localhap.cpp:505-512),
    # whatever the other graphs hold
    plain = os.path.join(workdir, "sc_plain.lh")
    with open(plain, "w") as f:
        f.write("\n".join(l for l in open(lhs[0]).read().splitlines()          # keep same-strand junctions only
                          if not (l.startswith("JUNC") and l.split()[1].split(":")[2] != l.split()[2].split(":")[2])) + "\n")
    o = compare_sc(lib, oracle, [plain, lhs[1]], [])
    assert o["paths"] == [[list(range(1, 49))], [list(range(1, 49))]] and not any("|" in l for l in o["log"])


def test_sc_bfb_hostsim(hostsim_lib, oracle, workdir):
    check_sc(hostsim_lib, oracle, workdir)


@pytest.mark.gpu
def test_sc_bfb_gpu(hip_lib, oracle, workdir):
    check_sc(hip_lib, oracle, workdir)


def test_sc_joint_ilp_equals_literal_restatement(hostsim_lib, oracle, workdir):
    """ambi_ilp_build_sc (closed form, O(nnz)) == the oracle's literal restatement of BFB_ILP_SC (LGM.cpp:4754-5093) for
    G = 1, 2, 3: CSR, bounds, objective, integrality -- including the reference's epsilon columns numbered by the running
    row counter (LGM.cpp:4815)."""
    for G, n_seg in ((1, 12), (2, 9), (3, 7), (2, 20)):
        samples, lhs, _ = make_cells(workdir, "ilp%d_%d" % (G, n_seg), G, n_seg=n_seg, n_junc=2 * n_seg + 4, K=3)
        want = oracle.ilp_sc(lhs, 0)
        assert want["ok"]
        graphs = [api.Graph(hostsim_lib, p) for p in lhs]
        hostsim_lib.ambi_graph_recalculate(graphs[0].h)
        n = n_seg
        seg = np.zeros((G, n)); fold = np.zeros((G, n))
        for k, g in enumerate(graphs):
            b = api.Batch(hostsim_lib)
            b.add_chromosome(g, 0, [], [])
            b.upload(); b.run(0); b.download()
            prep = b.unit_prepare(0, n)
            fold[k] = np.asarray(prep["junc_cn"])[1:, 1]
            seg[k] = np.asarray(prep["seg_cn"])[1:] if k == 0 else g.segments()["cn"]      # getIndelBias: first graph only
            b.close()
        m = api.IlpModel.joint(hostsim_lib, graphs[0], 0, seg, fold)
        A = m.arrays()
        assert m.n_cols == want["n_cols"] and m.n_int == want["n_int"] and m.n_rows == len(want["row_lo"])
        assert A["row_ptr"].tolist() == want["row_ptr"] and A["col"].tolist() == want["col"] and A["val"].tolist() == want["val"]
        for key in ("row_lo", "row_up", "col_lo", "col_up", "obj"):
            assert A[key].tolist() == want[key], key
        # what the running row counter does to the epsilon columns: in range for every G, but for G > 1 the segment
        # epsilons of the later graphs land among the linking epsilons
        assert max(want["col"]) < want["n_cols"]
        m.close()
        for g in graphs:
            g.close()


def check_sc_cli(exe, oracle, cwd):
    """the drop-in CLI: `--op sc_bfb --in_lh a.lh,b.lh`; stdout line for line the oracle's, time.csv row (localhap.cpp:666-678)"""
    import test_cli_dropin as t
    _, lhs, sols = make_cells(cwd, "cli", 2, n_seg=60, n_junc=126, n_chr=3, translocations=1, prop="PROP C2:chr1:chr2 M:chr1", n_del=1)
    bindir = os.path.join(cwd, "bin")
    t.fake_cbc(bindir, sols)
    r = t.run_cli(exe, cwd, bindir, "--op", "sc_bfb", "--in_lh", ",".join(lhs), "--lp_prefix", "cells")
    assert r.returncode == 0, r.stderr
    got = [l for l in r.stdout.splitlines() if not l.startswith("fake cbc")]
    want = oracle.run_sc_bfb(lhs, sols)
    assert got == want["log"]
    row = open(os.path.join(cwd, "time.csv")).read().strip().split(",")
    assert row[0] == lhs[0][:lhs[0].find(".")] and row[1:7] == [str(want["n_seg"]), "0", str(want["n_junc"]), str(want["cn_sum"]), str(want["path_len"]), str(want["max_cn"])]
    assert all(os.path.exists(os.path.join(cwd, "cells." + ext)) for ext in ("lp", "mps", "sol"))


def test_sc_bfb_cli(hostsim_lib, oracle, tmp_path):
    check_sc_cli(os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim", "Ambigram_hostsim"), oracle, str(tmp_path))


@pytest.mark.gpu
def test_sc_bfb_cli_gpu(hip_lib, oracle, tmp_path):
    check_sc_cli(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ambigram_amd", "bin", "Ambigram"), oracle, str(tmp_path))
