"""GPU parity tests: the HIP engine (libambigram_hip.so, through the C ABI) against the CPU oracle on the same
seeded inputs, plus size-independent properties at BASELINE.json's full sizes.  Bit-exact: integer/index work."""
import numpy as np
import pytest

import engine_checks as ec
from ambigram_amd import api, synth

pytestmark = pytest.mark.gpu


def test_backend_is_hip(hip_lib):
    assert hip_lib.ambi_backend_name() == b"hip"


def test_fixed_and_synthetic(hip_lib, oracle, workdir):
    ec.check_fixed_and_synthetic(hip_lib, oracle, workdir, small_only=False)


def test_search_budget(hip_lib, oracle, workdir):
    ec.check_search_budget(hip_lib, oracle, workdir)


def test_random_decompositions(hip_lib, oracle, workdir):
    st = ec.check_random_decompositions(hip_lib, oracle, workdir, range(120), budget=2)
    assert st["valid"] > 10 and st["none"] > 10, st


def test_random_decompositions_default_budget(hip_lib, oracle, workdir):
    ec.check_random_decompositions(hip_lib, oracle, workdir, range(200, 260))


def test_injected_validity(hip_lib, oracle, workdir):
    """first_valid > 0 / minimum index over concurrent chunks / error-before-valid / orientation flip / --all bitmaps,
    with injected verdicts (no known input has mixed validity: DESIGN.md section 2)."""
    ec.check_injected_validity(hip_lib, oracle, workdir)


def test_edge_cases(hip_lib, oracle, workdir):
    ec.check_edge_cases(hip_lib, oracle, workdir)


def test_juncs_file(hip_lib, oracle, workdir):
    ec.check_juncs_file(hip_lib, oracle, workdir)


def test_batch_many_units(hip_lib, oracle, workdir):
    ec.check_batch_many_units(hip_lib, oracle, workdir, 48)


def test_enumerate_variants(hip_lib, oracle, workdir):
    ec.check_enumerate_variants(hip_lib, oracle, workdir, big=True)


def test_config2_full_size_properties(hip_lib, oracle, workdir):
    """BASELINE config 2 (256 seg / 512 junc), wide tier K=19: R = C(18,9) = 48 620 orders.
    Properties that do not need the oracle at full size + one full oracle comparison."""
    s = synth.make_sample(256, 512, "wide", 19, seed=2000)
    lh, sols = s.write(workdir, "c2")
    g = api.Graph(hip_lib, lh)
    b = api.Batch(hip_lib)
    b.add_chromosome_sol(g, 0, sols[0])
    b.upload(); b.run(0); b.download()
    r = b.unit_result(0)
    assert r["status"] == 0 and r["num_orders"] == 48620 and r["n_nodes"] == 19
    K = 19
    orders = b.unit_orders(0, 0, r["num_orders"], K)
    # every row is a permutation of 0..K-1
    assert np.array_equal(np.sort(orders, axis=1), np.tile(np.arange(K, dtype=np.uint8), (len(orders), 1)))
    # rows are strictly increasing in lexicographic order (the reference's DFS emits them that way)
    a, bb = orders[:-1].astype(np.int16), orders[1:].astype(np.int16)
    neq = a != bb
    first = neq.argmax(axis=1)
    assert neq.any(axis=1).all()
    idx = np.arange(len(a))
    assert (a[idx, first] < bb[idx, first]).all()
    # every row respects the DAG (predecessors first)
    pat, loop, succ = b.unit_dag(0, K)
    pos = np.argsort(orders, axis=1)
    for i in range(K):
        for j in range(K):
            if (int(succ[i]) >> j) & 1:
                assert (pos[:, i] < pos[:, j]).all()
    # targetCN equals the planted copy numbers (localhap.cpp:222-232); the assembled path starts at the telomere
    path = b.unit_path(0, 0)
    prep = b.unit_prepare(0, 256)
    cn = g.segments()["cn"]
    assert np.array_equal(prep["target_cn"][1:], cn.astype(np.int32))
    assert path[0] == 1 and len(path) == r["path_len"]
    # consecutive path vertices are reference adjacencies or fold-backs on one segment (perfect FBIs here)
    d = np.abs(np.abs(path[1:]) - np.abs(path[:-1]))
    same_strand = (path[1:] > 0) == (path[:-1] > 0)
    assert ((d == 1) & same_strand | (d == 0) & ~same_strand).all()
    # and the whole thing equals the oracle
    oc = oracle.run_bfb(lh, sols, keep_orders=True)["chr"][0]
    assert orders.tolist() == oc["orders"]
    assert path.tolist() == oc["path"] and b.unit_path(0, 1).tolist() == oc["path_indel"]


def test_config3_batch_1024x64(hip_lib, oracle, workdir):
    """BASELINE config 3: 1024 independent 64-seg samples in one batch, every one of them against the oracle."""
    graphs, b, samples = [], api.Batch(hip_lib), []
    for i in range(1024):
        s = synth.config_sample(3, i, tier=("chain", "wide", "mixed")[i % 3], K=(9, 9, 7)[i % 3])
        lh, sols = s.write(workdir, "c3_%d" % i)
        g = api.Graph(hip_lib, lh)
        graphs.append(g)
        b.add_chromosome_sol(g, 0, sols[0])
        samples.append((lh, sols))
    b.upload(); b.run(0); b.download()
    for i in range(1024):
        r = b.unit_result(i)
        assert r["status"] == 0, (i, r)
        p = b.unit_path(i, 0)
        d = np.abs(np.abs(p[1:]) - np.abs(p[:-1]))
        same = (p[1:] > 0) == (p[:-1] > 0)
        assert len(p) > 64 and ((d == 1) & same | (d <= 2) & ~same).all(), i   # adjacency or fold-back (<= 2 apart)
    for i in range(1024):      # every sample against the oracle (it does a few hundred of these per second)
        o = oracle.run_bfb(*samples[i])
        oc = o["chr"][0]
        assert b.unit_path(i, 0).tolist() == oc["path"] and b.unit_path(i, 1).tolist() == oc["path_indel"], i
        assert b.unit_bkp(i).tolist() == oc["bkp"], i
        assert b.unit_out_juncs(i) == [tuple(x) for x in o["out_juncs"]], i
        r = b.unit_result(i)
        assert (r["num_orders"], r["first_valid"], r["bias"]) == (oc["num_orders"], oc["first_valid"], oc["bias"]), i


def test_config4_multichr_1024(hip_lib, oracle, workdir):
    """BASELINE config 4: 1024 seg / 2048 junc, 8 chromosomes, translocation + PROP C2 (BFB-TRX)."""
    import parity
    s = synth.config_sample(4, 0, tier="chain", K=7)
    lh, sols = s.write(workdir, "c4")
    assert parity.compare(hip_lib, oracle, lh, sols) == []


def test_pack_paths_matches_download(hip_lib, oracle, workdir):
    import torch
    b, graphs, expect = ec.check_batch_many_units(hip_lib, oracle, workdir, 9)
    n = b.size()
    lengths = torch.zeros(n, dtype=torch.int32, device="cuda")
    cap = sum(len(e["path_indel"]) for e in expect) + 8
    cells = torch.zeros(cap, dtype=torch.int32, device="cuda")
    total = torch.zeros(1, dtype=torch.int64, device="cuda")
    b.pack_paths(1, lengths.data_ptr(), cells.data_ptr(), cap, total.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert lengths.cpu().tolist() == [len(e["path_indel"]) for e in expect]
    flat = [v for e in expect for v in e["path_indel"]]
    assert int(total.item()) == len(flat)
    assert cells.cpu().tolist()[:len(flat)] == flat


def test_cli_on_gpu(hip_lib, oracle, tmp_path):
    """The drop-in CLI binary (ambigram_amd/bin/Ambigram, linked against libambigram_hip.so) on the GPU."""
    import os
    import test_cli_dropin as t
    exe = os.path.join(t.ROOT, "ambigram_amd", "bin", "Ambigram")
    assert os.path.exists(exe), "build the CLI first (__graft_entry__.build)"
    d1, d2, d3, d4, d5 = (tmp_path / x for x in "abcde")
    for d in (d1, d2, d3, d4, d5):
        d.mkdir()
    t.check_readme(exe, str(d1), oracle)
    t.check_trx(exe, str(d2), oracle)
    d2b = tmp_path / "b2"
    d2b.mkdir()
    t.check_trx_before(exe, str(d2b), oracle)                    # PROP I1 / C1 (README.md:134, :154-157)
    t.check_errors(exe, str(d3))
    t.test_cli_many_chromosomes_one_batch(exe, oracle, d4)       # 8 chromosomes: one probe batch + one reconstruct batch
    t.test_cli_with_a_solver_that_reads_the_lp(exe, d5)          # the written .lp solved for real (HiGHS stand-in for cbc)
    d6, d7 = tmp_path / "f", tmp_path / "g"
    d6.mkdir(); d7.mkdir()
    t.check_sample_list_over_devices(exe, str(d6), oracle, "0,0,0")   # several samples, one batch, three shares on the one GPU
    t.check_sample_list_over_devices(exe, str(d7), oracle, "all")     # ... and over every visible device


def test_large_lattice(hip_lib, oracle, workdir):
    ec.check_large_lattice(hip_lib, oracle, workdir, K=50, k2=5)


def test_all_mode(hip_lib, oracle, workdir):
    st = ec.check_all_mode(hip_lib, oracle, workdir, seeds=range(60))
    assert st["multi"] > 0, st


def test_all_mode_two_forms(hip_lib, workdir):
    ec.check_all_two_forms(hip_lib, workdir)


def test_large_batch_of_pending_units(hip_lib, oracle, workdir):
    """~1500 units in one launch, scan budget 1: the plan kernel (one workgroup looping over the units) runs beside the
    scan kernel, whose units end PENDING / NO_VALID_ORDER all the time."""
    st = ec.check_large_batch_of_pending_units(hip_lib, oracle, workdir, seeds=range(20000, 21500))
    assert st["units"] >= 1000 and st["none_pending"] >= 500, st


def test_arena_limit_refuses_the_units_beyond_it(hip_lib, workdir):
    """ORDERS_CAPACITY with the scan for the first valid order running beside the plan kernel (ordinary chain, 40 units) and
    through the express chain (6 units)."""
    assert ec.check_arena_limit(hip_lib, workdir, n_units=40) > 0
    assert ec.check_arena_limit(hip_lib, workdir, n_units=6, seeds=range(9400, 9406)) > 0


def test_mixed_batch(hip_lib, oracle, workdir):
    ec.check_mixed_batch(hip_lib, oracle, workdir, big=True)


@pytest.mark.gpu
def test_full_finish_stage_on_every_unit(hip_lib, oracle, workdir, monkeypatch):
    """The full finish stage (path cells in LDS) on every unit instead of the lean one (runs only): same results."""
    monkeypatch.setenv("AMBI_LEAN_FINISH", "0")
    ec.check_fixed_and_synthetic(hip_lib, oracle, workdir, small_only=True)
    ec.check_random_decompositions(hip_lib, oracle, workdir, range(200, 260), budget=2)
    ec.check_mixed_batch(hip_lib, oracle, workdir, big=True)


def test_resident_batch_run_again(hip_lib, oracle, workdir):
    """The batch stays resident and is run again (what the first run taught the engine -- which units take the general
    enumerate path, how large the order arena has to be -- is reused): same tables and paths every time."""
    items = []
    for i, (tier, K, nseg, njunc) in enumerate([("wide", 13, 64, 128), ("wide", 17, 128, 256), ("skew", 27, 64, 128), ("chain", 11, 64, 128), ("wide", 15, 96, 200)]):
        s = synth.make_sample(nseg, njunc, tier, K, seed=9100 + i)
        lh, sols = s.write(workdir, "sr%d" % i)
        items.append((lh, sols[0]))
    graphs, b = [], api.Batch(hip_lib)
    for lh, sol in items:
        g = api.Graph(hip_lib, lh)
        graphs.append(g)
        b.add_chromosome_sol(g, 0, sol)
    b.upload()
    snaps = []
    for rep in range(3):
        b.run(0); b.wait(); b.download()
        snap = []
        for u in range(len(items)):
            r = b.unit_result(u)
            assert r["status"] == 0, (rep, u, r)
            snap.append((r["num_orders"], b.unit_orders(u, 0, r["num_orders"], r["n_nodes"]).tolist(), b.unit_path(u, 0).tolist(),
                         b.unit_path(u, 1).tolist(), b.unit_bkp(u).tolist()))
        snaps.append(snap)
    assert snaps[0] == snaps[1] == snaps[2]
    for u, (lh, sol) in enumerate(items):
        o = oracle.run_bfb(lh, [sol], keep_orders=True)["chr"][0]
        assert snaps[2][u][0] == o["num_orders"] and snaps[2][u][1] == o["orders"] and snaps[2][u][2] == o["path"]
    b.close()
    for g in graphs:
        g.close()


@pytest.mark.gpu
def test_resident_large_batch_run_again(hip_lib, oracle, workdir, monkeypatch):
    """More units than the express path takes (two groups of 64), run again and again with the launch parameters the first
    run taught the engine: the oracle's order tables and paths every time."""
    specs = [("wide", 13, 64, 128), ("chain", 9, 40, 80), ("wide", 17, 96, 200), ("mixed", 11, 64, 128), ("wide", 15, 64, 128), ("chain", 5, 24, 48)]
    items = []
    for i in range(70):     # (two groups of 64 units: the second group's offsets start from the first group's sum)
        tier, K, nseg, njunc = specs[i % len(specs)]
        s = synth.make_sample(nseg, njunc, tier, K, seed=9300 + i, imperfect=i % 2, n_del=i % 3)
        lh, sols = s.write(workdir, "sl%d" % i)
        items.append((lh, sols[0]))
    want = [oracle.run_bfb(lh, [sol], keep_orders=True)["chr"][0] for lh, sol in items]
    for mode in ("resident",):
        graphs, b = [], api.Batch(hip_lib)
        for lh, sol in items:
            g = api.Graph(hip_lib, lh)
            graphs.append(g)
            b.add_chromosome_sol(g, 0, sol)
        b.upload()
        for rep in range(4):
            b.run(0); b.wait(); b.download()
            for u, o in enumerate(want):
                r = b.unit_result(u)
                assert r["num_orders"] == o["num_orders"], (mode, rep, u)
                if o["first_valid"] < 0:
                    continue
                assert r["status"] == 0, (mode, rep, u, r)
                assert b.unit_orders(u, 0, r["num_orders"], r["n_nodes"]).tolist() == o["orders"], (mode, rep, u)
                assert b.unit_path(u, 1).tolist() == o["path_indel"], (mode, rep, u)
        b.close()
        for g in graphs:
            g.close()


@pytest.mark.gpu
def test_run_length_exchange_payload(hip_lib, oracle, workdir):
    """pack_runs -> expand_runs on the GPU reproduces the downloaded paths (the payload of bench.py's gather), for both
    path kinds, with far fewer runs than cells."""
    import torch
    from ambigram_amd.dist import RunExchange
    items = []
    for i in range(12):
        s = synth.make_sample(64, 128, ("chain", "wide", "mixed")[i % 3], 9, seed=9300 + i, n_del=i % 3, imperfect=i % 2)
        lh, sols = s.write(workdir, "rl%d" % i)
        items.append((lh, sols[0]))
    graphs, b = [], api.Batch(hip_lib)
    for lh, sol in items:
        g = api.Graph(hip_lib, lh)
        graphs.append(g)
        b.add_chromosome_sol(g, 0, sol)
    b.upload(); b.run(0); b.wait(); b.download()
    U = len(items)
    for which in (0, 1):
        n_runs, n_cells = RunExchange.probe(b, U, "cuda", which)
        want = [b.unit_path(u, which).tolist() for u in range(U)]
        assert n_cells == sum(len(w) for w in want) and 0 < n_runs <= max(n_cells, 1)
        rx = RunExchange(hip_lib, U, n_runs, n_cells, "cuda", world=1, rank=0)
        rx.pack(b, which)
        rx.exchange()
        rx.expand()
        torch.cuda.synchronize()
        assert rx.collect()[0] == want
        assert n_runs * 8 < n_cells, (n_runs, n_cells)      # the point of the run-length form
    b.close()
    for g in graphs:
        g.close()


def test_max_sizes(hip_lib, oracle, workdir):
    ec.check_max_sizes(hip_lib, oracle, workdir)


def test_two_batches_of_different_sizes_alive(hip_lib, oracle, workdir):
    """Two resident batches at once -- a large-unit one uploaded first, a small-unit one second -- run alternately on two
    streams (what bench.py's `pipelined` leg does): per-kernel launch attributes are process-wide, the second upload must
    not take from the first what its launches need."""
    import torch
    specs = [[("wide", 19, 256, 512), ("wide", 17, 128, 256)], [("chain", 7, 40, 80), ("wide", 9, 40, 80), ("mixed", 8, 48, 100)]]
    batches, graphs, expect = [], [], []
    for k, group in enumerate(specs):
        b = api.Batch(hip_lib)
        exp = []
        for i, (tier, K, nseg, njunc) in enumerate(group):
            s = synth.make_sample(nseg, njunc, tier, K, seed=9500 + 10 * k + i)
            lh, sols = s.write(workdir, "tb%d_%d" % (k, i))
            g = api.Graph(hip_lib, lh)
            graphs.append(g)
            b.add_chromosome_sol(g, 0, sols[0])
            exp.append(oracle.run_bfb(lh, sols, keep_orders=True)["chr"][0])
        b.upload()
        batches.append(b)
        expect.append(exp)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for rep in range(3):
        for k in (0, 1, 0, 1):
            batches[k].run(0, streams[k].cuda_stream)
        for k in (0, 1):
            batches[k].wait(); batches[k].download()
            for u, oc in enumerate(expect[k]):
                r = batches[k].unit_result(u)
                assert r["status"] == 0 and r["num_orders"] == oc["num_orders"], (rep, k, u, r)
                assert batches[k].unit_orders(u, 0, r["num_orders"], r["n_nodes"]).tolist() == oc["orders"], (rep, k, u)
                assert batches[k].unit_path(u, 1).tolist() == oc["path_indel"], (rep, k, u)
    for b in batches:
        b.close()
    for g in graphs:
        g.close()


def test_express_path_equals_kernel_chain(hip_lib, oracle, workdir, monkeypatch):
    """Small batches take the express path (one kernel reconstructs every unit whose first order assembles, the lattice /
    order table follow behind; ambi_batch_wait_results returns before they are done).  Same results, same order tables as
    the ordinary kernel chain (AMBI_EXPRESS_UNITS=0), on every kind of unit."""
    import cases
    items = [(lh, sols[0]) for _, lh, sols in cases.synthetic_cases(workdir, small_only=True) if len(sols) == 1][:14]
    for seed in (3, 5, 8, 13):
        lh, sols = cases.random_decomposition(workdir, 1700 + seed)
        items.append((lh, sols[0]))

    def run(express, flags):
        monkeypatch.setenv("AMBI_EXPRESS_UNITS", "64" if express else "0")
        graphs, b = [], api.Batch(hip_lib)
        for lh, sol in items:
            g = api.Graph(hip_lib, lh); graphs.append(g)
            b.add_chromosome_sol(g, 0, sol)
        b.upload()
        b.run(flags); b.wait()                  # first run sizes the arena (ordinary chain)
        b.run(flags); b.wait_results(); b.wait(); b.download()
        out = []
        for u in range(len(items)):
            r = b.unit_result(u)
            rec = dict(r)
            if r["status"] == 0:
                rec["path"] = b.unit_path(u, 0).tolist(); rec["path_indel"] = b.unit_path(u, 1).tolist(); rec["bkp"] = b.unit_bkp(u).tolist()
                rec["out"] = b.unit_out_juncs(u)
                rec["orders"] = b.unit_orders(u, 0, min(r["num_orders"], 3000), r["n_nodes"]).tolist()
            out.append(rec)
        b.close()
        for g in graphs:
            g.close()
        return out

    for flags in (0, api.FLAG_REVERSED):
        a, c = run(True, flags), run(False, flags)
        for u, (x, y) in enumerate(zip(a, c)):
            x.pop("reserved", None); y.pop("reserved", None)
            assert x == y, (flags, u, items[u][0])


@pytest.mark.gpu
def test_direct_full_finish_with_a_small_path_area(hip_lib, oracle, workdir, monkeypatch):
    """Units with deletion / duplication candidates get a full-stage launch of their own.  Its path area can be made smaller
    than the capacity bound of the batch (AMBI_DIRECT_CELLS: n cells, -1: sized from the paths of the batch's first run); a
    path that does not fit then goes through the list kernel behind it, which has the full area.  Forced here with areas
    far too small and just too small for the paths: same results as the launch with the whole area and as the run without
    the direct launch."""
    from ambigram_amd import synth
    items = []
    for i in range(48):   # more than the express path takes
        s = synth.make_sample(96, 192, "wide" if i % 3 else "chain", 9, seed=9100 + i, n_del=2 if i % 2 else 0, n_dup=1 if i % 4 == 1 else 0)
        lh, sols = s.write(workdir, "dfa%d" % i)
        items.append((lh, sols[0]))

    def run(cells, direct="1", flags=0):
        monkeypatch.setenv("AMBI_DIRECT_FULL", direct)
        if cells is None: monkeypatch.delenv("AMBI_DIRECT_CELLS", raising=False)
        else: monkeypatch.setenv("AMBI_DIRECT_CELLS", str(cells))
        graphs, b = [], api.Batch(hip_lib)
        for lh, sol in items:
            g = api.Graph(hip_lib, lh); graphs.append(g)
            b.add_chromosome_sol(g, 0, sol)
        b.upload()
        out = []
        for rep in range(3):                    # run 1 sizes the arena, run 2 the path area, run 3 uses it
            b.run(flags); b.wait()
        b.download()
        for u in range(len(items)):
            r = dict(b.unit_result(u))
            if r["status"] == 0:
                r["path"] = b.unit_path(u, 0).tolist(); r["path_indel"] = b.unit_path(u, 1).tolist(); r["out"] = b.unit_out_juncs(u)
            out.append(r)
        b.close()
        for g in graphs:
            g.close()
        return out

    for flags in (0, api.FLAG_REVERSED):
        ref = run(None, direct="0", flags=flags)
        assert sum(1 for r in ref if r["status"] == 0 and r["path"] != r["path_indel"]) >= 3, "the batch has no path-editing units"
        longest = max(len(r["path"]) for r in ref if r["status"] == 0)
        for cells in (None, -1, 64, longest - 8, longest + 8):
            got = run(cells, flags=flags)
            for u, (x, y) in enumerate(zip(got, ref)):
                assert x == y, (flags, cells, u, items[u][0])
