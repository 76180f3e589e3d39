"""Multi-process path (world_size 2, gloo, CPU): sample sharding + the single end-of-batch gather of
ambigram_amd/dist.py, with the engine's stage code running on the host simulation.  The GPU run uses the same
PathExchange over RCCL (bench.py)."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import json, os, sys
    sys.path.insert(0, %(root)r)
    import torch
    import torch.distributed as dist
    from ambigram_amd import api, synth
    from ambigram_amd.dist import PathExchange, RunExchange, shard

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    lib = api.load(%(lib)r)
    tmp = %(tmp)r
    N = 10                                    # samples of the whole job
    mine = shard(N, rank, world)
    batch, graphs = api.Batch(lib), []
    for i in mine:
        s = synth.make_sample(40, 80, ("chain", "wide", "mixed")[i %% 3], 7, seed=3000 + i, n_del=i %% 2)
        lh, sols = s.write(tmp, "d%%d" %% i)
        g = api.Graph(lib, lh); graphs.append(g)
        batch.add_chromosome_sol(g, 0, sols[0])
    batch.upload(); batch.run(0); batch.download()
    n_units = batch.size()
    assert n_units == len(mine)
    cells_needed = sum(batch.unit_result(u)["path_indel_len"] for u in range(n_units))
    got = None
    if N %% world == 0:                       # the cell form needs the same unit count on every rank
        px = PathExchange(n_units, cells_needed, "cpu", world=world, rank=rank)
        batch.pack_paths(1, px.lengths.data_ptr(), px.cells.data_ptr(), px.cell_cap, px.total.data_ptr())
        px.exchange()
        got = px.collect()
    # the same gather in run-length form (what bench.py does): pack -> all_gather of the counts + gather of the runs ->
    # rank 0 expands every rank's runs
    n_runs, n_cells = RunExchange.probe(batch, n_units, "cpu")
    assert n_cells == cells_needed and 0 < n_runs <= n_cells
    rx = RunExchange(lib, n_units, n_runs, n_cells, "cpu", world=world, rank=rank)
    rx.pack(batch, 1)
    rx.exchange()
    rx.expand()
    got_runs = rx.collect()
    if rank == 0:
        assert len(got_runs) == world and all(len(p) == rx.unit_cap for p in got_runs)
        assert got is None or got_runs == got, "run-length exchange differs from the cell exchange"
        json.dump(got_runs, open(os.path.join(tmp, "gathered.json"), "w"))
        json.dump({"runs": [int(x) for x in rx.counts_all.view(world, 2, rx.unit_cap)[:, 1, :].sum(1)], "cells": [int(x) for x in rx.counts_all.view(world, 2, rx.unit_cap)[:, 0, :].sum(1)]},
                  open(os.path.join(tmp, "payload.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()
''')


@pytest.mark.parametrize("world", [2, 4])
def test_gather_matches_oracle(oracle, hostsim_lib, workdir, world):
    """world 2: 5 + 5 samples, cell form and run-length form agree; world 4: 3 + 3 + 2 + 2 samples -- ranks with fewer
    units than the agreed capacity send empty unit slots (the padding path of RunExchange)."""
    tmp = os.path.join(workdir, "dist%d" % world)
    os.makedirs(tmp, exist_ok=True)
    lib_path = os.path.join(ROOT, "tests", "hostsim", "libambigram_hostsim.so")
    script = os.path.join(tmp, "worker.py")
    with open(script, "w") as f:
        f.write(WORKER % dict(root=ROOT, lib=lib_path, tmp=tmp))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29533 + world), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, script], env=dict(env, RANK=str(r))) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    import json
    from ambigram_amd import synth
    from ambigram_amd.dist import shard
    got = json.load(open(os.path.join(tmp, "gathered.json")))
    assert len(got) == world
    for r in range(world):
        mine = shard(10, r, world)
        assert all(p == [] for p in got[r][len(mine):])          # unused unit slots of a rank are empty
        for k, i in enumerate(mine):
            s = synth.make_sample(40, 80, ("chain", "wide", "mixed")[i % 3], 7, seed=3000 + i, n_del=i % 2)
            lh, sols = s.write(tmp, "o%d" % i)
            want = oracle.run_bfb(lh, sols)["chr"][0]["path_indel"]
            assert got[r][k] == want, (r, i)


ALL_WORKER = textwrap.dedent('''
    import json, os, sys
    sys.path.insert(0, %(root)r)
    sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import torch
    import torch.distributed as dist
    import cases
    from ambigram_amd import api, synth
    from ambigram_amd.dist import all_mode_merge

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    lib = api.load(%(lib)r)
    tmp = os.path.join(%(tmp)r, "r%%d" %% rank)
    # EVERY rank holds the same units: wide samples (all orders valid) and random decompositions (few or none valid, flips)
    items = []
    for i, (tier, K) in enumerate([("wide", 11), ("wide", 13), ("mixed", 9)]):
        s = synth.make_sample(64, 128, tier, K, seed=6100 + i)
        lh, sols = s.write(tmp, "w%%d" %% i)
        items.append((lh, sols[0]))
    for seed in range(40):
        lh, sols = cases.random_decomposition(tmp, 2600 + seed)
        items.append((lh, sols[0]))

    def run(shard):
        graphs, b = [], api.Batch(lib)
        for lh, sol in items:
            g = api.Graph(lib, lh); graphs.append(g)
            b.add_chromosome_sol(g, 0, sol)
        if shard:
            b.all_set_shard(rank, world)
        b.upload(); b.run(api.FLAG_ALL); b.wait()
        if shard:
            all_mode_merge(b, "cpu")
        b.download()
        out = []
        for u in range(len(items)):
            r = b.unit_result(u)
            out.append([r["status"], r["evaluated"], b.all_orders(u, 0).tolist(), b.all_orders(u, 1).tolist()])
        b.close()
        return out

    sharded, alone = run(True), run(False)
    assert sharded == alone, [i for i, (x, y) in enumerate(zip(sharded, alone)) if x != y]
    assert sum(1 for x in alone if x[0] == 0 and len(x[2]) > 64) >= 2       # really several chunks per unit
    if rank == 0:
        json.dump(dict(units=len(items), valid=[len(x[2]) + len(x[3]) for x in alone]), open(os.path.join(%(tmp)r, "all.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()
''')


@pytest.mark.parametrize("world", [2, 3])
def test_all_mode_orders_dealt_over_ranks(hostsim_lib, workdir, world):
    """--all for wide samples with the 64-order chunks of every unit dealt over the ranks (order-level
    sharding): one all-reduce of the bitmaps, then every rank holds the same lists a single rank computes."""
    tmp = os.path.join(workdir, "distall%d" % world)
    os.makedirs(tmp, exist_ok=True)
    lib_path = os.path.join(ROOT, "tests", "hostsim", "libambigram_hostsim.so")
    script = os.path.join(tmp, "worker.py")
    with open(script, "w") as f:
        f.write(ALL_WORKER % dict(root=ROOT, lib=lib_path, tmp=tmp))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29563 + world), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, script], env=dict(env, RANK=str(r))) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
