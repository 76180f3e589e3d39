"""indelBFB on the RUNS of the path (stage_finish_edit / indel_bfb_runs, csrc/ambi_finish.hpp) against indelBFB on the cells (stage_finish,
the statement-by-statement transcription of LocalGenomicMap.cpp:3746-3837 that rounds 1-3 pinned on the oracle): the same batch of SV-rich
units through both forms, every header field, both paths, the output junctions and the run-length form compared unit by unit -- on the host
simulation (AMBI_HOSTSIM_EDIT=0/1) and on the GPU (AMBI_DIRECT_EDIT=0/1: the direct launch as the edit kernel or as the full stage with
the cells in device memory).  The oracle comparison of the same stages is in test_hostsim_parity.py / test_gpu_parity.py."""
import os
import random

import numpy as np
import pytest

from ambigram_amd import api, synth


def _batch(lib, workdir, tag, n_units, seed0):
    rng = random.Random(seed0)
    b = api.Batch(lib)
    keep = []
    for i in range(n_units):
        big = (i % 3 == 0)
        tier = ["wide", "chain", "skew", "mixed"][i % 4]
        K = rng.choice([5, 7, 9, 11, 13]) if tier == "wide" else rng.choice([5, 8, 12, 17])
        s = synth.make_sample(256 if big else rng.choice([32, 64, 100]), 512 if big else rng.choice([64, 128, 200]), tier, K, seed=seed0 + i,
                              n_del=rng.randint(0, 8), n_dup=rng.randint(0, 8), near_inv=rng.randint(0, 5), imperfect=rng.randint(0, 2),
                              name="%s_%d" % (tag, i))
        lh, sols = s.write(workdir)
        g = api.Graph(lib, lh)
        keep.append(g)
        b.add_chromosome_sol(g, 0, sols[0])
    return b, keep


def _results(b, n_units):
    b.upload(); b.run(0); b.wait()
    b.runs_to_host(1, 0); v = b.runs_wait(0)
    runs = [b.runs_unit_path(0, u).tolist() for u in range(n_units)]
    b.download()
    out = []
    for u in range(n_units):
        out.append((b.unit_result(u), b.unit_path(u, 0).tolist(), b.unit_path(u, 1).tolist(), b.unit_out_juncs(u), runs[u]))
    return out


def check_edit_stage(lib, workdir, tag, env_name, n_units=120, seed0=8100):
    res = {}
    for on in ("1", "0"):
        os.environ[env_name] = on
        try:
            b, keep = _batch(lib, workdir, "%s%s" % (tag, on), n_units, seed0)
            res[on] = _results(b, n_units)
            b.close()
            for g in keep:
                g.close()
        finally:
            os.environ.pop(env_name, None)
    edited = sum(1 for r in res["1"] if r[1] != r[2])
    assert edited >= n_units // 4, edited                       # the batch really edits paths
    for u, (x, y) in enumerate(zip(res["1"], res["0"])):
        assert x == y, (u, x[0], y[0])
        assert x[4] == x[2]                                     # the run-length form expands to the final path
    return edited


def check_hand_over(lib, workdir, tag, env_name):
    """run lists limited to a few runs (AMBI_EDIT_RUN_CAP, a test hook): the edit stage hands most units on to the full stage -- on the GPU
    through the device-side list the ambi_finish_ext_kernel launch behind the edit kernel walks"""
    os.environ["AMBI_EDIT_RUN_CAP"] = "24"
    try:
        check_edit_stage(lib, workdir, tag, env_name, n_units=60, seed0=9100)
    finally:
        os.environ.pop("AMBI_EDIT_RUN_CAP", None)


def test_edit_stage_on_the_host_simulation(hostsim_lib, workdir):
    check_edit_stage(hostsim_lib, workdir, "eh", "AMBI_HOSTSIM_EDIT")
    check_hand_over(hostsim_lib, workdir, "ehh", "AMBI_HOSTSIM_EDIT")


@pytest.mark.gpu
def test_runs_that_outgrow_their_slots_on_the_gpu(hip_lib, workdir):
    """run slots limited to 8 runs per unit (AMBI_RUN_SLOTS, a test hook): the finish stages flag the units whose final path has more, and
    ambi_batch_runs_wait then serves the batch through the pack kernels -- the expanded runs still equal the final paths"""
    os.environ["AMBI_RUN_SLOTS"] = "8"
    try:
        b, keep = _batch(hip_lib, workdir, "ro", 80, 9700)
        b.upload(); b.run(0); b.wait()
        b.runs_to_host(1, 0); v = b.runs_wait(0)
        runs = [b.runs_unit_path(0, u).tolist() for u in range(80)]
        b.download()
        assert sum(len(r) > 0 for r in runs) == 80 and v["n_runs"] > 8 * 80 // 2     # really more runs than the slots held
        for u in range(80):
            assert runs[u] == b.unit_path(u, 1).tolist(), u
        b.close()
        for g in keep:
            g.close()
    finally:
        os.environ.pop("AMBI_RUN_SLOTS", None)


@pytest.mark.gpu
def test_edit_stage_on_the_gpu(hip_lib, workdir):
    check_edit_stage(hip_lib, workdir, "eg", "AMBI_DIRECT_EDIT", n_units=200, seed0=8600)
    check_hand_over(hip_lib, workdir, "egh", "AMBI_DIRECT_EDIT")
