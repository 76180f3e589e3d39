"""ambi_batch_run_sharded: the C-level multi-device driver (one host thread per device, units dealt round-robin, results merged
on the host).  N shares on ONE device (the way to exercise the N > 1 path on a one-GPU box) and on the host
simulation; the merged results must equal a single-device run and the oracle, for every getter."""
import numpy as np
import pytest

from ambigram_amd import api, synth


def _batch(lib, workdir, tag, count):
    gs, samples = [], []
    b1, bn = api.Batch(lib), api.Batch(lib)
    for i in range(count):
        s = synth.make_sample((40, 64, 96)[i % 3], (90, 128, 200)[i % 3], ("chain", "wide", "mixed")[i % 3], (7, 9, 11)[i % 3], seed=4100 + i,
                              imperfect=i % 2, n_del=i % 3, n_dup=(i + 1) % 2, near_inv=i % 4, name="%s%d" % (tag, i))
        lh, sols = s.write(workdir)
        g = api.Graph(lib, lh)
        gs.append(g); samples.append((lh, sols))
        b1.add_chromosome_sol(g, 0, sols[0]); bn.add_chromosome_sol(g, 0, sols[0])
    return gs, samples, b1, bn


def _check(lib, oracle, workdir, tag, devices, count=13, flags=0):
    gs, samples, b1, bn = _batch(lib, workdir, tag, count)
    b1.upload(); b1.run(flags); b1.download()
    bn.run_sharded(flags, devices=devices)
    for u in range(count):
        r1, rn = b1.unit_result(u), bn.unit_result(u)
        assert r1 == rn, (u, r1, rn)
        for which in (0, 1):
            assert b1.unit_path(u, which).tolist() == bn.unit_path(u, which).tolist(), u
        assert b1.unit_bkp(u).tolist() == bn.unit_bkp(u).tolist() and b1.unit_out_juncs(u) == bn.unit_out_juncs(u)
        n = gs[u].n_seg
        p1, pn = b1.unit_prepare(u, n), bn.unit_prepare(u, n)
        for k in p1:
            assert np.array_equal(p1[k], pn[k]), (u, k)
        if r1["status"] == 0:
            K = r1["n_nodes"]
            for a, c in zip(b1.unit_dag(u, K), bn.unit_dag(u, K)):
                assert np.array_equal(a, c), u
            assert np.array_equal(b1.unit_orders(u, 0, r1["num_orders"], K), bn.unit_orders(u, 0, r1["num_orders"], K)), u
            if flags & api.FLAG_ALL:
                for ps in (0, 1):
                    assert b1.all_orders(u, ps).tolist() == bn.all_orders(u, ps).tolist(), (u, ps)
                idx = bn.all_orders(u, 0)
                if len(idx):
                    assert [p.tolist() for p in b1.all_paths(u, 0, 0, min(4, len(idx)), 4096)] == [p.tolist() for p in bn.all_paths(u, 0, 0, min(4, len(idx)), 4096)]
    for u in (0, count // 2, count - 1):
        oc = oracle.run_bfb(*samples[u])["chr"][0]
        assert bn.unit_path(u, 1).tolist() == oc["path_indel"], u
    # a second call reuses the resident shares
    bn.run_sharded(flags, devices=devices)
    assert bn.unit_path(count - 1, 1).tolist() == b1.unit_path(count - 1, 1).tolist()
    with pytest.raises(api.AmbiError):
        bn.upload()                      # a batch is either uploaded to one device or sharded
    b1.close(); bn.close()


@pytest.mark.parametrize("shares", [1, 2, 3, 5])
def test_sharded_on_the_host_simulation(hostsim_lib, oracle, workdir, shares):
    _check(hostsim_lib, oracle, workdir, "shh%d_" % shares, [0] * shares)


def test_sharded_all_mode_on_the_host_simulation(hostsim_lib, oracle, workdir):
    _check(hostsim_lib, oracle, workdir, "shha", [0, 0, 0], count=7, flags=api.FLAG_ALL)


def test_sharded_needs_a_device_list_without_devices(hostsim_lib, workdir):
    gs, samples, b1, bn = _batch(hostsim_lib, workdir, "shnd", 2)
    with pytest.raises(api.AmbiError) as e:
        bn.run_sharded(0)                # the host simulation reports no device: "all visible devices" is nothing
    assert e.value.code == -30
    b1.close(); bn.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shares", [2, 4])
def test_sharded_n_shares_on_one_gpu(hip_lib, oracle, workdir, shares):
    _check(hip_lib, oracle, workdir, "shg%d_" % shares, [0] * shares, count=21)


@pytest.mark.gpu
def test_sharded_all_visible_devices_and_all_mode(hip_lib, oracle, workdir):
    _check(hip_lib, oracle, workdir, "shgv", None, count=9)
    _check(hip_lib, oracle, workdir, "shga", [0, 0, 0], count=7, flags=api.FLAG_ALL)
