"""The order table's rows hold 5 bits per node for units of up to 32 nodes and 6 bits up to 63 (csrc/ambi_orders.hpp: RowBits /
row_node; a field may straddle two dwords).  Every node count 1..63 -- every row width 1..12 dwords, every straddling pattern --
through the block emission, the copy of the first rows and ambi_batch_unit_orders' unpacking: the oracle's orders, node by node."""
import pytest

from ambigram_amd import api, synth


def _check_every_node_count(lib, oracle, workdir, tag):
    seen = set()
    for K in range(1, 64):
        for tier in ("chain", "wide", "mixed", "skew"):
            if (tier == "wide" and K > 15) or (tier == "mixed" and K > 13) or (tier == "skew" and K < 21):      # (order counts grow like C(K, K/2): kept small)
                continue
            try:
                s = synth.make_sample(160 if K > 32 else 80, 330 if K > 32 else 170, tier, K, seed=500 + K)
            except Exception:
                continue
            lh, sols = s.write(workdir, "rp_%s_%d%s" % (tag, K, tier))
            if oracle.run_bfb(lh, sols, keep_orders=False)["chr"][0]["num_orders"] > 40000:
                continue
            o = oracle.run_bfb(lh, sols, keep_orders=True)["chr"][0]
            g = api.Graph(lib, lh)
            b = api.Batch(lib)
            b.add_chromosome_sol(g, 0, sols[0])
            b.upload(); b.run(0); b.download()
            r = b.unit_result(0)
            assert r["n_nodes"] == len(o["node2pat"]) and r["num_orders"] == o["num_orders"], (K, tier)
            if r["num_orders"] > 0:
                assert b.unit_orders(0, 0, r["num_orders"], r["n_nodes"]).tolist() == o["orders"], (K, tier)
            seen.add(r["n_nodes"])
            b.close(); g.close()
    assert seen >= set(range(1, 64)), sorted(set(range(1, 64)) - seen)


def test_row_packing_every_node_count_host(hostsim_lib, oracle, workdir):
    _check_every_node_count(hostsim_lib, oracle, workdir, "cpu")


@pytest.mark.gpu
def test_row_packing_every_node_count_gpu(hip_lib, oracle, workdir):
    _check_every_node_count(hip_lib, oracle, workdir, "gpu")
