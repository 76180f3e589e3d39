"""CPU-only: the engine's SPMD stage code (the source the HIP kernels instantiate), the packing, the result-blob
layout and the C ABI, executed on the 1-thread host group, against the oracle."""
import pytest

import engine_checks as ec


def test_backend_name(hostsim_lib):
    assert hostsim_lib.ambi_backend_name() == b"hostsim"


def test_fixed_and_synthetic(hostsim_lib, oracle, workdir):
    ec.check_fixed_and_synthetic(hostsim_lib, oracle, workdir, small_only=False)


def test_search_budget(hostsim_lib, oracle, workdir):
    ec.check_search_budget(hostsim_lib, oracle, workdir)


def test_random_decompositions(hostsim_lib, oracle, workdir):
    st = ec.check_random_decompositions(hostsim_lib, oracle, workdir, range(120), budget=2)
    assert st["valid"] > 10 and st["none"] > 10, st


def test_edge_cases(hostsim_lib, oracle, workdir):
    ec.check_edge_cases(hostsim_lib, oracle, workdir)


def test_juncs_file(hostsim_lib, oracle, workdir):
    ec.check_juncs_file(hostsim_lib, oracle, workdir)


def test_batch_many_units(hostsim_lib, oracle, workdir):
    ec.check_batch_many_units(hostsim_lib, oracle, workdir, 12)


def test_enumerate_variants(hostsim_lib, oracle, workdir):
    ec.check_enumerate_variants(hostsim_lib, oracle, workdir)


def test_large_lattice(hostsim_lib, oracle, workdir):
    ec.check_large_lattice(hostsim_lib, oracle, workdir, K=50, k2=5)


def test_all_mode(hostsim_lib, oracle, workdir):
    st = ec.check_all_mode(hostsim_lib, oracle, workdir)
    assert st["multi"] > 0, st


@pytest.mark.parametrize("search_order", ["asc", "desc", "shuffle"])
def test_injected_validity(hostsim_lib, oracle, workdir, monkeypatch, search_order):
    """first_valid > 0, minimum-index search, error-before-valid, orientation flip, --all bitmaps -- with injected
    verdicts; the chunks of the parallel search are taken in ascending, descending and shuffled order."""
    monkeypatch.setenv("AMBI_HOSTSIM_SEARCH_ORDER", search_order)
    ec.check_injected_validity(hostsim_lib, oracle, workdir)


def test_all_mode_two_forms(hostsim_lib, workdir):
    ec.check_all_two_forms(hostsim_lib, workdir)


def test_mixed_batch(hostsim_lib, oracle, workdir):
    ec.check_mixed_batch(hostsim_lib, oracle, workdir)


def test_large_batch_of_pending_units(hostsim_lib, oracle, workdir):
    st = ec.check_large_batch_of_pending_units(hostsim_lib, oracle, workdir, seeds=range(20000, 20150))
    assert st["units"] >= 100


def test_arena_limit_refuses_the_units_beyond_it(hostsim_lib, workdir):
    """ORDERS_CAPACITY: ordinary chain (40 units) and express chain (6 units)."""
    assert ec.check_arena_limit(hostsim_lib, workdir, n_units=40) > 0
    assert ec.check_arena_limit(hostsim_lib, workdir, n_units=6, seeds=range(9400, 9406)) > 0


def test_full_finish_stage_on_every_unit(hostsim_lib, oracle, workdir, monkeypatch):
    """By default the lean finish stage takes every unit and the full one only those it hands over; here the full stage
    (path cells in group memory, edits in place) runs on every unit, as it does when SVs chain or edit the path."""
    monkeypatch.setenv("AMBI_HOSTSIM_LEAN_FINISH", "0")
    ec.check_fixed_and_synthetic(hostsim_lib, oracle, workdir, small_only=True)
    ec.check_random_decompositions(hostsim_lib, oracle, workdir, range(200, 230), budget=2)
    ec.check_mixed_batch(hostsim_lib, oracle, workdir)


def test_max_sizes(hostsim_lib, oracle, workdir):
    ec.check_max_sizes(hostsim_lib, oracle, workdir)
