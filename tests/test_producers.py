"""Input-side producers: `.lh` from seg.txt + sv.txt, `.juncs` from 10x barcodes / optical mapping.
Golden vectors: tests/golden/producers.json, outputs of the reference's own scripts (tests/golden/make_producer_golden.py)."""
import json
import os
import subprocess
import sys

import pytest

from ambigram_amd import producers

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "producers.json")))


@pytest.mark.parametrize("case", GOLD["generate_lh"], ids=lambda c: "seg%d_sv%d" % (len(c["seg"]), len(c["sv"]) - 1))
def test_generate_lh_golden(case):
    assert producers.generate_lh(case["seg"], case["sv"], **case["kw"]) == case["lh"]


@pytest.mark.parametrize("case", GOLD["barcode_to_juncs"], ids=lambda c: "seg%d_bed%d" % (len(c["seg"]), len(c["bed"])))
def test_barcode_to_juncs_golden(case):
    if case.get("raises"):
        with pytest.raises(IndexError):
            producers.barcode_to_juncs(case["seg"], case["bed"])
    else:
        assert producers.barcode_to_juncs(case["seg"], case["bed"]) == case["juncs"]


@pytest.mark.parametrize("case", GOLD["om_to_juncs"], ids=lambda c: "om%d" % len(c["om"]))
def test_om_to_juncs_golden(case):
    assert producers.om_to_juncs(case["om"]) == case["juncs"]


def test_generated_lh_loads_in_the_reader(tmp_path):
    """Producer -> reader: the `.lh` text parses with the engine's reader (same segment / junction counts)."""
    from ambigram_amd import api
    lib = api.load()
    for k, case in enumerate(GOLD["generate_lh"]):
        if case["kw"].get("is_depth") or case["kw"].get("is_seg_depth") or case["kw"].get("is_sv_depth"):
            continue      # CN = -1 files go through calculateCopyNum with header-derived ratios; covered by the reader's own tests
        p = tmp_path / ("g%d.lh" % k)
        p.write_text(producers.generate_lh(case["seg"], case["sv"], **case["kw"]))
        g = api.Graph(lib, str(p))
        assert g.n_seg == len(case["seg"])
        n_junc_lines = sum(1 for l in case["lh"].splitlines() if l.startswith("JUNC"))
        assert 0 <= g.n_junc <= n_junc_lines      # the reader drops exact / complement duplicates (Graph.cpp:592-595)


@pytest.mark.skipif(not os.path.exists("/root/reference/script/bfb_scripts.py"), reason="reference scripts only exist in the build container")
def test_generate_lh_against_live_reference(tmp_path):
    segs = ["chr2:1000-2000\t2\n", "chr2:2001-2600\t4\n", "chr2:2601-4000\t2\n", "chr6:500-900\t1\n"]
    svs = ["h\n", "chr2\t2600\t+\tchr2\t2598\t-\t1\n", "chr2\t2001\t-\tchr2\t2003\t+\t1\n", "chr2\t4000\t+\tchr6\t500\t+\t1\n",
           "chr6\t501\t-\tchr2\t3999\t-\t2\n"]
    (tmp_path / "seg.txt").write_text("".join(segs))
    (tmp_path / "sv.txt").write_text("".join(svs))
    r = subprocess.run([sys.executable, "/root/reference/script/bfb_scripts.py", "generate_lh", "-sv", "sv.txt", "-seg", "seg.txt", "-s", "live"],
                       cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert producers.generate_lh(segs, svs) == (tmp_path / "live.lh").read_text()
