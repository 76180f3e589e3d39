"""Generates tests/golden/graph_*.json from the REAL reference graph model (oracle/_ref/ref_graph_dump, built from
/root/reference/src/{Graph,Segment,Vertex,Edge,Junction,Weight,Exceptions}.cpp by `make -C oracle ref`).
Container-only: the GPU box has no /root/reference; it uses the committed JSON files.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_py  # noqa: E402
from ambigram_amd import synth  # noqa: E402

CASES = {
    "readme6": os.path.join(ROOT, "tests", "data", "readme6.lh"),
    "trx_c2": os.path.join(ROOT, "tests", "data", "trx_c2.lh"),
    "quirks": os.path.join(ROOT, "tests", "data", "quirks.lh"),
    "quirks2": os.path.join(ROOT, "tests", "data", "quirks2.lh"),     # CRLF, virus segments, recomputed AVG_PLOIDY, 3 chromosomes
}
# (.lh, .juncs): the graph after the graph-level effects of readComponents, driven through the reference's graph API
JUNCS_CASES = {
    "readme6__juncs": (os.path.join(ROOT, "tests", "data", "readme6.lh"), os.path.join(ROOT, "tests", "data", "readme6.juncs")),
    "quirks2__juncs": (os.path.join(ROOT, "tests", "data", "quirks2.lh"), os.path.join(ROOT, "tests", "data", "quirks2.juncs")),
}


def main():
    oracle_py.build(ref=True)
    out_dir = os.path.dirname(os.path.abspath(__file__))
    tmp = os.path.join(out_dir, "_tmp")
    os.makedirs(tmp, exist_ok=True)
    s = synth.make_sample(24, 48, "chain", 5, seed=7, imperfect=1, n_del=1, n_dup=1)
    lh, _ = s.write(tmp, "syn24")
    cases = dict(CASES)
    cases["syn24"] = lh
    with open(os.path.join(out_dir, "syn24.lh"), "w") as f:
        f.write(s.lh_text)
    for name, path in cases.items():
        # (the same run also writes the graph back through the reference's Graph::writeGraph: written_<name>.lh pins the product's writer)
        d = oracle_py.ref_graph_dump(path, write_to=os.path.join(out_dir, "written_%s.lh" % name))
        assert d and d["ok"], (name, d)
        d["log"] = [l for l in d["log"] if l != "write seg"]
        with open(os.path.join(out_dir, "graph_%s.json" % name), "w") as f:
            json.dump(d, f, indent=0, sort_keys=True)
        print("wrote graph_%s.json: %d segs, %d juncs" % (name, len(d["segs"]), len(d["juncs"])))
    for name, (lh, juncs) in JUNCS_CASES.items():
        d = oracle_py.ref_graph_dump(lh, juncs, write_to=os.path.join(out_dir, "written_%s.lh" % name))
        assert d and d["ok"], (name, d)
        d["log"] = [l for l in d["log"] if l != "write seg"]
        with open(os.path.join(out_dir, "graph_%s.json" % name), "w") as f:
            json.dump(d, f, indent=0, sort_keys=True)
        print("wrote graph_%s.json: %d segs, %d juncs" % (name, len(d["segs"]), len(d["juncs"])))


if __name__ == "__main__":
    main()
