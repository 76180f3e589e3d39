"""Stage-by-stage comparison of an engine run (ambigram_amd.api.reconstruct_sample) with the CPU oracle."""
import numpy as np

from ambigram_amd import api


def compare(lib, oracle, lh, sols, juncs="", reversed_=False, all_=False, keep_orders=True, **kw):
    """Returns a list of human-readable differences (empty list == parity)."""
    o = oracle.run_bfb(lh, sols, juncs=juncs, reversed_=reversed_, all_=all_, keep_orders=keep_orders)
    e = api.reconstruct_sample(lib, lh, sols, juncs=juncs, reversed_=reversed_, all_=all_, keep_orders=keep_orders, **kw)
    diffs = []
    if not o["ok"]:
        return ["oracle failed: " + o["err"]]
    if not e["ok"]:
        return ["engine failed: " + e["err"]]
    if o["log"] != e["log"]:
        for i, (a, b) in enumerate(zip(o["log"], e["log"])):
            if a != b:
                diffs.append("stdout line %d differs: oracle %r engine %r" % (i, a[:120], b[:120]))
                break
        else:
            diffs.append("stdout line count %d vs %d" % (len(o["log"]), len(e["log"])))
    for c, (oc, ec) in enumerate(zip(o["chr"], e["chr"])):
        if oc["shortcut"] != ec.get("shortcut") or oc["infeasible"] != ec.get("infeasible"):
            diffs.append("chr %d shortcut/infeasible flags differ" % c)
            continue
        if oc["bias"] != ec["bias"]:
            diffs.append("chr %d bias %s vs %s" % (c, oc["bias"], ec["bias"]))
        s = oc["start"]
        if not np.array_equal(np.array(oc["junc_cn"]).reshape(-1, 2)[s:], ec["junc_cn"][1:]):
            diffs.append("chr %d junc_cn differs" % c)
        if not np.array_equal(np.array(oc["seg_cn"])[s - 1:oc["end"]], ec["seg_cn"][1:]):
            diffs.append("chr %d seg_cn after getIndelBias differs" % c)
        inv_o = dict(zip(oc["inv_seg"], oc["inv_junc"]))
        inv_e = {s - 1 + i: int(j) for i, j in enumerate(ec["inv_junc"]) if i >= 1 and j >= 0}
        if inv_o != inv_e:
            diffs.append("chr %d fold-back map differs" % c)
        if ec.get("path_indel_from_runs") != ec["path_indel"]:
            diffs.append("chr %d: the final path in run-length form (runs_to_host) expands to %d cells, the path has %d" % (c, len(ec.get("path_indel_from_runs") or []), len(ec["path_indel"])))
        if oc["shortcut"] or oc["infeasible"]:
            if oc["path_indel"] != ec["path_indel"] and oc["path_indel"]:
                diffs.append("chr %d reference path differs" % c)
            continue
        for k in ["num_orders", "first_valid", "first_forward", "evaluated"]:
            if oc[k] != ec[k]:
                diffs.append("chr %d %s %s vs %s" % (c, k, oc[k], ec[k]))
        if oc["node2pat"] != [[] if r[0] == 0 else r for r in ec["node2pat"]]:
            diffs.append("chr %d node2pat differs" % c)
        if oc["node2loop"] != [[] if r[0] == 0 else r for r in ec["node2loop"]]:
            diffs.append("chr %d node2loop differs" % c)
        if [sum(1 << j for j in set(a)) for a in oc["adj"]] != ec["succ"]:
            diffs.append("chr %d DAG adjacency differs" % c)
        if keep_orders and oc["orders"] != ec["orders"]:
            diffs.append("chr %d order table differs (%d vs %d rows)" % (c, len(oc["orders"]), len(ec["orders"])))
        for k in ["bkp", "path", "path_indel"]:
            if oc[k] != ec[k]:
                diffs.append("chr %d %s differs (len %d vs %d)" % (c, k, len(oc[k]), len(ec[k])))
        if all_ and oc["all_paths"] != ec.get("all_paths"):
            diffs.append("chr %d --all paths differ (%d vs %d valid orders)" % (c, len(oc["all_paths"]), len(ec.get("all_paths") or [])))
        if oc["indel_printed"] != ec["indel_printed"]:
            diffs.append("chr %d indel_printed differs" % c)
    if [tuple(x) for x in o["out_juncs"]] != e["out_juncs"]:
        diffs.append("output junctions differ")
    if o.get("trx_before") or e.get("trx_before"):   # PROP I1 / C1: the paths after virusBFB, over the segments of the file
        if o.get("trx_before") != e.get("trx_before") or o["paths"] != e["paths"]:
            diffs.append("TRX-BFB: paths after virusBFB differ")
    if o["trx_path"] != e["trx_path"]:
        diffs.append("BFB-TRX path differs")
    return diffs
