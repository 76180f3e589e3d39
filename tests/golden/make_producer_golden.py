#!/usr/bin/env python3
"""Makes tests/golden/producers.json: inputs and outputs of the reference's OWN producer scripts
(script/bfb_scripts.py generate_lh / OM2juncs, script/process_barcode.py), run here as child processes on seeded random
inputs.  Container only (needs /root/reference); the JSON it writes is data -- inputs and expected outputs -- and is
what travels.

    python3 tests/golden/make_producer_golden.py
"""
import json
import os
import random
import subprocess
import sys
import tempfile

REF = "/root/reference/script"
HERE = os.path.dirname(os.path.abspath(__file__))


def run(cmd, cwd):
    r = subprocess.run([sys.executable] + cmd, cwd=cwd, capture_output=True, text=True)
    return r.returncode, r.stderr


def seg_file(rng, n_chr, per_chr, chr18=False):
    lines, names = [], ["chr%d" % (i + 1) for i in range(n_chr)]
    if chr18:
        names[-1] = "chr18"
    for name in names:
        pos = rng.randint(1000, 5000)
        for _ in range(per_chr):
            length = rng.randint(200, 3000)
            cn = rng.choice(["1", "2", "3", "4", "1.25", "2.5", "1.11", "6"])
            lines.append("%s:%d-%d\t%s\n" % (name, pos, pos + length, cn))
            pos += length + 1
    return lines


def sv_file(rng, segs, n_sv):
    spans = []
    for line in segs:
        name, cn = line.strip("\n").split("\t")
        chrom, iv = name.split(":")
        spans.append((chrom, int(iv.split("-")[0]), int(iv.split("-")[1])))
    lines = ["chrom_5p\tbkpos_5p\tstrand_5p\tchrom_3p\tbkpos_3p\tstrand_3p\tavg_cn\n"]
    for _ in range(n_sv):
        a, b = rng.choice(spans), rng.choice(spans)
        if rng.random() < 0.4:
            b = a                                                    # fold-back-like / duplicates
        pa = rng.choice([a[1], a[2]]) + rng.randint(-30, 30)
        pb = rng.choice([b[1], b[2]]) + rng.randint(-30, 30)
        chrom_b = b[0] if rng.random() > 0.05 else "chrUn"           # a chromosome without segments
        lines.append("%s\t%d\t%s\t%s\t%d\t%s\t%s\n" % (a[0], pa, rng.choice("+-"), chrom_b, pb, rng.choice("+-"),
                                                     rng.choice(["1", "2", "0.5", "1.5", "3"])))
    return lines


def main():
    rng = random.Random(20241004)
    cases = {"generate_lh": [], "barcode_to_juncs": [], "om_to_juncs": []}
    for k in range(14):
        d = tempfile.mkdtemp()
        segs = seg_file(rng, rng.randint(1, 3), rng.randint(2, 6), chr18=(k % 5 == 4))
        svs = sv_file(rng, segs, rng.randint(0, 14))
        open(os.path.join(d, "seg.txt"), "w").writelines(segs)
        open(os.path.join(d, "sv.txt"), "w").writelines(svs)
        kw, argv = {}, []
        if k % 3 == 1:
            kw["coverage"] = 45; argv += ["-c", "45"]
        if k % 4 == 2:
            kw["purity"] = 0.8; argv += ["-p", "0.8"]
        if k == 5:
            kw["is_depth"] = "True"; argv += ["-d", "True"]
        if k == 6:
            kw["is_seg_depth"] = "True"; argv += ["-d1", "True"]
        if k == 7:
            kw["is_sv_depth"] = "False"; argv += ["-d2", "False"]    # the string 'False' is not False either
        if k == 8:
            kw["prop"] = "PROP M:chr1 C2:chr1:chr2\n"; argv += ["-pr", kw["prop"]]
        rc, err = run([os.path.join(REF, "bfb_scripts.py"), "generate_lh", "-sv", "sv.txt", "-seg", "seg.txt", "-s", "out"] + argv, d)
        assert rc == 0, err
        cases["generate_lh"].append({"seg": segs, "sv": svs, "kw": kw, "lh": open(os.path.join(d, "out.lh")).read()})
    for k in range(8):
        d = tempfile.mkdtemp()
        segs = seg_file(rng, rng.randint(1, 2), rng.randint(4, 6))
        spans = [(l.split("\t")[0].split(":")[0], int(l.split(":")[1].split("-")[0]), int(l.split("\t")[0].split("-")[1])) for l in segs]
        bed = []
        for _ in range(rng.randint(20, 80)):
            a = rng.randrange(len(spans)); b = min(len(spans) - 1, a + rng.randint(0, 3))
            if spans[a][0] != spans[b][0]:
                b = a
            chrom = spans[a][0] if rng.random() < 0.7 else spans[a][0][3:]      # with and without the "chr" prefix
            bed.append("%s\t%d\t%d\tBX%03d\n" % (chrom, spans[a][1] + rng.randint(-50, 400), spans[b][2] + rng.randint(-400, 50), rng.randint(0, 25)))
        open(os.path.join(d, "seg.txt"), "w").writelines(segs)
        open(os.path.join(d, "bc.bed"), "w").writelines(bed)
        rc, err = run([os.path.join(REF, "process_barcode.py"), "-bed", "bc.bed", "-seg", "seg.txt", "-s", "out"], d)
        assert rc == 0, err
        cases["barcode_to_juncs"].append({"seg": segs, "bed": bed, "juncs": open(os.path.join(d, "out.juncs")).read()})
    # fewer than five links: the reference dies with an IndexError and writes nothing
    d = tempfile.mkdtemp()
    segs = ["chr1:100-200\t2\n", "chr1:201-300\t2\n", "chr1:301-400\t2\n"]
    bed = ["chr1\t100\t300\tBX001\n"]
    open(os.path.join(d, "seg.txt"), "w").writelines(segs)
    open(os.path.join(d, "bc.bed"), "w").writelines(bed)
    rc, err = run([os.path.join(REF, "process_barcode.py"), "-bed", "bc.bed", "-seg", "seg.txt", "-s", "out"], d)
    assert rc != 0 and "IndexError" in err and not os.path.exists(os.path.join(d, "out.juncs"))
    cases["barcode_to_juncs"].append({"seg": segs, "bed": bed, "juncs": None, "raises": "IndexError"})
    for k in range(4):
        d = tempfile.mkdtemp()
        lines = ["# SegAligner\n"] + ["%s%d\t%d\t0.9\n" % (rng.choice(["", "-"]), rng.randint(1, 12), i) for i in range(rng.randint(1, 9))]
        open(os.path.join(d, "om.txt"), "w").writelines(lines)
        rc, err = run([os.path.join(REF, "bfb_scripts.py"), "OM2juncs", "-i", "om.txt", "-p", "out"], d)
        assert rc == 0, err
        cases["om_to_juncs"].append({"om": lines, "juncs": open(os.path.join(d, "out.juncs")).read()})
    json.dump(cases, open(os.path.join(HERE, "producers.json"), "w"), indent=0)
    print({k: len(v) for k, v in cases.items()})


if __name__ == "__main__":
    main()
