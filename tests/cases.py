"""Shared parity cases: deterministic synthetic samples + random decompositions."""
import os
import random

from ambigram_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "tests", "data")


def fixed_cases():
    return [
        ("readme6", os.path.join(DATA, "readme6.lh"), [os.path.join(DATA, "readme6.sol")]),
        ("trx_c2", os.path.join(DATA, "trx_c2.lh"), [os.path.join(DATA, "trx_c2_chr0.sol"), os.path.join(DATA, "trx_c2_chr1.sol")]),
        # inputs reconstructed from the outputs README.md:166 / :146 hold (tests/golden/known_answers.json)
        ("readme_c2", os.path.join(DATA, "readme_c2.lh"), [os.path.join(DATA, "readme_c2_chr0.sol"), os.path.join(DATA, "readme_c2_chr1.sol")]),
        ("readme_i2", os.path.join(DATA, "readme_i2.lh"), [os.path.join(DATA, "readme_i2_chr0.sol"), os.path.join(DATA, "readme_i2_chr1.sol")]),
    ]


def synthetic_cases(workdir, small_only=True):
    out = []
    specs = [(64, 128, "chain", 9), (64, 128, "wide", 9), (64, 128, "mixed", 9), (64, 128, "wide", 13), (40, 90, "mixed", 7)]
    if not small_only:
        specs += [(256, 512, "chain", 11), (256, 512, "wide", 19), (256, 512, "mixed", 11)]
    for (n, m, tier, K) in specs:
        for seed in range(3):
            s = synth.make_sample(n, m, tier, K, seed, imperfect=seed % 3, n_del=seed % 2, n_dup=(seed + 1) % 2)
            lh, sols = s.write(workdir)
            out.append((s.name, lh, sols))
    # SV-heavy samples: deletions, duplications and short inversions that really edit the path (indelBFB)
    for seed in range(6):
        s = synth.make_sample(48, 70 + 10 * seed, ("chain", "wide", "mixed")[seed % 3], 7, 700 + seed, imperfect=seed % 2,
                              n_del=2 + seed, n_dup=1 + seed, near_inv=3 + seed, name="svheavy%d" % seed)
        lh, sols = s.write(workdir)
        out.append((s.name, lh, sols))
    # multi-chromosome with translocation (BFB-TRX, PROP C2)
    s = synth.make_sample(96, 200, "chain", 5, seed=11, n_chr=3, translocations=1, prop="PROP C2:chr1:chr2 M:chr1", name="multi3")
    lh, sols = s.write(workdir)
    out.append((s.name, lh, sols))
    # copy numbers left to the engine's reader (CN <= 0 in the file -> calculateHapDepth / calculateCopyNum, Graph.cpp:312-405):
    # every third SEG and JUNC loses its CN column value; AVG_PLOIDY is given so that the reference's ratio is initialised
    for seed in range(2):
        s = synth.make_sample(40, 80, ("chain", "wide")[seed], 7, seed=60 + seed, imperfect=seed, n_del=1, name="cnle0_%d" % seed)
        out_lines, k = [], 0
        for line in s.lh_text.splitlines():
            if line.startswith("AVG_TUMOR_PLOIDY"):
                out_lines.append("AVG_PLOIDY 2")
            if line.startswith(("SEG ", "JUNC ")):
                k += 1
                if k % 3 == 0:
                    t = line.split(" ")
                    t[4 if t[0] == "JUNC" else 3] = "-1" if k % 2 else "0"
                    line = " ".join(t)
            out_lines.append(line)
        s.lh_text = "\n".join(out_lines) + "\n"
        lh, sols = s.write(workdir)
        out.append((s.name, lh, sols))
    # BFB-TRX with insertion groups (junction pairs that leave the main chromosome and come back: LGM.cpp:4120-4190),
    # alone and mixed with a concatenation; both PROP spellings
    for seed in range(6):
        s = synth.make_sample(72 + 8 * seed, 150, ("chain", "mixed", "wide")[seed % 3], 5, seed=40 + seed, n_chr=3 + seed % 2,
                              translocations=seed % 2, trx_insertions=1 + seed % 3,
                              prop=("PROP I2:chr1:chr2 M:chr1", "PROP C2:chr1:chr3 M:chr1")[seed % 2], name="trxins%d" % seed)
        lh, sols = s.write(workdir)
        out.append((s.name, lh, sols))
    return out


def random_decomposition(workdir, seed, n=14):
    """A .lh with fold-backs everywhere + a RANDOM element set as .sol: for most element sets NO order assembles in the
    first orientation (often in neither) -- exercises the scan budget, the parallel search over whole passes and the
    orientation flip.  (All orders of a unit share their validity on every input known: tests/tools/search_mixed_validity.py.)"""
    rng = random.Random(seed)
    L = ["SAMPLE_NAME rnd%d" % seed, "AVG_CHR_SEG_DP 30", "AVG_WHOLE_HOST_DP 30", "AVG_JUNC_DP 30", "PURITY 1",
         "AVG_TUMOR_PLOIDY 2", "PLOIDY 2m1", "VIRUS_START %d" % (n + 1), "SOURCE 1", "SINK %d" % n]
    for i in range(1, n + 1):
        L.append("SEG H:%d:chr1:%d:%d 60.0 %d.0" % (i, i * 1000, i * 1000 + 999, rng.randint(2, 8)))
    for i in range(1, n):
        L.append("JUNC H:%d:+ H:%d:+ 30.0 1.0 U B" % (i, i + 1))
    for i in range(1, n + 1):
        r = rng.random()
        if r < 0.4:
            L.append("JUNC H:%d:+ H:%d:- 30.0 %d.0 U B" % (i, i, rng.randint(1, 2)))
        elif r < 0.6 and i < n:
            L.append("JUNC H:%d:+ H:%d:- 30.0 1.0 U B" % (i, i + 1))
        if rng.random() < 0.4:
            L.append("JUNC H:%d:- H:%d:+ 30.0 1.0 U B" % (i, i))
    K = rng.randint(2, 9)
    num_pat = n * (n + 1) // 2
    els = set()
    # bias towards shared endpoints so that the DAG has edges
    pts = [1, n, rng.randint(2, n - 1), rng.randint(2, n - 1)]
    while len(els) < K:
        a = rng.choice(pts + [rng.randint(1, n)])
        b = rng.choice(pts + [rng.randint(1, n)])
        if a > b:
            a, b = b, a
        els.add((rng.random() < 0.7, a, b))
    rows = []
    for (is_loop, a, b) in sorted(els):
        col = synth.rank_ab(a, b, 1, n) + (num_pat if is_loop else 0)
        rows.append((col, 1 if not is_loop else rng.randint(1, 2)))
    rows.sort()
    os.makedirs(workdir, exist_ok=True)
    lh = os.path.join(workdir, "rnd%d.lh" % seed)
    sol = os.path.join(workdir, "rnd%d.sol" % seed)
    with open(lh, "w") as f:
        f.write("\n".join(L) + "\n")
    with open(sol, "w") as f:
        f.write("Optimal - objective value 0.00000000\n")
        for col, cn in rows:
            f.write("%7d x%-7d %15d %15d\n" % (col, col, cn, 0))
    return lh, [sol]
