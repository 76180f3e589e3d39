"""The drop-in CLI (ambigram_amd/csrc/ambigram_cli.cpp) keeps the reference's contract: flags, stdout
lines, side files, exit codes.  CPU run: the CLI source linked against the host simulation; the external `cbc`
(localhap.cpp:179-181) is a script on PATH that hands back the planted solution, the way a real solver would write
`<prefix>.sol`.  The GPU variant of this test lives in test_gpu_parity.py::test_cli_on_gpu."""
import os
import stat
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "tests", "data")


def fake_cbc(bindir, sols):
    """`cbc <p>.lp solve solu <p>.sol`: copies the next planted solution to $4 (one per call)."""
    os.makedirs(bindir, exist_ok=True)
    counter = os.path.join(bindir, "calls")
    with open(counter, "w") as f:
        f.write("0")
    exe = os.path.join(bindir, "cbc")
    lines = ["#!/bin/sh", "n=$(cat %s)" % counter, "echo $((n+1)) > %s" % counter, "test -s \"$1\" || exit 3", "case $n in"]
    for i, s in enumerate(sols):
        lines.append("  %d) cp %s \"$4\" ;;" % (i, s))
    lines += ["esac", "echo \"fake cbc: call $n\"", ""]
    with open(exe, "w") as f:
        f.write("\n".join(lines))
    os.chmod(exe, os.stat(exe).st_mode | stat.S_IEXEC)
    return exe


def run_cli(exe, cwd, bindir, *args):
    env = dict(os.environ, PATH=bindir + os.pathsep + os.environ.get("PATH", ""))
    return subprocess.run([exe] + list(args), cwd=cwd, env=env, capture_output=True, text=True, timeout=300)


@pytest.fixture(scope="module")
def cli(hostsim_lib):
    exe = os.path.join(ROOT, "tests", "hostsim", "Ambigram_hostsim")
    assert os.path.exists(exe)
    return exe


def check_readme(exe, cwd, oracle):
    bindir = os.path.join(cwd, "bin")
    fake_cbc(bindir, [os.path.join(DATA, "readme6.sol")])
    lh = os.path.join(DATA, "readme6.lh")
    r = run_cli(exe, cwd, bindir, "--op", "bfb", "--in_lh", lh, "--lp_prefix", "readme")
    assert r.returncode == 0, r.stderr
    want = oracle.run_bfb(lh, [os.path.join(DATA, "readme6.sol")])["log"]
    got = [l for l in r.stdout.splitlines() if not l.startswith("fake cbc")]
    assert got == want
    assert got[-1] == "1+2+3+4+5+6+|6-5-4-3-2-|2+3+4+|4-3-|3+4+|4-3-2-|2+3+4+5+6+|6-5-4-3-2-1-"   # README.md:122
    assert all(os.path.exists(os.path.join(cwd, "readme." + ext)) for ext in ("lp", "mps", "sol"))      # the three side files (LGM.cpp:4749-4750, localhap.cpp:179)
    # time.csv: name,nSeg,nInv,nOtherJunc,cnSum,pathLen,maxCN,seconds  (readme6,6,4,0,32,32,8,<sec>)
    row = open(os.path.join(cwd, "time.csv")).read().strip().split(",")
    assert row[1:7] == ["6", "4", "0", "32", "32", "8"]
    sv = open(os.path.join(cwd, "simulation_sv.txt")).read().strip().splitlines()
    assert sum(l.endswith("input") for l in sv) == 4 and sum(l.endswith("output") for l in sv) >= 1
    # simulation_sv.txt, derived by hand from localhap.cpp:326-337 + :267-289 + Vertex.cpp:27-29: every row is
    # lh \t juncs \t chrom(u) \t u.getEnd() \t dir(u) \t chrom(v) \t v.getStart() \t dir(v) \t CN \t input|output with u -> v edge A of
    # the junction; getEnd() of a '-' vertex is the segment's START, getStart() of a '-' vertex its END.  Input rows = the four
    # JUNC lines; output rows = the non-adjacent steps of README.md:122 in order of first appearance, counted: 6+|6- twice,
    # 2-|2+ twice, 4+|4- twice, 3-|3+ once.
    row = lambda cu, pu, du, cv, pv, dv, cn, kind: "\t".join([lh, "", cu, str(pu), du, cv, str(pv), dv, cn, kind])
    assert sv[:8] == [
        row("chr7", 55282001, "-", "chr7", 55282001, "+", "2", "input"),
        row("chr7", 55283001, "-", "chr7", 55283001, "+", "1", "input"),
        row("chr7", 55285000, "+", "chr7", 55285000, "-", "2", "input"),
        row("chr7", 55286001, "-", "chr7", 55286001, "+", "2", "input"),
        row("chr7", 55287000, "+", "chr7", 55287000, "-", "2", "output"),
        row("chr7", 55282001, "-", "chr7", 55282001, "+", "2", "output"),
        row("chr7", 55285000, "+", "chr7", 55285000, "-", "2", "output"),
        row("chr7", 55283001, "-", "chr7", 55283001, "+", "1", "output"),
    ]
    # reversed
    fake_cbc(bindir, [os.path.join(DATA, "readme6.sol")])
    r = run_cli(exe, cwd, bindir, "--op", "bfb", "--in_lh", lh, "--lp_prefix", "readme", "--reversed", "true")
    assert r.stdout.splitlines()[-1] == "6-5-4-3-2-1-|1+2+3+4+5+6+|6-5-4-3-2-|2+3+4+|4-3-|3+4+|4-3-2-|2+3+4+5+6+"
    # --all: one line per valid order, the oracle's stdout line for line
    for extra in ([], ["--reversed", "true"]):
        fake_cbc(bindir, [os.path.join(DATA, "readme6.sol")])
        r = run_cli(exe, cwd, bindir, "--op", "bfb", "--in_lh", lh, "--lp_prefix", "readme", "--all", "true", *extra)
        assert r.returncode == 0, r.stderr
        want = oracle.run_bfb(lh, [os.path.join(DATA, "readme6.sol")], all_=True, reversed_=bool(extra))["log"]
        got = [l for l in r.stdout.splitlines() if not l.startswith("fake cbc")]
        assert got == want and len(got) > 5


def check_trx(exe, cwd, oracle):
    bindir = os.path.join(cwd, "bin")
    sols = [os.path.join(DATA, "trx_c2_chr0.sol"), os.path.join(DATA, "trx_c2_chr1.sol")]
    fake_cbc(bindir, sols)
    lh = os.path.join(DATA, "trx_c2.lh")
    r = run_cli(exe, cwd, bindir, "--op", "bfb", "--in_lh", lh, "--lp_prefix", "trx")
    assert r.returncode == 0, r.stderr
    got = [l for l in r.stdout.splitlines() if not l.startswith("fake cbc")]
    assert got == oracle.run_bfb(lh, sols)["log"]
    assert got[-2:] == ["BFB with translocation:", "1+2+3+4+|4-3-2-|2+3+||6+7+|7-6-|6+7+|7-6-"]


def check_trx_before(exe, cwd, oracle):
    """PROP I1 / C1 (TRX-BFB): stdout line for line as the oracle's -- incl. the lines of the rebuild -- the reference-held README lines, ./new.lh
    (LGM.cpp:4294 / :4394) and the side-file rows: `input` rows are the REBUILT graph's junctions, `output` rows name segments of the FILE."""
    import json
    known = json.load(open(os.path.join(ROOT, "tests", "golden", "known_answers.json")))
    for key in ("readme_i1", "readme_c1"):
        d = known[key]
        lh, sols = os.path.join(ROOT, d["lh"]), [os.path.join(ROOT, x) for x in d["sols"]]
        sub = os.path.join(cwd, key)
        os.makedirs(sub, exist_ok=True)
        bindir = os.path.join(sub, "bin")
        fake_cbc(bindir, sols)
        r = run_cli(exe, sub, bindir, "--op", "bfb", "--in_lh", lh, "--lp_prefix", key)
        assert r.returncode == 0, r.stderr
        got = [l for l in r.stdout.splitlines() if not l.startswith("fake cbc")]
        assert got == oracle.run_bfb(lh, sols)["log"]
        for line in d["reference_held_lines"]:
            assert line in got
        new_lh = open(os.path.join(sub, "new.lh")).read().splitlines()
        n_seg = sum(l.startswith("SEG ") for l in new_lh)
        assert n_seg == {"readme_i1": 6, "readme_c1": 7}[key] and new_lh[0] == "SAMPLE_NAME TEST"
        sv = open(os.path.join(sub, "simulation_sv.txt")).read().strip().splitlines()
        ins = [l.split("\t") for l in sv if l.endswith("input")]
        outs = [l.split("\t") for l in sv if l.endswith("output")]
        assert len(ins) == sum(l.startswith("JUNC ") for l in new_lh)
        if key == "readme_i1":
            # the restored path 1+2+3+||6+||4+|4-||6-||3-2-|2+3+||6+... : its first junction step is 3+ -> 6+ (chr8 -> virus), counted with
            # its complement 6- -> 3-: four times in all
            assert outs[0][2:9] == ["chr8", "4000", "+", "virus", "1", "+", "4"]
        row = open(os.path.join(sub, "time.csv")).read().strip().split(",")
        assert row[1] == str(n_seg)


def check_errors(exe, cwd):
    bindir = os.path.join(cwd, "nobin")
    os.makedirs(bindir, exist_ok=True)
    r = run_cli(exe, cwd, bindir, "--op", "bfb", "--in_lh", os.path.join(cwd, "missing.lh"), "--lp_prefix", "x")
    assert r.returncode == 1 and "Cannot open file" in r.stderr          # Graph.cpp:111-114
    env_path = os.environ.get("PATH", "")
    if not any(os.path.exists(os.path.join(p, "cbc")) for p in env_path.split(os.pathsep) if p):
        r = run_cli(exe, cwd, bindir, "--op", "bfb", "--in_lh", os.path.join(DATA, "readme6.lh"), "--lp_prefix", "nosol")
        assert r.returncode == 1 and "ILP error: cannot open file" in r.stderr   # localhap.cpp:187-190


def test_cli_readme(cli, oracle, tmp_path):
    check_readme(cli, str(tmp_path), oracle)


def test_cli_two_chromosome_trx(cli, oracle, tmp_path):
    check_trx(cli, str(tmp_path), oracle)


def test_cli_trx_before(cli, oracle, tmp_path):
    check_trx_before(cli, str(tmp_path), oracle)


def test_cli_errors(cli, tmp_path):
    check_errors(cli, str(tmp_path))


def test_cli_with_a_solver_that_reads_the_lp(cli, tmp_path):
    """End to end with a REAL solve: the .lp this CLI writes is read back by an independent LP-format parser and solved
    as a MILP (HiGHS via scipy; Cbc is not in the image), the .sol goes through the reference's token scan, and the
    README example comes out as README.md:122 prints it."""
    cwd = str(tmp_path)
    bindir = os.path.join(cwd, "bin")
    os.makedirs(bindir)
    exe = os.path.join(bindir, "cbc")
    with open(exe, "w") as f:
        f.write("#!/bin/sh\npython3 %s \"$@\" | sed 's/^/fake cbc: /'\n" % os.path.join(ROOT, "tests", "tools", "cbc_highs.py"))
    os.chmod(exe, os.stat(exe).st_mode | stat.S_IEXEC)
    r = run_cli(cli, cwd, bindir, "--op", "bfb", "--in_lh", os.path.join(DATA, "readme6.lh"), "--lp_prefix", "solved")
    assert r.returncode == 0, r.stderr
    assert "optimal" in r.stdout
    assert r.stdout.splitlines()[-1] == "1+2+3+4+5+6+|6-5-4-3-2-|2+3+4+|4-3-|3+4+|4-3-2-|2+3+4+5+6+|6-5-4-3-2-1-"   # README.md:122
    sol = open(os.path.join(cwd, "solved.sol")).read()
    assert sol.startswith("Optimal - objective value")
    picked = {int(l.split()[1][1:]): float(l.split()[2]) for l in sol.splitlines()[1:] if int(l.split()[1][1:]) < 42}
    assert picked == {26: 1, 29: 1, 31: 1, 33: 1}       # l(1,6), l(2,4), l(2,6), l(3,4): SURVEY.md appendix B.4


def test_cli_solver_failures(cli, tmp_path):
    """solver plumbing around localhap.cpp:179-190: a solver that exits non-zero without a .sol, one that never
    returns (--solver_timeout, an extension flag), and a stale .sol left by an earlier run must not be taken."""
    cwd = str(tmp_path)
    bindir = os.path.join(cwd, "bin")
    os.makedirs(bindir)
    exe = os.path.join(bindir, "cbc")
    lh = os.path.join(DATA, "readme6.lh")

    def script(body):
        with open(exe, "w") as f:
            f.write("#!/bin/sh\n" + body + "\n")
        os.chmod(exe, os.stat(exe).st_mode | stat.S_IEXEC)

    with open(os.path.join(cwd, "f.sol"), "w") as f:       # stale solution from "an earlier run"
        f.write(open(os.path.join(DATA, "readme6.sol")).read())
    script("exit 7")
    r = run_cli(cli, cwd, bindir, "--op", "bfb", "--in_lh", lh, "--lp_prefix", "f")
    assert r.returncode == 1 and "status 7" in r.stderr and "ILP error: cannot open file ./f.sol" in r.stderr
    script("sleep 30")
    r = run_cli(cli, cwd, bindir, "--op", "bfb", "--in_lh", lh, "--lp_prefix", "f", "--solver_timeout", "1")
    assert r.returncode == 1 and "did not finish within 1 s" in r.stderr and "cannot open file ./f.sol" in r.stderr


def test_cli_many_chromosomes_one_batch(cli, oracle, tmp_path):
    """BASELINE config 4 shape through the CLI: 8 chromosomes, a concatenation and insertion groups (BFB-TRX) -- one probe
    batch, one reconstruct batch; stdout equals the oracle's line for line."""
    from ambigram_amd import synth
    cwd = str(tmp_path)
    s = synth.make_sample(256, 520, "chain", 5, seed=4004, n_chr=8, translocations=1, trx_insertions=3, prop="PROP C2:chr1:chr2 M:chr1", name="c4small")
    lh, sols = s.write(cwd)
    bindir = os.path.join(cwd, "bin")
    fake_cbc(bindir, sols)
    r = run_cli(cli, cwd, bindir, "--op", "bfb", "--in_lh", lh, "--lp_prefix", "c4")
    assert r.returncode == 0, r.stderr
    got = [l for l in r.stdout.splitlines() if not l.startswith("fake cbc")]
    want = oracle.run_bfb(lh, sols)
    assert got == want["log"]
    assert "insert" in want["trx_trace"]


def check_sample_list_over_devices(exe, cwd, oracle, devices):
    """Extension: --in_lh lists several samples, their chromosomes are ONE reconstruct batch dealt over --devices; every sample's
    lines and side-file rows are those of a run on that sample alone (here: the oracle's), in sample order."""
    from ambigram_amd import synth
    bindir = os.path.join(cwd, "bin")
    samples = [(os.path.join(DATA, "readme6.lh"), [os.path.join(DATA, "readme6.sol")]),
               (os.path.join(DATA, "readme_i2.lh"), [os.path.join(DATA, "readme_i2_chr0.sol"), os.path.join(DATA, "readme_i2_chr1.sol")])]
    for i in range(3):
        s = synth.make_sample(64, 128, ("wide", "chain", "mixed")[i], 9, seed=7100 + i, n_del=i, n_dup=1, name="cl%d" % i)
        samples.append(s.write(cwd))
    fake_cbc(bindir, [p for _, sols in samples for p in sols])
    args = ["--op", "bfb", "--in_lh", ",".join(lh for lh, _ in samples), "--lp_prefix", "many"]
    if devices:
        args += ["--devices", devices]
    r = run_cli(exe, cwd, bindir, *args)
    assert r.returncode == 0, r.stderr
    got = [l for l in r.stdout.splitlines() if not l.startswith("fake cbc")]
    logs = [oracle.run_bfb(lh, sols)["log"] for lh, sols in samples]
    # per sample: the loading lines first (printed while the sample goes through its ILP stage), the path lines after the batch
    want_front, want_back = ["bfb"], []
    for lh, sols in samples:
        log = oracle.run_bfb(lh, sols)["log"][1:]
        k = next(i for i, l in enumerate(log) if l == "Declare done" or (l and l[0].isdigit() and ("+" in l or "-" in l) and " " not in l))
        want_front += log[:k]; want_back += log[k:]
    assert got == want_front + want_back
    assert "1+2+3+||5+6+7+|7-6-||8+9+||4-3-2-|2+3+4+|4-3-" in got          # README.md:146 among them
    rows = open(os.path.join(cwd, "time.csv")).read().strip().splitlines()
    assert len(rows) == len(samples) and rows[0].split(",")[1:7] == ["6", "4", "0", "32", "32", "8"]
    assert len(logs) == len(samples)


def test_cli_sample_list_sharded(cli, oracle, tmp_path):
    check_sample_list_over_devices(cli, str(tmp_path), oracle, "0,0,0")


def test_cli_sample_list_one_device(cli, oracle, tmp_path):
    check_sample_list_over_devices(cli, str(tmp_path), oracle, "")
