"""Producers of the two text inputs of `Ambigram --op bfb` (the callers on the input side of the path).

Host-side text processing, mirrored in the reference's own language (Python):

* `generate_lh`      -- seg.txt + sv.txt -> `.lh`      (script/bfb_scripts.py:500-611: findSegment, hasDuplicateSV, generate_lh)
* `barcode_to_juncs` -- seg.txt + 10x barcode BED -> `.juncs` (script/process_barcode.py:3-93)
* `om_to_juncs`      -- SegAligner optical-mapping alignment -> `.juncs` (script/bfb_scripts.py:280-298)

The functions take and return text (lists of lines / one string); the thin `write_*` helpers put the result where the
reference's command line would (`<sample>.lh`, `<sample>.juncs`).  Behaviour, including the quirks, follows the
reference and is pinned by `tests/golden/producers.json` (made by running the reference's scripts, see
`tests/golden/make_producer_golden.py`):

* segment ids are 1-based line numbers of seg.txt; a new SOURCE/SINK pair starts whenever the chromosome name changes
  from that of the CURRENT source segment (bfb_scripts.py:551-556);
* a breakpoint is mapped to the segment of the same chromosome whose left (or right) end is nearest; when no segment
  of that chromosome exists the id is `len(segs)` (bfb_scripts.py:500-512);
* adjacent `+/+` (or `-/-` in reverse) SVs are dropped, duplicates keep the larger copy number as TEXT (the comparison
  is numeric, the stored value is the string of the file) (bfb_scripts.py:563-574);
* chromosome `chr18` is written as `virus` (bfb_scripts.py:587-589); depth = copy number x 30 through Python's float
  formatting (`60.0`, `37.5`, ...) unless a depth flag says the column already is a depth (then CN = -1);
* `barcode_to_juncs` writes the FIVE heaviest links; with fewer than five links the reference dies with an
  IndexError before it opens the output file -- so does this function.
The reference's progress chatter on stdout (`print(segs)` ...) is not reproduced.
"""
import os

__all__ = ["generate_lh", "write_lh", "barcode_to_juncs", "write_barcode_juncs", "om_to_juncs", "write_om_juncs"]


# ---------------------------------------------------------------------------------------------------------------
# generate_lh
# ---------------------------------------------------------------------------------------------------------------
def _nearest_segment(segs, chrom, pos, strand, is_start):
    """bfb_scripts.py:500-512.  The right end of a segment is matched for a 5' '+' or a 3' '-' breakpoint."""
    use_right_end = (strand == "+") if is_start else (strand == "-")
    best_id, best_dist = len(segs), float("inf")
    p = int(pos)
    for sid, schrom, sstart, send, _cn in segs:
        if schrom != chrom:
            continue
        dist = abs(int(send if use_right_end else sstart) - p)
        if dist < best_dist:
            best_id, best_dist = sid, dist
    return best_id


def _find_duplicate(svs, seg1, str1, seg2, str2):
    """bfb_scripts.py:514-526: index of an SV joining the same two segment ends (either orientation of the record)."""
    for rec in svs:
        if rec[0] == seg1 and rec[2] == seg2:
            if rec[1] == str1 and rec[3] == str2:
                return svs.index(rec)
        elif rec[0] == seg2 and rec[2] == seg1:
            if str1 != str2:
                if rec[1] == str1 and rec[3] == str2:
                    return svs.index(rec)
            elif rec[1] != str1 and rec[3] != str2:
                return svs.index(rec)
    return -1


def generate_lh(seg_lines, sv_lines, coverage=30, purity=1, is_depth=False, is_seg_depth=False, is_sv_depth=False, prop=""):
    """Text of the `.lh` file.  `seg_lines`: "chr:start-end<TAB>cn" per line; `sv_lines`: header line, then
    "chr5p pos5p strand5p chr3p pos3p strand3p cn" (tab separated).  The three flags are compared with `== False`
    exactly as the reference does with its argparse strings: anything but the literal False switches the column to
    "is a depth" (bfb_scripts.py:533-537, 590-601)."""
    segs, sources, sinks = [], [1], []
    for n, line in enumerate(seg_lines, start=1):
        cols = line.strip("\n").split("\t")
        chrom, interval = cols[0].split(":")
        lo, hi = interval.split("-")[0], interval.split("-")[1]
        segs.append([n, chrom, lo, hi, cols[1]])
        if chrom != segs[sources[-1] - 1][1]:
            sinks.append(n - 1)
            sources.append(n)
    sinks.append(len(segs))

    svs = []
    for line in sv_lines[1:]:
        c = line.strip("\n").split("\t")
        a = _nearest_segment(segs, c[0], c[1], c[2], True)
        b = _nearest_segment(segs, c[3], c[4], c[5], False)
        if c[2] == c[5] and ((c[2] == "+" and int(a) + 1 == int(b)) or (c[2] == "-" and int(a) == int(b) + 1)):
            continue                                  # the reference adjacency, not an SV
        at = _find_duplicate(svs, a, c[2], b, c[5])
        if at != -1:
            if float(c[6]) > float(svs[at][-1]):
                svs[at][-1] = c[6]
        else:
            svs.append([a, c[2], b, c[5], c[6]])

    out = ["SAMPLE group1\n",
           "AVG_CHR_SEG_DP {}\n".format(coverage), "AVG_WHOLE_HOST_DP {}\n".format(coverage), "AVG_JUNC_DP {}\n".format(coverage),
           "PURITY {}\n".format(purity), "AVG_TUMOR_PLOIDY 2\n", "PLOIDY 2m1\n", "VIRUS_START 7\n",
           "SOURCE {}\n".format(",".join(str(s) for s in sources)), "SINK {}\n".format(",".join(str(s) for s in sinks))]
    seg_cn_given = is_seg_depth == False and is_depth == False      # noqa: E712 (the reference's comparison)
    sv_cn_given = is_sv_depth == False and is_depth == False        # noqa: E712
    for sid, chrom, lo, hi, val in segs:
        name = "virus" if chrom == "chr18" else chrom
        depth, cn = (float(val) * 30, val) if seg_cn_given else (val, -1)
        out.append("SEG H:{}:{}:{}:{} {} {}\n".format(sid, name, lo, hi, depth, cn))
    for a, s1, b, s2, val in svs:
        depth, cn = (float(val) * 30, val) if sv_cn_given else (val, -1)
        out.append("JUNC H:{}:{} H:{}:{} {} {} U B\n".format(a, s1, b, s2, depth, cn))
    out.append(prop)
    return "".join(out)


def write_lh(seg_path, sv_path, sample_name="test", **kw):
    """`preBFB generate_lh -sv SV -seg SEG -s NAME`: writes ./NAME.lh, returns its path."""
    text = generate_lh(open(seg_path).readlines(), open(sv_path).readlines(), **kw)
    path = "{}.lh".format(sample_name)
    with open(path, "w") as f:
        f.write(text)
    return os.path.abspath(path)


# ---------------------------------------------------------------------------------------------------------------
# barcode -> .juncs
# ---------------------------------------------------------------------------------------------------------------
def _read_segments(seg_lines):
    segs = []
    for line in seg_lines:                                                   # process_barcode.py:3-11
        name = line.split("\t")[0]
        chrom, span = name.split(":")[0], name.split(":")[1]
        segs.append((chrom, int(span.split("-")[0]), int(span.split("-")[1])))
    return segs


def _barcodes_per_segment(bed_lines, segs):
    """process_barcode.py:13-49: every molecule (chr, from, to, barcode) is assigned to the run of segments between the
    segment whose start is nearest to `from` and the one whose end is nearest to `to` (the first / last segment of the
    file catch everything before / behind them)."""
    n = len(segs)
    group = [[] for _ in range(n)]
    for line in bed_lines:
        c = line.strip("\n").split("\t")
        chrom = c[0] if c[0][0] == "c" else "chr" + c[0]
        p1, p2, code = int(c[1]), int(c[2]), c[3]
        first = last = -1
        d1 = d2 = float("inf")
        for i, (schrom, sstart, send) in enumerate(segs):
            if schrom != chrom:
                continue
            if i == 0 and p1 <= sstart:
                first = i
            elif i == n - 1 and p2 >= send:
                last = i
            else:
                if abs(sstart - p1) < d1:
                    first, d1 = i, abs(sstart - p1)
                if abs(send - p2) < d2:
                    last, d2 = i, abs(send - p2)
        if first > last or not (0 <= first < n) or not (0 <= last < n):
            continue
        for i in range(first, last + 1):
            group[i].append(code)
    return group


def _shared_barcodes(group, i, j):
    if i >= j:                                                               # process_barcode.py:51-57
        return 0
    common = set(group[i])
    for k in range(i + 1, j + 1):
        common &= set(group[k])
    return len(common)


def barcode_to_juncs(seg_lines, bed_lines):
    """Text of the `.juncs` file: the five heaviest links (shared barcodes x span), each as the run `i+ .. j+`."""
    segs = _read_segments(seg_lines)
    group = _barcodes_per_segment(bed_lines, segs)
    runs, start = [], 0                                                      # process_barcode.py:66-73: runs of one chromosome
    for i in range(1, len(segs)):
        if segs[i][0] != segs[start][0]:
            runs.append((start, i - 1))
            start = i
    if start < len(segs):
        runs.append((start, len(segs) - 1))
    links = []
    for lo, hi in runs:
        for i in range(lo, hi):
            for j in range(i + 1, hi + 1):
                links.append([i + 1, j + 1, _shared_barcodes(group, i, j) * (j - i)])
    links.sort(key=lambda rec: rec[2], reverse=True)                         # stable: ties keep generation order
    text = ""
    for k in range(5):                                                       # IndexError with fewer than five links, as the reference
        a, b, _w = links[k]
        text += "".join("{}+ ".format(s) for s in range(a, b)) + "{}+\n".format(b)
    return text


def write_barcode_juncs(seg_path, bed_path, sample_name="sample"):
    """`process_barcode.py -bed BED -seg SEG -s NAME`: writes ./NAME.juncs, returns its path."""
    text = barcode_to_juncs(open(seg_path).readlines(), open(bed_path).readlines())
    path = "{}.juncs".format(sample_name)
    with open(path, "w") as f:
        f.write(text)
    return os.path.abspath(path)


# ---------------------------------------------------------------------------------------------------------------
# optical mapping -> .juncs
# ---------------------------------------------------------------------------------------------------------------
def om_to_juncs(lines):
    """bfb_scripts.py:280-298: first column of every non-comment line is a (possibly '-'-prefixed) segment id; one line
    of `id+` / `id-` tokens, without the trailing blank."""
    text = ""
    for line in lines:
        if line.startswith("#"):
            continue
        seg = line.split("\t")[0]
        text += (seg[1:] + seg[0] + " ") if seg.startswith("-") else (seg + "+ ")
    return text[:-1]


def write_om_juncs(om_path, prefix="test"):
    text = om_to_juncs(open(om_path).readlines())
    path = "{}.juncs".format(prefix)
    with open(path, "w") as f:
        f.write(text)
    return os.path.abspath(path)
