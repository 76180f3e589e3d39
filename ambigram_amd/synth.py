"""Deterministic synthetic `.lh` + planted `.sol` generator.

A *planted decomposition* (BFB patterns p(a,b) / loops l(a,b,cn) on the segments of each chromosome) is drawn
first and the `.lh` is derived from it: segment CN = sum(p) + 2*sum(l*cn) over the covering elements
(localhap.cpp:222-232), a fold-back junction at every loop end, the n-1 normal `+/+` adjacencies, and padding
junctions (distant inversions / deletions / duplications / translocations) up to the requested junction count.
The matching `.sol` lists the planted elements as CBC would (`<col> x<col> <value> <reduced cost>`; column =
numPat*[is loop] + rank(a,b), localhap.cpp:122-133), so the external solver is taken out of the loop exactly as
SURVEY.md 8c/8d prescribes.

DAG-shape tiers (K = number of planted elements, R = number of topological orders of the BFB DAG):
  chain : nested loops, each sharing an end with its parent            -> R = 1
  wide  : root l(s,e) + {l(s+i,e)} i=1..k + {l(s,e-j)} j=1..k           -> K = 2k+1, R = C(2k,k)
          (K=19 -> 48 620 orders, K=25 -> 2 704 156 orders)
  skew  : root + {l(s+i,e)} i=1..K-4 + {l(s,e-j)} j=1..3                -> R = C(K-1,3)  (wide rows, moderate R)
  mixed : chain of loops + one or two patterns hanging off the telomere
Seeds follow SURVEY.md 8d: seed = 1000*config + sample_index.
"""
from dataclasses import dataclass, field
from typing import List, Tuple
import os
import random


@dataclass
class Element:
    kind: str   # 'p' or 'l'
    a: int
    b: int
    cn: int


@dataclass
class SynthSample:
    name: str
    lh_text: str
    sol_texts: List[str]              # one per chromosome (every chromosome reaches the ILP)
    elements: List[List[Element]]     # planted elements per chromosome
    chr_ranges: List[Tuple[int, int]]
    n_seg: int
    n_junc: int
    meta: dict = field(default_factory=dict)

    def write(self, directory, stem=None):
        stem = stem or self.name
        os.makedirs(directory, exist_ok=True)
        lh = os.path.join(directory, stem + ".lh")
        with open(lh, "w") as f:
            f.write(self.lh_text)
        sols = []
        for c, t in enumerate(self.sol_texts):
            p = os.path.join(directory, "%s.chr%d.sol" % (stem, c))
            with open(p, "w") as f:
                f.write(t)
            sols.append(p)
        return lh, sols


def rank_ab(a, b, s, n):
    """rank of (a,b) in the lexicographic enumeration of all s<=a<=b<=s+n-1 (LGM.cpp:3254-3264)."""
    da = a - s
    return da * n - da * (da - 1) // 2 + (b - a)


def sol_text(elements, s, e):
    n = e - s + 1
    num_pat = n * (n + 1) // 2
    rows = []
    for el in elements:
        col = rank_ab(el.a, el.b, s, n) + (num_pat if el.kind == 'l' else 0)
        rows.append((col, el.cn))
    rows.sort()
    out = ["Optimal - objective value 0.00000000"]
    for col, cn in rows:
        out.append("%7d x%-7d %15d %15d" % (col, col, cn, 0))
    return "\n".join(out) + "\n"


def planted_chain(rng, s, e, K, cn_choices=(1, 2)):
    """K nested loops; each shares one end with its parent (alternating which end moves)."""
    els = [Element('l', s, e, rng.choice(cn_choices))]
    a, b = s, e
    move_start = True
    while len(els) < K and b - a >= 1:
        span = b - a
        step = rng.randint(1, max(1, span // max(2, (K - len(els)))))
        if move_start:
            a = min(a + step, b)
        else:
            b = max(b - step, a)
        els.append(Element('l', a, b, rng.choice(cn_choices)))
        move_start = not move_start
    return els


def planted_wide(rng, s, e, k, cn_choices=(1, 2)):
    """root + two incomparable families -> R = C(2k,k) topological orders."""
    assert e - s >= 2 * k + 1, "chromosome too short for wide tier"
    els = [Element('l', s, e, rng.choice(cn_choices))]
    for i in range(1, k + 1):
        els.append(Element('l', s + i, e, rng.choice(cn_choices)))
    for j in range(1, k + 1):
        els.append(Element('l', s, e - j, rng.choice(cn_choices)))
    return els


def planted_skew(rng, s, e, K, cn_choices=(1, 2), k2=3):
    """root + a long family of K-1-k2 loops and a short one of k2 -> R = C(K-1, k2): many nodes, moderate R."""
    k1 = K - 1 - k2
    assert k1 >= 1 and e - s >= k1 + k2 + 1, "chromosome too short for skew tier"
    els = [Element('l', s, e, rng.choice(cn_choices))]
    for i in range(1, k1 + 1):
        els.append(Element('l', s + i, e, rng.choice(cn_choices)))
    for j in range(1, k2 + 1):
        els.append(Element('l', s, e - j, rng.choice(cn_choices)))
    return els


def planted_mixed(rng, s, e, K, cn_choices=(1, 2)):
    """chain of loops followed by a short pattern chain hanging off the telomere (adds p-nodes to the DAG)."""
    kl = max(1, K - 2)
    els = planted_chain(rng, s, e, kl, cn_choices)
    x = rng.randint(s + 1, e) if e > s + 1 else e
    els.append(Element('p', s, x, 1))
    if len(els) < K and x - s >= 2:
        y = rng.randint(s + 1, x - 1)
        els.append(Element('p', y, x, 1))
    return els


def make_sample(n_seg=64, n_junc=128, tier="chain", K=9, seed=0, n_chr=1, imperfect=0, n_del=0, n_dup=0,
                translocations=0, name=None, prop=None, cn_choices=(1, 2), near_inv=0, trx_insertions=0):
    """Build one synthetic sample. `tier` in {chain, wide, skew, mixed}. For wide, K must be odd (K = 2k+1).
    `translocations`: single inter-chromosome junctions (concatenation groups of translocationBFB, LGM.cpp:4099-4119);
    `trx_insertions`: PAIRS of junctions that leave the main (first) chromosome and come back to it (insertion groups,
    LGM.cpp:4120-4190), in random orientation and written from either side."""
    rng = random.Random(seed)
    name = name or "syn_n%d_m%d_%s_K%d_s%d" % (n_seg, n_junc, tier, K, seed)
    # chromosome ranges
    base = n_seg // n_chr
    ranges = []
    lo = 1
    for c in range(n_chr):
        hi = lo + base - 1 if c < n_chr - 1 else n_seg
        ranges.append((lo, hi))
        lo = hi + 1
    seg_cn = [0] * (n_seg + 1)
    all_elements = []
    juncs = []          # (src, sdir, tgt, tdir, cn)
    seen = set()

    def key(sv):
        src, sd, tgt, td = sv
        u = src if sd == '+' else -src
        v = tgt if td == '+' else -tgt
        return (u, v), (-v, -u)

    def add(src, sd, tgt, td, cn):
        k1, k2 = key((src, sd, tgt, td))
        if k1 in seen or k2 in seen:
            return False
        seen.add(k1)
        juncs.append((src, sd, tgt, td, cn))
        return True

    sols = []
    for (s, e) in ranges:
        if tier == "chain":
            els = planted_chain(rng, s, e, K, cn_choices)
        elif tier == "wide":
            assert K % 2 == 1
            els = planted_wide(rng, s, e, (K - 1) // 2, cn_choices)
        elif tier == "mixed":
            els = planted_mixed(rng, s, e, K, cn_choices)
        elif tier.startswith("skew"):   # "skew" or "skew<k2>", e.g. skew5 -> R = C(K-1, 5)
            els = planted_skew(rng, s, e, K, cn_choices, k2=int(tier[4:] or 3))
        else:
            raise ValueError(tier)
        all_elements.append(els)
        for el in els:
            for i in range(el.a, el.b + 1):
                seg_cn[i] += el.cn if el.kind == 'p' else 2 * el.cn
        sols.append(sol_text(els, s, e))
    # normal adjacencies
    for (s, e) in ranges:
        for i in range(s, e):
            add(i, '+', i + 1, '+', max(1, min(seg_cn[i], seg_cn[i + 1])))
    # fold-back junctions at loop ends
    imp_left = imperfect
    for els, (s, e) in zip(all_elements, ranges):
        ends, starts = {}, {}
        for el in els:
            if el.kind == 'l':
                ends[el.b] = ends.get(el.b, 0) + el.cn
                starts[el.a] = starts.get(el.a, 0) + el.cn
        for b, cn in sorted(ends.items()):
            if imp_left > 0 and b + 1 <= e and (b + 1) not in ends:
                add(b, '+', b + 1, '-', cn)      # imperfect fold-back: joins two different segments
                imp_left -= 1
            else:
                add(b, '+', b, '-', cn)
        for a, cn in sorted(starts.items()):
            add(a, '-', a, '+', cn)
    # deletions / duplications (same-strand, non-adjacent; LGM.cpp:3699-3744, 3746-3837)
    for (s, e) in ranges[:1]:
        span = e - s
        for _ in range(n_del):
            if span < 6:
                break
            i = rng.randint(s, e - 3)
            add(i, '+', i + 2, '+', 1)
        for _ in range(n_dup):
            if span < 6:
                break
            i = rng.randint(s + 2, e - 1)
            add(i, '+', i - 1, '+', 1)
    # short inversions (strand switch 3..5 segments apart): the indelBFB "inversion" branch erases across them
    for (s, e) in ranges[:1]:
        for _ in range(near_inv):
            if e - s < 8:
                break
            i = rng.randint(s, e - 5)
            j = i + rng.randint(3, 5)
            if rng.random() < 0.5:
                add(i, '+', j, '-', 1)
            else:
                add(j, '-', i, '+', 1)
    # translocations between consecutive chromosomes
    for t in range(translocations):
        if n_chr < 2:
            break
        c0 = t % (n_chr - 1)
        (s0, e0), (s1, e1) = ranges[c0], ranges[c0 + 1]
        add(rng.randint(s0, e0), '+', rng.randint(s1, e1), '+', 1)
    # insertion groups: main chromosome i -> other chromosome [x..y] -> main chromosome i+1
    for t in range(trx_insertions):
        if n_chr < 2:
            break
        (s0, e0) = ranges[0]
        c = 1 + rng.randrange(n_chr - 1)
        (s1, e1) = ranges[c]
        i = rng.randint(s0, e0 - 1)
        x = rng.randint(s1, e1)
        y = rng.randint(x, e1)
        if rng.random() < 0.5:
            out, back = (i, '+', x, '+'), (y, '+', i + 1, '+')     # the stretch x..y inserted forward
        else:
            out, back = (i, '+', y, '-'), (x, '-', i + 1, '+')     # inserted reverse-complemented
        for (a, ad, b, bd) in (out, back):
            if rng.random() < 0.5:                                  # the same junction written from the other side
                flip = {'+': '-', '-': '+'}
                a, ad, b, bd = b, flip[bd], a, flip[ad]
            add(a, ad, b, bd, 1)
    # padding: distant head-to-head inversions  H:i:+ H:j:-  (|i-j| >= 6) inside a chromosome.  The reference ignores
    # them in getJuncCN/getIndelBias, but indelBFB collects, groups and looks every one of them up in the path
    # (LGM.cpp:3750-3818).  Head-to-head junctions cannot chain with one another in its deque grouping (an edge
    # target is always a '-' vertex, a group front always '+'), so they stay 2-vertex "inversion" groups whose ends
    # are more than 5 path cells apart: unexplained SVs that are examined but do not rewrite the BFB path.
    guard = 0
    while len(juncs) < n_junc and guard < 100 * n_junc:
        guard += 1
        (s, e) = ranges[rng.randrange(n_chr)]
        if e - s < 8:
            continue
        i = rng.randint(s, e - 7)
        j = rng.randint(i + 6, e)
        add(i, '+', j, '-', 1)
    # text
    L = []
    L.append("SAMPLE_NAME %s" % name)
    L.append("AVG_CHR_SEG_DP 30")
    L.append("AVG_WHOLE_HOST_DP 30")
    L.append("AVG_JUNC_DP 30")
    L.append("PURITY 1")
    L.append("AVG_TUMOR_PLOIDY 2")
    L.append("PLOIDY 2m1")
    L.append("VIRUS_START %d" % (n_seg + 1))
    L.append("SOURCE " + ",".join(str(s) for s, _ in ranges))
    L.append("SINK " + ",".join(str(e) for _, e in ranges))
    for c, (s, e) in enumerate(ranges):
        for i in range(s, e + 1):
            st = (i - s) * 1000 + 1
            L.append("SEG H:%d:chr%d:%d:%d %.1f %.1f" % (i, c + 1, st, st + 999, 15.0 * seg_cn[i], float(seg_cn[i])))
    for (src, sd, tgt, td, cn) in juncs:
        L.append("JUNC H:%d:%s H:%d:%s %.1f %.1f U B" % (src, sd, tgt, td, 15.0 * cn, float(cn)))
    if prop:
        L.append(prop)
    return SynthSample(name=name, lh_text="\n".join(L) + "\n", sol_texts=sols, elements=all_elements,
                       chr_ranges=ranges, n_seg=n_seg, n_junc=len(juncs),
                       meta={"tier": tier, "K": K, "seed": seed, "n_chr": n_chr})


# BASELINE.json configs: C1 = 64/128, C2 = 256/512, C3 = 1024 x 64-seg batch, C4 = 1024/2048 multi-chr
def config_sample(config, index=0, tier="chain", K=9):
    seed = 1000 * config + index
    if config == 1:
        return make_sample(64, 128, tier, K, seed)
    if config == 2:
        return make_sample(256, 512, tier, K, seed)
    if config == 3:
        return make_sample(64, 128, tier, K, seed)
    if config == 4:
        return make_sample(1024, 2048, tier, K, seed, n_chr=8, translocations=1, trx_insertions=3, prop="PROP C2:chr1:chr2 M:chr1")
    raise ValueError(config)


def joint_sol_texts(samples):
    """`--op sc_bfb`: the joint .sol of several samples with the same segmentation, one text per chromosome: the columns of
    sample k are shifted into block k (x + k * numComp, numComp = n(n+1) for a chromosome of n segments; localhap.cpp:540-566)."""
    out = []
    for c, (s, e) in enumerate(samples[0].chr_ranges):
        n = e - s + 1
        num_comp = n * (n + 1)
        lines = ["Optimal - objective value 0.00000000"]
        for k, smp in enumerate(samples):
            assert smp.chr_ranges == samples[0].chr_ranges
            for line in smp.sol_texts[c].splitlines()[1:]:
                t = line.split()
                col = int(t[0]) + k * num_comp
                lines.append("%7d x%-7d %15s %15s" % (col, col, t[2], t[3]))
        out.append("\n".join(lines) + "\n")
    return out
