#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: a stand-in for the external `cbc` executable (localhap.cpp:179-181) that really solves.

    cbc <prefix>.lp solve solu <prefix>.sol

Reads the CPLEX-LP text this repo's CLI writes (own strict parser below: it accepts only what the LP format defines, so a
malformed file fails loudly), solves the MILP with scipy's HiGHS (`scipy.optimize.milp`; COIN-OR Cbc itself is not in the
image) and writes the solution in the layout Cbc's `solu` command uses -- the layout the reference's token scan expects
(localhap.cpp:196-211): a status line with the objective value, then `index name value reduced-cost` per non-zero column.
"""
import re
import sys

import numpy as np
from scipy.optimize import Bounds, LinearConstraint, milp
from scipy.sparse import csr_matrix


def parse_lp(text):
    # strip comments, split into sections
    lines = [l for l in (ln.split("\\")[0].rstrip() for ln in text.splitlines()) if l.strip()]
    sec_names = {"minimize": "obj", "minimum": "obj", "min": "obj", "subject to": "rows", "such that": "rows", "st": "rows", "s.t.": "rows",
                 "bounds": "bounds", "bound": "bounds", "integers": "int", "integer": "int", "general": "int", "generals": "int",
                 "binary": "bin", "binaries": "bin", "end": "end"}
    sections, cur = {}, None
    for l in lines:
        key = l.strip().lower()
        if key in sec_names and not l.startswith(" "):
            cur = sec_names[key]
            sections.setdefault(cur, [])
            continue
        if cur is None:
            raise ValueError("text before the first section: %r" % l)
        sections[cur].append(l)
    if "end" not in sections:
        raise ValueError("no End")
    cols = {}

    def col(name):
        if not re.fullmatch(r"[A-Za-z_][A-Za-z0-9_.]*", name):
            raise ValueError("bad column name %r" % name)
        return cols.setdefault(name, len(cols))

    def linear(expr):
        """'2 x1 - x2 + 0.5 x3' -> {col: coef}"""
        toks = re.findall(r"[+-]|[0-9.]+(?:[eE][+-]?[0-9]+)?|[A-Za-z_][A-Za-z0-9_.]*", expr)
        if "".join(toks) != re.sub(r"\s+", "", expr):
            raise ValueError("unparsed characters in %r" % expr)
        out, sign, coef = {}, 1.0, None
        for t in toks:
            if t in "+-":
                sign = 1.0 if t == "+" else -1.0
            elif re.match(r"[0-9.]", t):
                coef = float(t)
            else:
                c = col(t)
                out[c] = out.get(c, 0.0) + sign * (1.0 if coef is None else coef)
                sign, coef = 1.0, None
        if coef is not None:
            raise ValueError("dangling coefficient in %r" % expr)
        return out

    def statements(body):
        """'name: expr' statements; continuation lines start with a blank"""
        out = []
        for l in body:
            if re.match(r"^\s*[A-Za-z_][A-Za-z0-9_.]*\s*:", l):
                out.append(l.strip())
            elif out:
                out[-1] += " " + l.strip()
            else:
                out.append(l.strip())
        return out

    obj = {}
    for st in statements(sections.get("obj", [])):
        obj = linear(st.split(":", 1)[1] if ":" in st else st)
    rows = []
    for st in statements(sections.get("rows", [])):
        body = st.split(":", 1)[1] if ":" in st else st
        m = re.fullmatch(r"(.*?)(<=|>=|=<|=>|=)\s*([+-]?[0-9.]+(?:[eE][+-]?[0-9]+)?)\s*", body)
        if not m:
            raise ValueError("bad constraint %r" % st)
        lhs, op, rhs = linear(m.group(1)), m.group(2), float(m.group(3))
        rows.append((lhs, op, rhs))
    lo, up = {}, {}
    for l in sections.get("bounds", []):
        t = l.split()
        if len(t) == 5 and t[1] == "<=" and t[3] == "<=":
            lo[col(t[2])], up[col(t[2])] = float(t[0]), float(t[4])
        elif len(t) == 3 and t[1] == "=":
            lo[col(t[0])] = up[col(t[0])] = float(t[2])
        elif len(t) == 3 and t[1] == ">=":
            lo[col(t[0])] = float(t[2])
        elif len(t) == 3 and t[1] == "<=":
            up[col(t[0])] = float(t[2])
        elif len(t) == 2 and t[1].lower() == "free":
            lo[col(t[0])], up[col(t[0])] = -np.inf, np.inf
        else:
            raise ValueError("bad bound %r" % l)
    ints = set()
    for l in sections.get("int", []):
        ints.update(col(t) for t in l.split())
    for l in sections.get("bin", []):
        for t in l.split():
            ints.add(col(t)); lo[col(t)], up[col(t)] = 0.0, 1.0
    return cols, obj, rows, lo, up, ints


def main(argv):
    if len(argv) != 5 or argv[2] != "solve" or argv[3] != "solu":
        print("usage: cbc <file>.lp solve solu <file>.sol", file=sys.stderr)
        return 2
    cols, obj, rows, lo, up, ints = parse_lp(open(argv[1]).read())
    n = len(cols)
    names = [None] * n
    for k, v in cols.items():
        names[v] = k
    c = np.zeros(n)
    for k, v in obj.items():
        c[k] = v
    data, ri, ci, lb, ub = [], [], [], [], []
    for r, (lhs, op, rhs) in enumerate(rows):
        for k, v in lhs.items():
            ri.append(r); ci.append(k); data.append(v)
        lb.append(rhs if op in (">=", "=>", "=") else -np.inf)
        ub.append(rhs if op in ("<=", "=<", "=") else np.inf)
    A = csr_matrix((data, (ri, ci)), shape=(len(rows), n))
    res = milp(c, constraints=LinearConstraint(A, lb, ub), integrality=np.array([1 if j in ints else 0 for j in range(n)]),
               bounds=Bounds(np.array([lo.get(j, 0.0) for j in range(n)]), np.array([up.get(j, np.inf) for j in range(n)])))
    with open(argv[4], "w") as f:
        if not res.success:
            f.write("Infeasible - objective value 0.00000000\n")
        else:
            f.write("Optimal - objective value %.8f\n" % res.fun)
            for j in range(n):
                v = res.x[j]
                if abs(v) > 1e-9:
                    # Cbc names the columns of an LP file by their names in it; the reference needs x<index> (localhap.cpp:204-206)
                    f.write("%7d %-8s %15.8g %15.8g\n" % (j, names[j], round(v) if j in ints else v, 0.0))
    print("cbc stand-in (HiGHS): %s, objective %s" % ("optimal" if res.success else "infeasible", res.fun if res.success else "-"))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
