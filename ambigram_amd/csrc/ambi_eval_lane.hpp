// ambi_eval_lane.hpp -- the per-order evaluation of getBFB (LGM.cpp:3519-3658) + imperfectFBI (:3431-3512) as ONE THREAD
// PER ORDER.
//
// ambi_eval.hpp evaluates one order with a whole wavefront (its scans, reductions and shifts are spread over the lanes):
// the right shape when ONE order decides a unit (the scan for the first valid order, the express path).  --all evaluates
// every order of every unit -- hundreds of millions of independent evaluations -- and there the wavefront form wastes
// the machine: 64 lanes cooperate on ~100 cells.  Here every lane walks its own order through the reference's scalar
// algorithm; the breakpoint cells of the 64 orders of a wavefront are interleaved in group memory (cell i of lane l at
// [i * 64 + l]), so a step in which all lanes touch "their cell i" is one conflict-free access.
// Same verdicts as eval_order (checked order by order against it and against the oracle in the --all tests).
#pragma once
#include "ambi_eval.hpp"

namespace ambi {

// cells of one lane, `stride` apart
struct LaneCells {
    cell_t* base;
    int stride;
    AMBI_HD cell_t get(int i) const { return base[(int64_t)i * stride]; }
    AMBI_HD void set(int i, int v) const { base[(int64_t)i * stride] = (cell_t)v; }
};

// imperfectFBI (LGM.cpp:3431-3512) on one lane's cells.  false: the reference would have touched the cell behind the end.
AMBI_HD bool imperfect_fbi_lane(const LaneCells& b, int L, const InvMap& inv) {
    int pos = 0;
    while (pos < L) {
        if (pos + 1 >= L) return false;
        const int c0 = b.get(pos), c1 = b.get(pos + 1);
        int r = L;
        // first later cell holding -c0: four cells per round (independent loads: one latency, not four)
        for (int q = pos + 3; q < L && r == L; q += 4) {
            const int last = L - 1;
            const int v0 = b.get(q), v1 = b.get(q + 1 < last ? q + 1 : last), v2 = b.get(q + 2 < last ? q + 2 : last), v3 = b.get(q + 3 < last ? q + 3 : last);
            if (v0 == -c0) r = q;
            else if (q + 1 < L && v1 == -c0) r = q + 1;
            else if (q + 2 < L && v2 == -c0) r = q + 2;
            else if (q + 3 < L && v3 == -c0) r = q + 3;
        }
        const int l = r - 1;
        if (r == L || b.get(l) != -c1) {
            int n0 = c0, n1 = c1;
            {
                const int id = iabs(c1);
                const int s = inv.src[id];
                if (s != 0) {
                    const int t = inv.tgt[id];
                    n1 = c1 > 0 ? ((s < t) ? s : t) : ((s < t) ? -t : -s);
                }
            }
            if (pos > 0) {
                const int id = iabs(c0);
                const int s = inv.src[id];
                if (s != 0 && iabs(b.get(pos - 1)) == id) {
                    const int other = (s == id) ? inv.tgt[id] : s;
                    n0 = c0 > 0 ? other : -other;
                }
            }
            if (n0 > 0 && iabs(n0) > iabs(n1)) n1 = n0;
            if (n0 < 0 && iabs(n0) < iabs(n1)) n1 = n0;
            b.set(pos, n0); b.set(pos + 1, n1);
            pos += 2;
        } else {
            int p1 = pos + ((l - pos) / 2), p2 = p1 + 1;
            while (p1 >= pos - 1 && p1 > 0) {
                const int v = b.get(p1);
                const int id = iabs(v);
                const int s = inv.src[id];
                if (s != 0) {
                    const int tt = inv.tgt[id];
                    if (p1 + 1 >= L) return false;
                    int w0, w1;
                    if (v > 0) { if (s < tt) { w0 = s; w1 = -tt; } else { w0 = tt; w1 = -s; } }
                    else { if (s < tt) { w0 = -tt; w1 = s; } else { w0 = -s; w1 = tt; } }
                    b.set(p1, w0); b.set(p1 + 1, w1);
                    if (p2 != p1 + 1) {
                        if (p1 > pos - 1) { if (p2 >= L) return false; b.set(p2, -w0); }
                        b.set(p2 - 1, -w1);
                    }
                }
                p1 -= 2; p2 += 2;
            }
            pos = r + 1;
        }
    }
    return true;
}

// One order on one lane.  ord[d * ord_stride] = node at position d.  Returns 1 valid / 0 invalid / negative Status, as
// eval_order does; *L_out = length of the breakpoint path.
AMBI_HD int eval_order_lane(const Dag& D, const uint8_t* ord, int ord_stride, bool forward, const InvMap& inv, const LaneCells& b, int cap, int* L_out) {
    const int K = D.K;
    int L = 0;
    *L_out = 0;
    int x = ord[0];
    const bool isPat = D.pat[x][0] != 0, isLoop = D.loop[x][0] != 0;
    if (!isPat && !isLoop) return ST_ERR_REF_UB;   // the reference indexes an empty vector
    {
        const int s = isPat ? D.pat[x][0] : D.loop[x][0], e = isPat ? D.pat[x][1] : D.loop[x][1];
        int q[4];
        if (forward) { q[0] = s; q[1] = e; q[2] = -e; q[3] = -s; } else { q[0] = -e; q[1] = -s; q[2] = s; q[3] = e; }
        const int len = isPat ? 2 : 4 * D.loop[x][2];
        if (len > cap) return ST_ERR_BKP_CAPACITY;
        for (int i = 0; i < len; i++) b.set(i, q[i & 3]);
        L = len;
    }
    int i;
    for (i = 1; i < K; i++) {
        x = ord[(int64_t)i * ord_stride];
        if (D.pat[x][0] != 0) {   // LGM.cpp:3572-3585
            const int s = D.pat[x][0], e = D.pat[x][1];
            if (L == 0) return ST_ERR_REF_UB;
            const int back = b.get(L - 1);
            if (L + 2 > cap) { *L_out = L; return ST_ERR_BKP_CAPACITY; }
            if (back == -s) { b.set(L, s); b.set(L + 1, e); L += 2; }
            else if (back == e) { b.set(L, -e); b.set(L + 1, -s); L += 2; }
            else break;
        } else if (D.loop[x][0] != 0) {   // LGM.cpp:3586-3644
            const int s = D.loop[x][0], e = D.loop[x][1], cn = D.loop[x][2];
            // reverse search: the last odd slot holding -s that passes its nesting test, else the last one holding e
            int f1 = -1, f2 = -1;
            // one candidate slot: true when the search is over (a -s slot outranks every e slot)
            auto examine = [&](int q, int c) -> bool {
                if (c != -s && c != e) return false;
                if (c == e && f2 >= 0) return false;
                bool skip = false;
                if (q < L - 2) {
                    const int xa = iabs(b.get(q - 1)), ya = iabs(b.get(q + 2));
                    skip = (c == -s) ? (xa < ya) : (xa > ya);
                }
                if (skip) return false;
                if (c == -s) { f1 = q; return true; }
                f2 = q;
                return false;
            };
            {   // odd slots from the top, four per round (independent loads)
                int q = (L - 1) | 1;
                if (q >= L) q -= 2;
                bool done = false;
                for (; !done && q - 6 >= 1; q -= 8) {
                    const int c0 = b.get(q), c1 = b.get(q - 2), c2 = b.get(q - 4), c3 = b.get(q - 6);
                    done = examine(q, c0) || examine(q - 2, c1) || examine(q - 4, c2) || examine(q - 6, c3);
                }
                for (; !done && q >= 1; q -= 2) done = examine(q, b.get(q));
            }
            const bool viaV1 = f1 >= 0;
            const int f = viaV1 ? f1 : f2;
            if (f < 0) break;
            const int cnt = 4 * cn;
            if (L + cnt > cap) { *L_out = L; return ST_ERR_BKP_CAPACITY; }
            const bool hasNext = (f + 1 != L);
            {   // cells [f+1, L) move up by cnt >= 4: four loads, then four stores (the stores of a round land above its loads)
                int q = L - 1;
                for (; q - 3 >= f + 1; q -= 4) {
                    const int v0 = b.get(q), v1 = b.get(q - 1), v2 = b.get(q - 2), v3 = b.get(q - 3);
                    b.set(q + cnt, v0); b.set(q - 1 + cnt, v1); b.set(q - 2 + cnt, v2); b.set(q - 3 + cnt, v3);
                }
                for (; q >= f + 1; q--) b.set(q + cnt, b.get(q));
            }
            int qq[4], fix0, fix1;
            if (viaV1) { qq[0] = s; qq[1] = e; qq[2] = -e; qq[3] = -s; fix0 = -s; fix1 = s; }
            else { qq[0] = -e; qq[1] = -s; qq[2] = s; qq[3] = e; fix0 = e; fix1 = -e; }
            for (int k = 0; k < cnt; k++) b.set(f + 1 + k, qq[k & 3]);
            b.set(f, fix0);
            if (hasNext) b.set(f + 1 + cnt, fix1);
            L += cnt;
        }
    }
    *L_out = L;
    const bool ok = imperfect_fbi_lane(b, L, inv);
    if (!ok && i == K) return ST_ERR_REF_UB;   // (see eval_finish: a stray read on an order that is invalid anyway is harmless)
    return (i == K) ? 1 : 0;
}

}  // namespace ambi
