// ambi_prepare.hpp -- the per-unit scans around the ILP: junction copy numbers, fold-back map, bias,
// indel CN bias, target CN and the BFB DAG, as SPMD code over one wavefront per unit on records the group staged
// into LDS with coalesced loads.
//
// Order-dependent pieces keep the reference's order: f64 accumulation per junction-CN slot (parallel only when no
// slot receives two contributions, which the .lh reader guarantees; serial otherwise), first-come fold-back claims
// (serial over the compacted fold-back list), deque chaining in getIndelBias ("next matching junction"
// min-reductions), and the library sort inside constructDAG (one thread replays libstdc++, ambi_sort.hpp).
//
// Reference: LocalGenomicMap.cpp (LGM.cpp) and localhap.cpp, cited per function.
#pragma once
#include "ambi_common.hpp"
#include "ambi_group.hpp"
#include "ambi_sort.hpp"

namespace ambi {

AMBI_HD void atomic_or_u64(uint64_t* p, uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicOr((unsigned long long*)p, (unsigned long long)v);
#else
    *p |= v;
#endif
}
AMBI_HD int atomic_inc_i32(int* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return atomicAdd(p, 1);
#else
    int o = *p; *p = o + 1; return o;
#endif
}
// localhap.cpp:150-153: sum of fold-back CN (the no-FBI shortcut test at :164), in the reference's order
AMBI_HD double inversion_cn_sum(int n, const double* junc_cn) {
    double s = 0;
    for (int i = 0; i <= n; i++) s += junc_cn[2 * i + 1];
    return s;
}

// the reference's rounding of junction copy numbers (LGM.cpp:4003-4004)
AMBI_HD double junc_cn_round(double cn) { return (0.5 < cn && cn < 1) ? 1.0 : cn; }

// LGM.cpp:3989-4050 getJuncCN.  junc_cn is (n+1) x 2 (row 0 unused), inv_junc[n+1] = junction index or -1.
// slot_cnt: [n+1] ints, fb: [m] 16-bit junction indices (fold-back junction list; m <= 65535 per unit).  Junction ends come from group memory, the copy
// numbers from the records in HBM (each is read once, by the thread that owns the junction / the slot).
AMBI_HD void atomic_min_slot(int32_t* p, int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMin(p, v);
#else
    if (v < *p) *p = v;
#endif
}

template <class G>
AMBI_HD void get_junc_cn_g(const G& g, int n, const JuncView& J, int m, double* junc_cn, int32_t* inv_junc,
                           int32_t* slot_cnt, uint16_t* fb) {
    for (int i = g.tid(); i <= n; i += g.size()) { junc_cn[2 * i] = 0.0; junc_cn[2 * i + 1] = 0.0; inv_junc[i] = -1; slot_cnt[i] = 0; }
    g.sync();
    // one pass over the junctions: normal (reference-adjacent) ones count into their slot, fold-backs are compacted
    // in junction order
    int nfb = 0;
    for (int base = 0; base < m; base += g.size()) {
        const int ji = base + g.tid();
        int q = 0;
        if (ji < m) {
            const JuncEnds E = J.e[ji];
            const int s = iabs(E.s), t = iabs(E.t);
            const bool same = (E.s < 0) == (E.t < 0);
            if (!(s < 1 || s > n || t < 1 || t > n)) {
                if (same) {
                    if (s + 1 == t) atomic_inc_i32(&slot_cnt[s]);
                    else if (s - 1 == t) atomic_inc_i32(&slot_cnt[t]);
                } else if (iabs(s - t) <= 2) q = 1;
            }
        }
        int tot;
        const int ex = g.exscan_i32(q, &tot);
        if (q) fb[nfb + ex] = (uint16_t)ji;
        nfb += tot;
    }
    g.sync();
    // Normal junctions.  A slot with ONE contribution is 0 + cn, whoever writes it.  A slot with several (an adjacency and a
    // tandem duplication i -> i-1 share one) is a sum in JUNCTION ORDER (f64 addition is not associative): those few
    // contributions are added one after the other, lowest junction index first -- the rounds of this loop run in junction
    // order (base, then q, then thread), and inside a round the pending threads take turns by thread index.
    // Four junctions per thread and round: the copy numbers come from the records in HBM (L2 by now), so the four
    // loads are issued together instead of one round trip per junction.
    for (int base = 0; base < m; base += 4 * g.size()) {
        int slot[4];
        double cn[4];
        bool shared[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int ji = base + q * g.size() + g.tid();
            slot[q] = -1;
            cn[q] = 0.0;
            shared[q] = false;
            if (ji < m) {
                const JuncEnds E = J.e[ji];
                const int s = iabs(E.s), t = iabs(E.t);
                if (!(s < 1 || s > n || t < 1 || t > n || (E.s < 0) != (E.t < 0))) {
                    if (s + 1 == t) slot[q] = s;
                    else if (s - 1 == t) slot[q] = t;
                }
                if (slot[q] >= 0) { cn[q] = J.cn[ji]; shared[q] = slot_cnt[slot[q]] != 1; }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (slot[q] >= 0 && !shared[q]) junc_cn[2 * slot[q]] = 0.0 + junc_cn_round(cn[q]);
            bool pend = shared[q];
            while (g.any(pend)) {
                const int lead = g.first_flag(pend);
                if (g.tid() == lead) { junc_cn[2 * slot[q]] += junc_cn_round(cn[q]); pend = false; }
                g.sync();
            }
        }
    }
    // fold-back claims are first come first served (LGM.cpp:4012-4041): serial over the compacted list.  A slot is
    // claimed at most once, so its copy number is 0 + cn of the claiming junction: assigned in parallel afterwards.
    bool claimed = false;
    if constexpr (G::kLaneArrays) {
        if (nfb <= g.size()) {
            // wavefront form: fold-back k's two ends sit in lane k; the serial walk is a scalar program reading them with
            // v_readlane (one LDS round trip per step, for the slot itself, instead of three dependent ones)
            const int k0 = g.tid();
            int ms = 0, mt = 0, mj = 0;
            if (k0 < nfb) { mj = fb[k0]; const JuncEnds E = J.e[mj]; ms = iabs(E.s); mt = iabs(E.t); }
            for (int k = 0; k < nfb; k++) {
                const int s = g.bcast_i32_u(ms, k), t = g.bcast_i32_u(mt, k), jj = g.bcast_i32_u(mj, k);
                if (inv_junc[s] < 0) { if (g.tid() == 0) inv_junc[s] = jj; }
                else if (inv_junc[t] < 0) { if (g.tid() == 0) inv_junc[t] = jj; }
                g.sync();
            }
            claimed = true;
        }
    }
    if (!claimed && g.tid() == 0) {
        for (int k = 0; k < nfb; k++) {
            const JuncEnds E = J.e[fb[k]];
            const int s = iabs(E.s), t = iabs(E.t);
            if (inv_junc[s] < 0) inv_junc[s] = fb[k];
            else if (inv_junc[t] < 0) inv_junc[t] = fb[k];
        }
    }
    g.sync();
    for (int i = g.tid(); i <= n; i += g.size()) {
        const int ji = inv_junc[i];
        if (ji >= 0) junc_cn[2 * i + 1] = 0.0 + junc_cn_round(J.cn[ji]);
    }
    g.sync();
    // LGM.cpp:4043-4049: remaining ends point at their junction, no copy number.  Serial in the reference, but here the
    // first fold-back (in list order) that touches a still-free end simply gets it -- a minimum over the list positions per
    // end (the slot counters are free by now and hold the claimant's position)
    for (int i = g.tid(); i <= n; i += g.size()) slot_cnt[i] = 0x7fffffff;
    g.sync();
    for (int k = g.tid(); k < nfb; k += g.size()) {
        const JuncEnds E = J.e[fb[k]];
        const int s = iabs(E.s), t = iabs(E.t);
        if (inv_junc[s] < 0) atomic_min_slot(&slot_cnt[s], k);
        if (inv_junc[t] < 0) atomic_min_slot(&slot_cnt[t], k);
    }
    g.sync();
    for (int i = g.tid(); i <= n; i += g.size())
        if (inv_junc[i] < 0 && slot_cnt[i] != 0x7fffffff) inv_junc[i] = fb[slot_cnt[i]];
    g.sync();
}

// localhap.cpp:141-146
template <class G>
AMBI_HD int compute_bias_g(const G& g, int n, const JuncView& J, const double* junc_cn, const int32_t* inv_junc) {
    int part = 0;
    for (int i = 1 + g.tid(); i <= n; i += g.size()) {
        if (junc_cn[2 * i + 1] > 0) {
            const int ji = inv_junc[i];
            if (ji >= 0) { const JuncEnds E = J.e[ji]; if (iabs(E.s) != iabs(E.t)) part += int(junc_cn[2 * i + 1]) % 2; }
        }
    }
    return 1 + g.sum_i32(part);
}

// |sum of fold-back CN| < 1e-6 ?  (localhap.cpp:150-153,164).  All entries are >= 0, so one entry >= 1e-6 settles
// it; only when every entry is tiny is the reference's serial sum replayed.
template <class G>
AMBI_HD bool no_foldback_g(const G& g, int n, const double* junc_cn, double* sum_out) {
    int big = 0, neg = 0;
    for (int i = g.tid(); i <= n; i += g.size()) {
        double v = junc_cn[2 * i + 1];
        big |= (v >= 0.000001);
        neg |= (v < 0);
    }
    const bool any_big = g.any(big != 0), any_neg = g.any(neg != 0);
    if (any_big && !any_neg) { *sum_out = 1.0; return false; }   // sum >= 1e-6 for certain (value itself is not used)
    double s = inversion_cn_sum(n, junc_cn);   // every thread, same order
    *sum_out = s;
    double a = s < 0 ? -s : s;
    return a < 0.000001;
}

// LGM.cpp:3699-3744 getIndelBias.  sv: [m] 16-bit junction indices, taken: [m] bytes, grp: [2m+4] ints.  Mutates seg_cn[1..n].
template <class G>
AMBI_HD void get_indel_bias_g(const G& g, int n, const JuncView& J, int m, double* seg_cn, uint16_t* sv, uint8_t* taken,
                              int32_t* grp) {
    int nsv = 0;
    for (int base = 0; base < m; base += g.size()) {
        int ji = base + g.tid();
        int q = 0;
        if (ji < m) {
            const JuncEnds E = J.e[ji];
            const int s = iabs(E.s), t = iabs(E.t);
            bool in = !(s < 1 || s > n || t < 1 || t > n);
            bool same = (E.s < 0) == (E.t < 0);
            bool normal = (E.s > 0 && t - s == 1) || (E.s < 0 && s - t == 1);
            q = (in && same && !normal) ? 1 : 0;
        }
        int tot;
        int ex = g.exscan_i32(q, &tot);
        if (q) { sv[nsv + ex] = (uint16_t)ji; taken[nsv + ex] = 0; }
        nsv += tot;
    }
    g.sync();
    auto vs = [&](int i) { return (int)J.e[sv[i]].s; };
    auto vt = [&](int i) { return (int)J.e[sv[i]].t; };
    for (int first = 0; first < nsv; first++) {
        if (taken[first]) continue;
        int head = m + 2, tail = m + 2;
        g.sync();
        if (g.tid() == 0) { grp[tail] = vs(first); grp[tail + 1] = vt(first); taken[first] = 1; }
        tail += 2;
        g.sync();
        int cursor = first + 1;
        while (true) {
            int front = grp[head], back = grp[tail - 1];
            int cand = 0x7fffffff;
            for (int i = cursor + g.tid(); i < nsv; i += g.size()) {
                if (taken[i]) continue;
                int s = vs(i), t = vt(i);
                if (t == front || s == -front || back == s || back == -t) { cand = i; break; }
            }
            cand = g.min_i32(cand);
            if (cand == 0x7fffffff) break;
            int s = vs(cand), t = vt(cand);
            int nf = head, nt = tail, wpos, wval;
            if (t == front) { nf = head - 1; wpos = nf; wval = s; }
            else if (s == -front) { nf = head - 1; wpos = nf; wval = -t; }
            else if (back == s) { wpos = tail; wval = t; nt = tail + 1; }
            else { wpos = tail; wval = -s; nt = tail + 1; }
            g.sync();
            if (g.tid() == 0) { grp[wpos] = wval; taken[cand] = 1; }
            head = nf; tail = nt;
            g.sync();
            cursor = cand + 1;
        }
        const int gs = tail - head;
        if (g.tid() == 0) {
            if (gs == 2) {
                int g0 = grp[head], g1 = grp[head + 1];
                if (g0 < g1) { for (int j = g0 + 1; j < g1; j++) seg_cn[iabs(j)] += 1; }
                else { for (int j = g1; j <= g0; j++) seg_cn[iabs(j)] -= 1; }
            } else {
                for (int j = 1; j < gs - 1; j++) seg_cn[iabs(grp[head + j])] -= 1;
            }
        }
        g.sync();
    }
}

// localhap.cpp:222-232.  target_cn[0..n] (local ids): integer adds, so a difference array + running sum gives the
// reference's numbers in any order.  diff: [n+2] ints of group memory; target_cn may be the same array (every index is
// read and then written by one thread).
template <class G>
AMBI_HD void target_cn_g(const G& g, const Element* el, int K, int n, int32_t* target_cn, int32_t* diff) {
    for (int i = g.tid(); i <= n + 1; i += g.size()) diff[i] = 0;
    g.sync();
    for (int e = g.tid(); e < K; e += g.size()) {
        const Element E = el[e];
        if (E.cn <= 0) continue;
        const int lo = E.a < 1 ? 1 : E.a, hi = E.b > n ? n : E.b;
        if (lo > hi) continue;
        const int w = E.is_loop ? E.cn * 2 : E.cn;
        atomic_add_i32(&diff[lo], w);
        atomic_add_i32(&diff[hi + 1], -w);
    }
    g.sync();
    int carry = 0;
    for (int base = 0; base <= n; base += g.size()) {
        const int i = base + g.tid();
        const int v = i <= n ? diff[i] : 0;
        int tot;
        const int ex = g.exscan_i32(v, &tot);
        if (i <= n) target_cn[i] = carry + ex + v;
        carry += tot;
    }
    g.sync();
}

// ---- std::map<std::string,int> iteration order of the keys "p:A,B" / "l:A,B" (localhap.cpp:122-133) ----
// Lexicographic order of decimal strings without building them: pad the shorter number with zeros; if the padded
// values tie, the shorter string is a proper prefix and sorts first (its terminator ',' / end-of-string is below '0').
AMBI_HD int dec_digits(uint32_t x) { int d = 1; while (x >= 10) { x /= 10; d++; } return d; }
AMBI_HD int dec_str_cmp(uint32_t x, uint32_t y) {   // <0, 0, >0
    int dx = dec_digits(x), dy = dec_digits(y);
    uint64_t a = x, b = y;
    for (int i = dx; i < dy; i++) a *= 10;
    for (int i = dy; i < dx; i++) b *= 10;
    if (a != b) return a < b ? -1 : 1;
    return dx - dy;
}
// the same order as one integer per number: the value zero-padded to ten digits, then the digit count
AMBI_HD uint64_t dec_key(uint32_t x) {
    const int d = 1 + (x >= 10u) + (x >= 100u) + (x >= 1000u) + (x >= 10000u) + (x >= 100000u) + (x >= 1000000u) +
                  (x >= 10000000u) + (x >= 100000000u) + (x >= 1000000000u);
    // x * 10^(10-d) without a loop of 64-bit multiplies: the power by a chain of selects, one multiply
    const uint32_t p32 = d >= 10 ? 1u : d == 9 ? 10u : d == 8 ? 100u : d == 7 ? 1000u : d == 6 ? 10000u : d == 5 ? 100000u :
                         d == 4 ? 1000000u : d == 3 ? 10000000u : d == 2 ? 100000000u : 1000000000u;
    const uint64_t v = (uint64_t)x * (uint64_t)p32;
    return (v << 4) | (uint64_t)d;
}
// std::string operator< on the two keys (A,B are ABSOLUTE segment ids)
AMBI_HD bool key_less(int l1, int A1, int B1, int l2, int A2, int B2) {
    if (l1 != l2) return l1 > l2;          // 'l' (0x6C) < 'p' (0x70): loops first
    int c = dec_str_cmp((uint32_t)A1, (uint32_t)A2);
    if (c != 0) return c < 0;
    return dec_str_cmp((uint32_t)B1, (uint32_t)B2) < 0;
}

// Work memory of construct_dag_g (group memory): node order + the record array the library sort permutes.
struct DagScratch {
    int32_t* idx;    // [64]
    Rec3* loops;     // [64]
};

// LGM.cpp:3276-3378 constructDAG.  `el` = the K solution elements with cn > 0 of this unit (any order, unique
// (kind,a,b)); seg_base turns local ids into the absolute ids the reference's string keys are made of.
// Every thread returns the same status.
template <class G>
AMBI_HD int construct_dag_g(const G& g, const Element* el, int K, int seg_base, Dag& D, const DagScratch& W, int64_t* clk = nullptr) {
    if (K > kMaxNodes) return ST_ERR_TOO_MANY_NODES;
    // node numbering = rank of the key in std::map order.  Every element's two decimal keys are computed once and
    // parked in D.succ / D.pred (initialised further down); the K x K comparison loop then only compares integers.
    for (int i = g.tid(); i < K; i += g.size()) {
        D.succ[i] = dec_key((uint32_t)(el[i].a + seg_base));
        D.pred[i] = dec_key((uint32_t)(el[i].b + seg_base));
    }
    g.sync();
    if constexpr (G::kLaneArrays) {
        // wavefront form: element i's keys stay in lane i's registers, element j's arrive by v_readlane
        const int i = g.tid();
        const bool in = i < K;
        const int li = in ? el[i].is_loop : 0;
        const uint64_t ka = in ? D.succ[i] : 0ull, kb = in ? D.pred[i] : 0ull;
        int r = 0;
        for (int j = 0; j < K; j++) {
            const int lj = g.bcast_i32_u(li, j);
            const uint64_t ja = g.bcast_u64(ka, j), jb = g.bcast_u64(kb, j);
            const bool less = (lj != li) ? (lj > li) : ((ja != ka) ? (ja < ka) : (jb < kb));   // key_less(j, i)
            r += (j != i && less) ? 1 : 0;
        }
        if (in) W.idx[r] = i;
    } else
    for (int i = g.tid(); i < K; i += g.size()) {
        const int li = el[i].is_loop;
        const uint64_t ka = D.succ[i], kb = D.pred[i];
        int r = 0;
        for (int j = 0; j < K; j++) {
            const int lj = el[j].is_loop;
            const uint64_t ja = D.succ[j], jb = D.pred[j];
            const bool less = (lj != li) ? (lj > li) : ((ja != ka) ? (ja < ka) : (jb < kb));   // key_less(j, i)
            r += (j != i && less) ? 1 : 0;
        }
        W.idx[r] = i;
    }
    g.sync();
    if (g.tid() == 0) D.K = K;
    g.sync();
    // node tables in map order; the loop records also go to W.loops, from where the sorted order gathers them
    for (int i = g.tid(); i < K; i += g.size()) {
        const Element& e = el[W.idx[i]];
        D.succ[i] = 0; D.pred[i] = 0;
        if (!e.is_loop) {
            D.pat[i][0] = e.a; D.pat[i][1] = e.b; D.pat[i][2] = e.cn;
            W.loops[i].v[0] = 0; W.loops[i].v[1] = 0; W.loops[i].v[2] = 0;
        } else {
            D.pat[i][0] = 0; D.pat[i][1] = 0; D.pat[i][2] = 0;
            W.loops[i].v[0] = e.a; W.loops[i].v[1] = e.b; W.loops[i].v[2] = e.cn;
        }
    }
    g.sync();
    clk_mark(g, clk, 22);
    // std::sort(node2loop, compareLoops) (LGM.cpp:3303): replay of the library's algorithm on one key word per record
    // (ambi_sort.hpp); afterwards position i takes the record whose original position is in the key's low byte
    int ub = 0;
    if constexpr (G::kLaneArrays) {
        // keys and the stack of parked parts in two vector registers, one element per lane; all lanes run the replay
        const int t = g.tid();
        LaneWords keys{t < K ? loop_sort_key(W.loops[t].v[0], W.loops[t].v[1], t) : 0u, t}, stk{0u, t};
        bool u = false;
        libstdcxx_sort_keys(keys, K, &u, stk);
        ub = u ? 1 : 0;
        if (!ub && t < K) { const Rec3 r = W.loops[keys.v & 255u]; D.loop[t][0] = r.v[0]; D.loop[t][1] = r.v[1]; D.loop[t][2] = r.v[2]; }
    } else {
        uint32_t* kw = reinterpret_cast<uint32_t*>(W.idx);   // the node order is consumed: idx is free
        uint32_t* sw = reinterpret_cast<uint32_t*>(D.pred);  // not yet in use (zeroed again below)
        for (int i = g.tid(); i < K; i += g.size()) kw[i] = loop_sort_key(W.loops[i].v[0], W.loops[i].v[1], i);
        g.sync();
        if (g.tid() == 0) {
            MemWords keys{kw}, stk{sw};
            bool u = false;
            libstdcxx_sort_keys(keys, K, &u, stk);
            ub = u ? 1 : 0;
        }
        ub = g.bcast_i32(ub, 0);
        g.sync();
        if (!ub) for (int i = g.tid(); i < K; i += g.size()) { const Rec3 r = W.loops[kw[i] & 255u]; D.loop[i][0] = r.v[0]; D.loop[i][1] = r.v[1]; D.loop[i][2] = r.v[2]; }
        for (int i = g.tid(); i < K; i += g.size()) D.pred[i] = 0;
    }
    if (ub) return ST_ERR_REF_UB;
    g.sync();
    clk_mark(g, clk, 23);
    // p -> p and p -> l edges: static tests, one thread per source node (LGM.cpp:3311-3338)
    for (int i = g.tid(); i < K; i += g.size()) {
        if (D.pat[i][0] == 0) continue;
        const int d1 = iabs(D.pat[i][0] - D.pat[i][1]);
        uint64_t out = 0;
        for (int j = 0; j < K; j++) {
            if (D.pat[j][0] != 0 && (D.pat[i][0] == D.pat[j][0] || D.pat[i][1] == D.pat[j][1]) && d1 > iabs(D.pat[j][0] - D.pat[j][1])) out |= (1ull << j);
            if (D.loop[j][0] != 0 && (D.pat[i][0] == D.loop[j][0] || D.pat[i][1] == D.loop[j][1]) && d1 > iabs(D.loop[j][0] - D.loop[j][1])) out |= (1ull << j);
        }
        D.succ[i] = out;
        uint64_t o = out;
        while (o) { int j = __builtin_ctzll(o); o &= o - 1; atomic_or_u64(&D.pred[j], 1ull << i); }
    }
    g.sync();
    clk_mark(g, clk, 24);
    // loops in index order (LGM.cpp:3339-3377).  The "inherited from a parent" rule reads pred[i] and succ[parent] as
    // they stand when loop i is processed, so the loop over i stays sequential -- but it needs no barrier: thread j
    // owns column j (pred[j]: every edge into j, written by nobody else in this loop), "succ[parent] has j" is the
    // same fact as "pred[j] has parent", and pred[i] is handed round by its owner.  Per step: l -> p over all j with
    // the state before the step, then l -> l (static test), exactly the reference's two inner loops.
    if constexpr (G::kLaneArrays) {
        // wavefront form (K <= 63 < 64 lanes): node j lives in lane j's registers for the whole loop -- its two records,
        // its column pred[j] -- and loop i's record and pred[i] come from lane i through v_readlane; the edges of step i
        // are one ballot.  No memory access inside the loop.
        const int j = g.tid();
        const bool in = j < K;
        uint64_t mine = in ? D.pred[j] : 0ull, row = 0;
        const int pa = in ? D.pat[j][0] : 0, pb = in ? D.pat[j][1] : 0, qa = in ? D.loop[j][0] : 0, qb = in ? D.loop[j][1] : 0;
        for (int i = 0; i < K; i++) {
            const int la = g.bcast_i32_u(qa, i), lb = g.bcast_i32_u(qb, i);
            if (la == 0) continue;
            const int d1 = iabs(la - lb);
            const uint64_t pred_i = g.bcast_u64(mine, i);
            bool edge = false;
            if (!((pred_i >> j) & 1ull) && pa != 0 && (la == pa || lb == pb)) edge = d1 > iabs(pa - pb) || (pred_i & mine) != 0;
            if (qa != 0 && (la == qa || lb == qb) && d1 > iabs(qa - qb)) edge = true;
            edge = edge && in;
            if (edge) mine |= 1ull << i;
            const uint64_t col = g.ballot_u64(edge);
            if (j == i) row = col;
        }
        if (in) { D.pred[j] = mine; D.succ[j] |= row; }
        g.sync();
        return ST_OK;
    }
    for (int i = 0; i < K; i++) {
        const int la = D.loop[i][0], lb = D.loop[i][1];
        if (la == 0) continue;
        const int d1 = iabs(la - lb);
        const int owner = i % g.size();
        const uint64_t pred_i = g.bcast_u64(g.tid() == owner ? D.pred[i] : 0ull, owner);
        for (int j = g.tid(); j < K; j += g.size()) {
            const uint64_t mine = D.pred[j];
            const int pa = D.pat[j][0], pb = D.pat[j][1], qa = D.loop[j][0], qb = D.loop[j][1];
            bool edge = false;
            if (!((pred_i >> j) & 1ull) && pa != 0 && (la == pa || lb == pb))                  // l -> p
                edge = d1 > iabs(pa - pb) || (pred_i & mine) != 0;                             //   own size, or inherited from a parent
            if (qa != 0 && (la == qa || lb == qb) && d1 > iabs(qa - qb)) edge = true;          // l1 -> l2
            if (edge) { D.pred[j] = mine | (1ull << i); atomic_or_u64(&D.succ[i], 1ull << j); }
        }
    }
    g.sync();
    return ST_OK;
}

}  // namespace ambi
