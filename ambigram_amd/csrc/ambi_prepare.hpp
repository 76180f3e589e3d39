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

// ---- serial forms (one thread); the group forms below fall back to them where order matters ----

// LGM.cpp:3989-4050  getJuncCN.  junc_cn is (n+1) x 2 (row 0 unused), inv_junc[n+1] = junction index or -1.
AMBI_HD void get_junc_cn(int n, const Junction* juncs, int m, double* junc_cn, int32_t* inv_junc) {
    for (int i = 0; i <= n; i++) { junc_cn[2 * i] = 0.0; junc_cn[2 * i + 1] = 0.0; inv_junc[i] = -1; }
    for (int ji = 0; ji < m; ji++) {
        const Junction& J = juncs[ji];
        int s = J.src, t = J.tgt;
        if (s < 1 || s > n || t < 1 || t > n) continue;
        double cn = J.cn;
        if (0.5 < cn && cn < 1) cn = 1;
        if (J.sdir == J.tdir) {
            if (s + 1 == t) junc_cn[2 * s] += cn;
            else if (s - 1 == t) junc_cn[2 * t] += cn;
        } else if (iabs(s - t) <= 2) {
            if (inv_junc[s] < 0) { inv_junc[s] = ji; junc_cn[2 * s + 1] += cn; }
            else if (inv_junc[t] < 0) { inv_junc[t] = ji; junc_cn[2 * t + 1] += cn; }
        }
    }
    for (int ji = 0; ji < m; ji++) {   // LGM.cpp:4043-4049
        const Junction& J = juncs[ji];
        int s = J.src, t = J.tgt;
        if (s < 1 || s > n || t < 1 || t > n) continue;
        if (J.sdir == J.tdir || iabs(s - t) > 2) continue;
        if (inv_junc[s] < 0) inv_junc[s] = ji;
        if (inv_junc[t] < 0) inv_junc[t] = ji;
    }
}

// localhap.cpp:150-153: sum of fold-back CN (the no-FBI shortcut test at :164), in the reference's order
AMBI_HD double inversion_cn_sum(int n, const double* junc_cn) {
    double s = 0;
    for (int i = 0; i <= n; i++) s += junc_cn[2 * i + 1];
    return s;
}

// ---- group forms ----

// getJuncCN.  slot_cnt: [n+1] ints, fb: [m] ints (fold-back junction list).
template <class G>
AMBI_HD void get_junc_cn_g(const G& g, int n, const Junction* juncs, int m, double* junc_cn, int32_t* inv_junc,
                           int32_t* slot_cnt, int32_t* fb) {
    for (int i = g.tid(); i <= n; i += g.size()) { junc_cn[2 * i] = 0.0; junc_cn[2 * i + 1] = 0.0; inv_junc[i] = -1; slot_cnt[i] = 0; }
    g.sync();
    // normal (reference-adjacent) junctions: which slot, and does any slot get two of them?
    for (int ji = g.tid(); ji < m; ji += g.size()) {
        const Junction& J = juncs[ji];
        int s = J.src, t = J.tgt;
        if (s < 1 || s > n || t < 1 || t > n || J.sdir != J.tdir) continue;
        if (s + 1 == t) atomic_inc_i32(&slot_cnt[s]);
        else if (s - 1 == t) atomic_inc_i32(&slot_cnt[t]);
    }
    g.sync();
    int multi = 0;
    for (int i = g.tid(); i <= n; i += g.size()) multi |= (slot_cnt[i] > 1);
    const bool serial_normal = g.any(multi != 0);
    if (!serial_normal) {
        for (int ji = g.tid(); ji < m; ji += g.size()) {
            const Junction& J = juncs[ji];
            int s = J.src, t = J.tgt;
            if (s < 1 || s > n || t < 1 || t > n || J.sdir != J.tdir) continue;
            double cn = J.cn;
            if (0.5 < cn && cn < 1) cn = 1;
            if (s + 1 == t) junc_cn[2 * s] = 0.0 + cn;
            else if (s - 1 == t) junc_cn[2 * t] = 0.0 + cn;
        }
    } else if (g.tid() == 0) {
        for (int ji = 0; ji < m; ji++) {
            const Junction& J = juncs[ji];
            int s = J.src, t = J.tgt;
            if (s < 1 || s > n || t < 1 || t > n || J.sdir != J.tdir) continue;
            double cn = J.cn;
            if (0.5 < cn && cn < 1) cn = 1;
            if (s + 1 == t) junc_cn[2 * s] += cn;
            else if (s - 1 == t) junc_cn[2 * t] += cn;
        }
    }
    // fold-back junctions, compacted in junction order; the first-come claims stay serial (LGM.cpp:4012-4049)
    int nfb = 0;
    for (int base = 0; base < m; base += g.size()) {
        int ji = base + g.tid();
        int q = 0;
        if (ji < m) {
            const Junction& J = juncs[ji];
            int s = J.src, t = J.tgt;
            q = (!(s < 1 || s > n || t < 1 || t > n) && J.sdir != J.tdir && iabs(s - t) <= 2) ? 1 : 0;
        }
        int tot;
        int ex = g.exscan_i32(q, &tot);
        if (q) fb[nfb + ex] = ji;
        nfb += tot;
    }
    g.sync();
    if (g.tid() == 0) {
        for (int k = 0; k < nfb; k++) {
            const Junction& J = juncs[fb[k]];
            int s = J.src, t = J.tgt;
            double cn = J.cn;
            if (0.5 < cn && cn < 1) cn = 1;
            if (inv_junc[s] < 0) { inv_junc[s] = fb[k]; junc_cn[2 * s + 1] += cn; }
            else if (inv_junc[t] < 0) { inv_junc[t] = fb[k]; junc_cn[2 * t + 1] += cn; }
        }
        for (int k = 0; k < nfb; k++) {
            const Junction& J = juncs[fb[k]];
            if (inv_junc[J.src] < 0) inv_junc[J.src] = fb[k];
            if (inv_junc[J.tgt] < 0) inv_junc[J.tgt] = fb[k];
        }
    }
    g.sync();
}

// localhap.cpp:141-146
template <class G>
AMBI_HD int compute_bias_g(const G& g, int n, const Junction* juncs, const double* junc_cn, const int32_t* inv_junc) {
    int part = 0;
    for (int i = 1 + g.tid(); i <= n; i += g.size()) {
        if (junc_cn[2 * i + 1] > 0) {
            int ji = inv_junc[i];
            if (ji >= 0 && juncs[ji].src != juncs[ji].tgt) part += int(junc_cn[2 * i + 1]) % 2;
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

// LGM.cpp:3699-3744 getIndelBias.  sv: [m] ints, taken: [m] bytes, grp: [2m+4] ints.  Mutates seg_cn[1..n].
template <class G>
AMBI_HD void get_indel_bias_g(const G& g, int n, const Junction* juncs, int m, double* seg_cn, int32_t* sv, uint8_t* taken,
                              int32_t* grp) {
    int nsv = 0;
    for (int base = 0; base < m; base += g.size()) {
        int ji = base + g.tid();
        int q = 0;
        if (ji < m) {
            const Junction& J = juncs[ji];
            int s = J.src, t = J.tgt;
            bool in = !(s < 1 || s > n || t < 1 || t > n);
            bool normal = (J.sdir > 0 && t - s == 1) || (J.sdir < 0 && s - t == 1);
            q = (in && J.sdir == J.tdir && !normal) ? 1 : 0;
        }
        int tot;
        int ex = g.exscan_i32(q, &tot);
        if (q) { sv[nsv + ex] = ji; taken[nsv + ex] = 0; }
        nsv += tot;
    }
    g.sync();
    auto vs = [&](int i) { const Junction& J = juncs[sv[i]]; return J.sdir < 0 ? -J.src : J.src; };
    auto vt = [&](int i) { const Junction& J = juncs[sv[i]]; return J.tdir < 0 ? -J.tgt : J.tgt; };
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

// localhap.cpp:222-232.  target_cn[0..n] (local ids)
template <class G>
AMBI_HD void target_cn_g(const G& g, const Element* el, int K, int n, int32_t* target_cn) {
    for (int i = g.tid(); i <= n; i += g.size()) {
        int acc = 0;
        for (int e = 0; e < K; e++)
            if (el[e].cn > 0 && i >= 1 && el[e].a <= i && i <= el[e].b) acc += el[e].is_loop ? el[e].cn * 2 : el[e].cn;
        target_cn[i] = acc;
    }
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
AMBI_HD int construct_dag_g(const G& g, const Element* el, int K, int seg_base, Dag& D, const DagScratch& W) {
    if (K > kMaxNodes) return ST_ERR_TOO_MANY_NODES;
    // node numbering = rank of the key in std::map order
    for (int i = g.tid(); i < K; i += g.size()) {
        int r = 0;
        for (int j = 0; j < K; j++)
            if (j != i && key_less(el[j].is_loop, el[j].a + seg_base, el[j].b + seg_base, el[i].is_loop, el[i].a + seg_base, el[i].b + seg_base)) r++;
        W.idx[r] = i;
    }
    if (g.tid() == 0) D.K = K;
    g.sync();
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
    int ub = 0;
    if (g.tid() == 0) {
        bool u = false;
        libstdcxx_sort_loops(W.loops, K, &u);   // LGM.cpp:3303
        ub = u ? 1 : 0;
    }
    ub = g.bcast_i32(ub, 0);
    if (ub) return ST_ERR_REF_UB;
    g.sync();
    for (int i = g.tid(); i < K; i += g.size()) { D.loop[i][0] = W.loops[i].v[0]; D.loop[i][1] = W.loops[i].v[1]; D.loop[i][2] = W.loops[i].v[2]; }
    g.sync();
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
    // loops in index order (LGM.cpp:3339-3377): the "inherited from a parent" rule reads pred[i] and succ[parent] as
    // they stand when loop i is processed, so the loop over i stays sequential; the scan over j is parallel.
    for (int i = 0; i < K; i++) {
        if (D.loop[i][0] == 0) continue;
        const int d1 = iabs(D.loop[i][0] - D.loop[i][1]);
        const uint64_t pred_i = D.pred[i];
        g.sync();
        for (int j = g.tid(); j < K; j += g.size()) {   // l -> p
            if (pred_i & (1ull << j)) continue;
            if (D.pat[j][0] != 0 && (D.loop[i][0] == D.pat[j][0] || D.loop[i][1] == D.pat[j][1])) {
                bool add = d1 > iabs(D.pat[j][0] - D.pat[j][1]);
                if (!add) {
                    uint64_t par = pred_i;
                    while (par) { int p = __builtin_ctzll(par); par &= par - 1; if (D.succ[p] & (1ull << j)) { add = true; break; } }
                }
                if (add) { atomic_or_u64(&D.succ[i], 1ull << j); atomic_or_u64(&D.pred[j], 1ull << i); }
            }
        }
        g.sync();
        for (int j = g.tid(); j < K; j += g.size()) {   // l1 -> l2
            if (D.loop[j][0] != 0 && (D.loop[i][0] == D.loop[j][0] || D.loop[i][1] == D.loop[j][1]) && d1 > iabs(D.loop[j][0] - D.loop[j][1])) {
                atomic_or_u64(&D.succ[i], 1ull << j);
                atomic_or_u64(&D.pred[j], 1ull << i);
            }
        }
        g.sync();
    }
    return ST_OK;
}

}  // namespace ambi
