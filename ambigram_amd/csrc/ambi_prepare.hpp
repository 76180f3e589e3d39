// ambi_prepare.hpp -- the small per-unit scans around the ILP: junction copy numbers, fold-back map, bias,
// indel CN bias, target CN and the BFB DAG.  These stages are order-dependent scans over <= a few thousand
// records (first-come claims in getJuncCN, sequential f64 accumulation, deque chaining in getIndelBias, the
// library sort in constructDAG), so each is executed by ONE thread of the unit's group on data the whole group
// staged into LDS with coalesced loads; throughput comes from running thousands of units concurrently.
//
// Reference: LocalGenomicMap.cpp (LGM.cpp) and localhap.cpp, cited per function.
#pragma once
#include "ambi_common.hpp"
#include "ambi_sort.hpp"

namespace ambi {

// LGM.cpp:3989-4050  getJuncCN.  junc_cn is (n+1) x 2 (row 0 unused), inv_junc[n+1] = junction index or -1.
// Local ids: the unit's interval is [1, n].
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
    // second pass (LGM.cpp:4043-4049): fold-back junctions whose ends are still unrecorded
    for (int ji = 0; ji < m; ji++) {
        const Junction& J = juncs[ji];
        int s = J.src, t = J.tgt;
        if (s < 1 || s > n || t < 1 || t > n) continue;
        if (J.sdir == J.tdir || iabs(s - t) > 2) continue;
        if (inv_junc[s] < 0) inv_junc[s] = ji;
        if (inv_junc[t] < 0) inv_junc[t] = ji;
    }
}

// localhap.cpp:141-146
AMBI_HD int compute_bias(int n, const Junction* juncs, const double* junc_cn, const int32_t* inv_junc) {
    int bias = 1;
    for (int i = 1; i <= n; i++) {
        if (junc_cn[2 * i + 1] > 0) {
            int ji = inv_junc[i];
            if (ji >= 0 && juncs[ji].src != juncs[ji].tgt) bias += int(junc_cn[2 * i + 1]) % 2;
        }
    }
    return bias;
}

// localhap.cpp:150-153: sum of fold-back CN (the no-FBI shortcut test at :164)
AMBI_HD double inversion_cn_sum(int n, const double* junc_cn) {
    double s = 0;
    for (int i = 0; i <= n; i++) s += junc_cn[2 * i + 1];
    return s;
}

// LGM.cpp:3699-3744 getIndelBias.  scratch_sv: m ints, scratch_grp: 2*m+4 ints.  Mutates seg_cn[1..n]
// (seg_cn is indexed by local id, slot 0 unused).
AMBI_HD void get_indel_bias(int n, const Junction* juncs, int m, double* seg_cn, int32_t* sv, int32_t* grp) {
    int nsv = 0;
    for (int ji = 0; ji < m; ji++) {
        const Junction& J = juncs[ji];
        int s = J.src, t = J.tgt;
        if (s < 1 || s > n || t < 1 || t > n) continue;
        if (J.sdir != J.tdir) continue;
        if ((J.sdir > 0 && t - s == 1) || (J.sdir < 0 && s - t == 1)) continue;
        sv[nsv++] = ji;
    }
    while (nsv > 0) {
        int head = m + 2, tail = m + 2;   // group = grp[head, tail)
        int w = 0;
        for (int i = 0; i < nsv; i++) {
            const Junction& J = juncs[sv[i]];
            int s = J.sdir < 0 ? -J.src : J.src, t = J.tdir < 0 ? -J.tgt : J.tgt;
            bool take = true;
            if (head == tail) { grp[tail++] = s; grp[tail++] = t; }
            else if (t == grp[head]) grp[--head] = s;
            else if (s == -grp[head]) grp[--head] = -t;
            else if (grp[tail - 1] == s) grp[tail++] = t;
            else if (grp[tail - 1] == -t) grp[tail++] = -s;
            else take = false;
            if (!take) sv[w++] = sv[i];
        }
        nsv = w;
        int gs = tail - head;
        if (gs == 2) {
            int g0 = grp[head], g1 = grp[head + 1];
            if (g0 < g1) { for (int j = g0 + 1; j < g1; j++) seg_cn[iabs(j)] += 1; }
            else { for (int j = g1; j <= g0; j++) seg_cn[iabs(j)] -= 1; }
        } else {
            for (int j = 1; j < gs - 1; j++) seg_cn[iabs(grp[head + j])] -= 1;
        }
    }
}

// localhap.cpp:222-232.  target_cn[1..n] (local ids) accumulates; caller zeroes it.
AMBI_HD void add_target_cn(const Element* el, int K, int n, int32_t* target_cn) {
    for (int e = 0; e < K; e++) {
        if (el[e].cn <= 0) continue;
        int add = el[e].is_loop ? el[e].cn * 2 : el[e].cn;
        for (int i = el[e].a; i <= el[e].b; i++)
            if (i >= 1 && i <= n) target_cn[i] += add;
    }
}

// ---- std::map<std::string,int> iteration order of the keys "p:A,B" / "l:A,B" (localhap.cpp:122-133) ----
AMBI_HD int key_text(char* out, int is_loop, int A, int B) {
    int p = 0;
    out[p++] = is_loop ? 'l' : 'p';
    out[p++] = ':';
    char tmp[12];
    int t = 0, x = A;
    if (x == 0) tmp[t++] = '0';
    while (x > 0) { tmp[t++] = (char)('0' + x % 10); x /= 10; }
    while (t > 0) out[p++] = tmp[--t];
    out[p++] = ',';
    x = B;
    if (x == 0) tmp[t++] = '0';
    while (x > 0) { tmp[t++] = (char)('0' + x % 10); x /= 10; }
    while (t > 0) out[p++] = tmp[--t];
    out[p] = 0;
    return p;
}

// std::string operator< on the two keys (A,B are ABSOLUTE segment ids)
AMBI_HD bool key_less(int l1, int A1, int B1, int l2, int A2, int B2) {
    char k1[32], k2[32];
    int n1 = key_text(k1, l1, A1, B1), n2 = key_text(k2, l2, A2, B2);
    int n = n1 < n2 ? n1 : n2;
    for (int i = 0; i < n; i++) {
        unsigned char c1 = (unsigned char)k1[i], c2 = (unsigned char)k2[i];
        if (c1 != c2) return c1 < c2;
    }
    return n1 < n2;
}

// LGM.cpp:3276-3378 constructDAG.  `el` = the K solution elements with cn > 0 of this unit (any order, unique
// (kind,a,b)); seg_base turns local ids into the absolute ids the reference's string keys are made of.
// Returns ST_OK or an error status.
AMBI_HD int construct_dag(const Element* el, int K, int seg_base, Dag& D) {
    if (K > kMaxNodes) return ST_ERR_TOO_MANY_NODES;
    D.K = K;
    // nodes in std::map key order: insertion sort of indices by key_less (K <= 64)
    int idx[kMaxNodes];
    for (int i = 0; i < K; i++) {
        int j = i;
        while (j > 0 && key_less(el[i].is_loop, el[i].a + seg_base, el[i].b + seg_base,
                                 el[idx[j - 1]].is_loop, el[idx[j - 1]].a + seg_base, el[idx[j - 1]].b + seg_base)) {
            idx[j] = idx[j - 1];
            j--;
        }
        idx[j] = i;
    }
    Rec3 loops[kMaxNodes];
    for (int i = 0; i < K; i++) {
        const Element& e = el[idx[i]];
        D.succ[i] = 0; D.pred[i] = 0;
        if (!e.is_loop) {
            D.pat[i][0] = e.a; D.pat[i][1] = e.b; D.pat[i][2] = e.cn;
            loops[i].v[0] = 0; loops[i].v[1] = 0; loops[i].v[2] = 0;
        } else {
            D.pat[i][0] = 0; D.pat[i][1] = 0; D.pat[i][2] = 0;
            loops[i].v[0] = e.a; loops[i].v[1] = e.b; loops[i].v[2] = e.cn;
        }
    }
    bool ub = false;
    libstdcxx_sort_loops(loops, K, &ub);   // LGM.cpp:3303
    if (ub) return ST_ERR_REF_UB;
    for (int i = 0; i < K; i++) { D.loop[i][0] = loops[i].v[0]; D.loop[i][1] = loops[i].v[1]; D.loop[i][2] = loops[i].v[2]; }
    auto edge = [&](int i, int j) { D.succ[i] |= (1ull << j); D.pred[j] |= (1ull << i); };
    for (int i = 0; i < K; i++) {
        if (D.pat[i][0] != 0) {
            int d1 = iabs(D.pat[i][0] - D.pat[i][1]);
            for (int j = 0; j < K; j++)   // p1 -> p2
                if (D.pat[j][0] != 0 && (D.pat[i][0] == D.pat[j][0] || D.pat[i][1] == D.pat[j][1]))
                    if (d1 > iabs(D.pat[j][0] - D.pat[j][1])) edge(i, j);
            for (int j = 0; j < K; j++)   // p -> l
                if (D.loop[j][0] != 0 && (D.pat[i][0] == D.loop[j][0] || D.pat[i][1] == D.loop[j][1]))
                    if (d1 > iabs(D.loop[j][0] - D.loop[j][1])) edge(i, j);
        }
    }
    for (int i = 0; i < K; i++) {
        if (D.loop[i][0] != 0) {
            int d1 = iabs(D.loop[i][0] - D.loop[i][1]);
            for (int j = 0; j < K; j++) {   // l -> p
                if (D.pred[i] & (1ull << j)) continue;   // the pattern is a parent of the loop
                if (D.pat[j][0] != 0 && (D.loop[i][0] == D.pat[j][0] || D.loop[i][1] == D.pat[j][1])) {
                    if (d1 > iabs(D.pat[j][0] - D.pat[j][1])) edge(i, j);
                    else {
                        uint64_t par = D.pred[i];
                        bool inherit = false;
                        while (par) {
                            int p = __builtin_ctzll(par);
                            par &= par - 1;
                            if (D.succ[p] & (1ull << j)) { inherit = true; break; }
                        }
                        if (inherit) edge(i, j);
                    }
                }
            }
            for (int j = 0; j < K; j++)   // l1 -> l2
                if (D.loop[j][0] != 0 && (D.loop[i][0] == D.loop[j][0] || D.loop[i][1] == D.loop[j][1]))
                    if (d1 > iabs(D.loop[j][0] - D.loop[j][1])) edge(i, j);
        }
    }
    return ST_OK;
}

}  // namespace ambi
