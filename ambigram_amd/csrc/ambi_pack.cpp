// ambi_pack.cpp -- see ambi_pack.hpp
#include "ambi_pack.hpp"
#include "ambi_wide.hpp"

#include <algorithm>
#include <cstdlib>
#include <map>

namespace ambi {

int HostBatch::add_unit(int n_seg, int seg_base, const double* cn_local, int n_junc, const int32_t* j_src,
                        const int32_t* j_tgt, const int8_t* j_sdir, const int8_t* j_tdir, const double* j_cn, int n_elem,
                        const int32_t* e_is_loop, const int32_t* e_a, const int32_t* e_b, const int32_t* e_cn,
                        int infeasible, int has_components) {
    if (n_seg < 1 || n_seg > kMaxSegLocal || n_junc < 0 || n_junc > 65535 || n_elem < 0) return ST_ERR_BAD_INPUT;   // 16-bit junction indices in LDS
    if (n_elem > kMaxNodesWide) return ST_ERR_TOO_MANY_NODES;   // (64..255: a wide unit, ambi_wide.hpp)
    UnitIn U{};
    U.n_seg = n_seg; U.seg_base = seg_base; U.n_junc = n_junc; U.n_elem = n_elem;
    U.infeasible = infeasible; U.has_components = has_components;
    U.seg_off = (int64_t)seg_cn.size();
    seg_cn.push_back(0.0);
    for (int i = 0; i < n_seg; i++) seg_cn.push_back(cn_local[i]);
    U.junc_off = (int64_t)juncs.size();
    for (int j = 0; j < n_junc; j++) {
        if (j_src[j] < 1 || j_src[j] > n_seg || j_tgt[j] < 1 || j_tgt[j] > n_seg) return ST_ERR_BAD_INPUT;
        Junction J{};
        J.src = j_src[j]; J.tgt = j_tgt[j]; J.sdir = j_sdir[j] > 0 ? 1 : -1; J.tdir = j_tdir[j] > 0 ? 1 : -1;
        J.same_chr = 1; J.cn = j_cn[j];
        juncs.push_back(J);
        junc_ends.push_back(ambi::junc_ends(J));
        junc_cn.push_back(J.cn);
    }
    U.elem_off = (int64_t)elems.size();
    int64_t L = 0;
    for (int e = 0; e < n_elem; e++) {
        if (e_a[e] < 1 || e_b[e] > n_seg || e_a[e] > e_b[e] || e_cn[e] <= 0) return ST_ERR_BAD_INPUT;
        Element E{e_is_loop[e] ? 1 : 0, e_a[e], e_b[e], e_cn[e]};
        elems.push_back(E);
        L += E.is_loop ? 4ll * E.cn : 2;
    }
    // bkp never grows beyond one seed/append per pattern (2 cells) and 4*cn cells per loop (LGM.cpp:3527-3643)
    int64_t bkp_cap = std::max<int64_t>(4, (L + 7) & ~int64_t(7));
    if (bkp_cap > 32760) return ST_ERR_BKP_CAPACITY;
    // every breakpoint pair expands to at most n cells (LGM.cpp:3661-3670).  indelBFB can only GROW the path through
    // a duplication (same-strand SV pointing backwards, LGM.cpp:3794-3805: repeats a stretch) or an insertion group
    // (:3820-3832: adds at most one cell per chained SV); size the head-room accordingly.
    int64_t bound = std::max<int64_t>(64, ((L + 1) / 2) * (int64_t)n_seg);
    int64_t n_sv = 0;
    bool may_dup = false;
    for (int j = 0; j < n_junc; j++) {
        const bool same = (j_sdir[j] > 0) == (j_tdir[j] > 0);
        const bool normal = same && ((j_sdir[j] > 0 && j_tgt[j] - j_src[j] == 1) || (j_sdir[j] <= 0 && j_src[j] - j_tgt[j] == 1));
        const bool fbi = !same && std::abs(j_src[j] - j_tgt[j]) <= 2;
        if (normal || fbi) continue;
        n_sv++;
        any_sv = true;
        if (same) may_dup = true;
    }
    int64_t path_cap = std::min<int64_t>(kPathCapLimit, bound * (may_dup ? 2 : 1) + n_sv + 64);
    path_cap = (path_cap + 7) & ~int64_t(7);
    U.bkp_cap = (int32_t)bkp_cap;
    U.path_cap = (int32_t)path_cap;
    U.out_cap = (int32_t)(bkp_cap + 64);
    U.ideal_cap = ideal_cap;
    U.direct_full = may_dup ? 1 : 0;
    units.push_back(U);
    junc_global.emplace_back();
    max_n = std::max(max_n, n_seg); max_m = std::max(max_m, n_junc); max_k = std::max(max_k, n_elem);
    max_bkp = std::max(max_bkp, U.bkp_cap); max_path = std::max(max_path, U.path_cap); max_out = std::max(max_out, U.out_cap);
    return (int)units.size() - 1;
}

int HostBatch::add_graph_chr(const LhGraph& g, int chr, const SolFile* sol, int block, int n_blocks) {
    if (chr < 0 || chr >= g.n_chr()) return ST_ERR_BAD_INPUT;
    const int s = g.source_ids[chr], e = g.sink_ids[chr];
    if (s < 1 || e > g.n_seg() || s > e) return ST_ERR_BAD_INPUT;
    const int n = e - s + 1, base = s - 1;
    std::vector<double> cn(n);
    for (int i = 0; i < n; i++) cn[i] = g.seg_cn[s - 1 + i];
    // junctions with both ends inside [s,e]; every per-chromosome scan of the reference skips the others
    // (LGM.cpp:3999, :3708, :3755)
    std::vector<int32_t> js, jt, gidx; std::vector<int8_t> jsd, jtd; std::vector<double> jc;
    for (int j = 0; j < g.n_junc(); j++) {
        int a = g.j_src[j], b = g.j_tgt[j];
        if (a < s || a > e || b < s || b > e) continue;
        js.push_back(a - base); jt.push_back(b - base); jsd.push_back(g.j_sdir[j]); jtd.push_back(g.j_tdir[j]);
        jc.push_back(g.j_cn[j]); gidx.push_back(j);
    }
    // solution columns -> elements: elementCN[x] = value, the last line for a column wins (localhap.cpp:204-211)
    std::vector<int32_t> el_loop, el_a, el_b, el_cn;
    int infeasible = 0;
    if (sol) {
        infeasible = sol->infeasible ? 1 : 0;
        std::map<int, int> value;
        const int num_comp = n * (n + 1);   // 2 * numPat columns per graph
        for (size_t i = 0; i < sol->col.size(); i++) {
            int x = sol->col[i];
            if (n_blocks > 0) {              // joint solution: columns of this graph's block, epsilons (x >= numComp*G) excluded (localhap.cpp:536)
                if (x < 0 || x >= num_comp * n_blocks || x / num_comp != block) continue;
                x -= block * num_comp;
            }
            value[x] = sol->val[i];
        }
        for (auto& kv : value) {
            int il, a, b;
            if (kv.second <= 0) continue;
            if (!column_to_element(kv.first, s, e, &il, &a, &b)) continue;   // epsilon / bias columns
            el_loop.push_back(il); el_a.push_back(a - base); el_b.push_back(b - base); el_cn.push_back(kv.second);
        }
    }
    int has_comp = n_blocks > 0 ? 1 : 0;
    for (auto& c : g.components)
        if (!c.empty() && c[0] >= 1 && c[0] <= g.n_seg() && g.seg_partition[c[0] - 1] == chr) has_comp = 1;
    int u = add_unit(n, base, cn.data(), (int)js.size(), js.data(), jt.data(), jsd.data(), jtd.data(), jc.data(),
                     (int)el_a.size(), el_loop.data(), el_a.data(), el_b.data(), el_cn.data(), infeasible, has_comp);
    if (u >= 0) junc_global[u] = gidx;
    return u;
}

int HostBatch::add_unit_from(const HostBatch& src, int u) {
    if (u < 0 || u >= (int)src.units.size()) return ST_ERR_BAD_INPUT;
    UnitIn U = src.units[u];
    const UnitIn& S = src.units[u];
    U.seg_off = (int64_t)seg_cn.size();
    seg_cn.insert(seg_cn.end(), src.seg_cn.begin() + S.seg_off, src.seg_cn.begin() + S.seg_off + S.n_seg + 1);
    U.junc_off = (int64_t)juncs.size();
    juncs.insert(juncs.end(), src.juncs.begin() + S.junc_off, src.juncs.begin() + S.junc_off + S.n_junc);
    junc_ends.insert(junc_ends.end(), src.junc_ends.begin() + S.junc_off, src.junc_ends.begin() + S.junc_off + S.n_junc);
    junc_cn.insert(junc_cn.end(), src.junc_cn.begin() + S.junc_off, src.junc_cn.begin() + S.junc_off + S.n_junc);
    U.elem_off = (int64_t)elems.size();
    elems.insert(elems.end(), src.elems.begin() + S.elem_off, src.elems.begin() + S.elem_off + S.n_elem);
    U.ideal_cap = ideal_cap;
    units.push_back(U);
    junc_global.push_back(src.junc_global[u]);
    if ((size_t)u < src.inject_unit.size() && !src.inject_unit[u].empty()) { inject_unit.resize(units.size()); inject_unit.back() = src.inject_unit[u]; }
    for (int j = 0; j < S.n_junc; j++) {   // (any_sv as add_unit derives it)
        const Junction& J = src.juncs[S.junc_off + j];
        const bool same = J.sdir == J.tdir;
        const bool normal = same && ((J.sdir > 0 && J.tgt - J.src == 1) || (J.sdir <= 0 && J.src - J.tgt == 1));
        const bool fbi = !same && std::abs(J.src - J.tgt) <= 2;
        if (!normal && !fbi) any_sv = true;
    }
    max_n = std::max(max_n, U.n_seg); max_m = std::max(max_m, U.n_junc); max_k = std::max(max_k, U.n_elem);
    max_bkp = std::max(max_bkp, U.bkp_cap); max_path = std::max(max_path, U.path_cap); max_out = std::max(max_out, U.out_cap);
    return (int)units.size() - 1;
}

void HostBatch::finalize() {
    int64_t off = header_bytes();
    off = (off + 15) & ~int64_t(15);
    int64_t islots = 0, sints = 0;
    scratch_off.resize(units.size());
    for (size_t u = 0; u < units.size(); u++) {
        UnitIn& U = units[u];
        UnitLayout L = unit_layout(U.n_seg, U.bkp_cap, U.path_cap, U.out_cap);
        U.res_off = off;
        off += (L.total + 15) & ~int64_t(15);
        U.ideal_cap = ideal_cap;
        U.ideal_off = islots;
        islots += ideal_cap;
        scratch_off[u] = sints;
        sints += 3ll * U.n_junc + 8;
    }
    result_bytes = off;
    // slots for the final paths in run-length form: a run per breakpoint pair at most, twice that where a duplication can copy a
    // stretch of the path (units that go straight to the full finish stage), a few more for the single cells an insertion adds
    run_slot.assign(units.size() + 1, 0);
    for (size_t u = 0; u < units.size(); u++) {
        const UnitIn& U = units[u];
        const int64_t base = U.bkp_cap / 2 + 8;
        int64_t want = U.direct_full ? 2 * base + 32 : base + 8;
        { const char* lim = ambi_env("AMBI_RUN_SLOTS"); if (lim && atoll(lim) > 0 && atoll(lim) < want) want = atoll(lim); }   // tests: small slots, so that the route for paths whose runs outgrow them is reached
        run_slot[u + 1] = run_slot[u] + ((want + 3) & ~int64_t(3));
    }
    wide_index.assign(units.size(), -1);
    n_wide = 0;
    for (size_t u = 0; u < units.size(); u++) if (units[u].n_elem > kMaxNodes) wide_index[u] = n_wide++;
    ideal_slots = islots;
    scratch_ints = sints;
    inject.clear(); inject_off.clear();
    bool any = false;
    for (auto& v : inject_unit) any = any || !v.empty();
    if (any) {
        inject_off.assign(2 * units.size(), -1);
        for (size_t u = 0; u < units.size(); u++) {
            inject_off[2 * u + 1] = 0;
            if (u < inject_unit.size() && !inject_unit[u].empty()) {
                inject_off[2 * u] = (int64_t)inject.size();
                inject_off[2 * u + 1] = (int64_t)inject_unit[u].size();
                inject.insert(inject.end(), inject_unit[u].begin(), inject_unit[u].end());
            }
        }
    }
}

}  // namespace ambi
