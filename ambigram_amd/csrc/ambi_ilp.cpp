// ambi_ilp.cpp -- see ambi_ilp.hpp
#include "ambi_ilp.hpp"
#include "ambi_ilp_rows.hpp"

#include <cfloat>
#include <cstdio>
#include <set>
#include <utility>

namespace ambi {

namespace {
struct Tri {   // index arithmetic of the (a,b), s <= a <= b <= e triangle in lexicographic order
    int s, e, n, num_pat;
    Tri(int s_, int e_) : s(s_), e(e_), n(e_ - s_ + 1), num_pat((e_ - s_ + 1) * (e_ - s_ + 2) / 2) {}
    int P(int a, int b) const { int da = a - s; return da * n - da * (da - 1) / 2 + (b - a); }
    int L(int a, int b) const { return num_pat + P(a, b); }
};
}  // namespace

void build_bfb_ilp(int s, int e, const double* seg_cn, const double* fold_cn, int bias, double max_cn_total,
                   const std::vector<std::vector<int32_t>>& components, bool juncs_info, IlpModel& m) {
    const double INF = DBL_MAX;
    const Tri T(s, e);
    const int n = T.n, num_pat = T.num_pat, num_el = 2 * num_pat, num_eps = 2 * n;
    const int num_var = num_el + num_eps + 1;
    m = IlpModel();
    m.n_cols = num_var;
    m.n_int = num_el;
    m.row_ptr.push_back(0);
    auto put = [&](int c, double v) { m.col.push_back(c); m.val.push_back(v); };
    auto end_row = [&](double lo, double up) { m.row_ptr.push_back((int64_t)m.col.size()); m.row_lo.push_back(lo); m.row_up.push_back(up); };

    // -- per segment: CN fit +-eps (LGM.cpp:4426-4451) and fold-back fit +-eps (LGM.cpp:4453-4494)
    for (int i = s; i <= e; i++) {
        const int k = i - s;
        for (int rep = 0; rep < 2; rep++) {
            // elements covering i, in (a,b) order: patterns (coef 1), then loops (coef 2)
            for (int a = s; a <= i; a++) for (int b = i; b <= e; b++) put(T.P(a, b), 1);
            for (int a = s; a <= i; a++) for (int b = i; b <= e; b++) put(T.L(a, b), 2);
            put(num_el + 2 * k, rep == 0 ? 1 : -1);
            if (rep == 0) end_row(seg_cn[k], INF); else end_row(-INF, seg_cn[k]);
        }
        for (int rep = 0; rep < 2; rep++) {
            // patterns starting at i (when another one exists) and ending at i get 0.5, in column order
            // (column order = (a,b) lexicographic: rows a < i first, then the a == i row)
            if (i > s) for (int a = s; a < i; a++) put(T.P(a, i), 0.5);
            if (i < e) for (int b = i; b <= e; b++) put(T.P(i, b), 0.5);
            else if (i > s) put(T.P(i, i), 0.5);
            // loops with an end at i: coef 1, column order
            for (int a = s; a < i; a++) put(T.L(a, i), 1);
            for (int b = i; b <= e; b++) put(T.L(i, b), 1);
            put(num_el + 2 * k + 1, rep == 0 ? 1 : -1);
            if (rep == 0) end_row(fold_cn[k], INF); else end_row(-INF, fold_cn[k]);
        }
    }
    put(num_var - 1, 1); end_row(bias, bias);   // LGM.cpp:4498-4503

    // -- pattern nesting (LGM.cpp:4544-4583)
    for (int a = s; a <= e; a++) for (int b = a; b <= e; b++) {
        if (a > s || b < e) {
            for (int j = s; j < a; j++) put(T.P(j, b), 1);
            for (int j = b + 1; j <= e; j++) put(T.P(a, j), 1);
            put(T.P(a, b), -1); end_row(0, INF);
        }
        if (a < b) {
            for (int j = a; j < b; j++) put(T.P(a, j), 1);
            for (int j = a + 1; j <= b; j++) put(T.P(j, b), 1);
            put(T.P(a, b), 1); end_row(0, 2);
        }
    }
    // -- child loops need a parent (LGM.cpp:4587-4612)
    for (int a = s; a <= e; a++) for (int b = a; b <= e; b++) {
        if (a > s || b < e) {
            for (int j = s; j < a; j++) { put(T.P(j, b), 1); put(T.L(j, b), 1); }
            for (int j = b + 1; j <= e; j++) { put(T.P(a, j), 1); put(T.L(a, j), 1); }
            put(T.L(a, b), -1); end_row(0, INF);
        }
    }
    // -- loop + child loops (LGM.cpp:4615-4646)
    for (int a = s; a <= e; a++) for (int b = a + 1; b <= e; b++) {
        for (int rep = 0; rep < 2; rep++) {
            for (int j = a; j < b; j++) put(T.L(a, j), 1);
            for (int j = a + 1; j <= b; j++) put(T.L(j, b), 1);
            put(rep == 0 ? T.L(a, b) : T.P(a, b), 1); end_row(0, 2);
        }
    }
    // -- pattern + child loops/patterns (LGM.cpp:4649-4681)
    for (int a = s; a <= e; a++) for (int b = a + 1; b <= e; b++) {
        for (int j = a; j < b; j++) put(T.L(a, j), 1);
        for (int j = a + 1; j <= b; j++) put(T.P(j, b), 1);
        put(T.P(a, b), 1); end_row(0, 2);
        for (int j = a; j < b; j++) put(T.P(a, j), 1);
        for (int j = a + 1; j <= b; j++) put(T.L(j, b), 1);
        put(T.P(a, b), 1); end_row(0, 2);
    }
    // -- linked/long-read components (LGM.cpp:4684-4703)
    if (!components.empty() && juncs_info) {
        std::set<std::pair<int, int>> seen;
        for (auto& c : components) {
            if (c.empty()) continue;
            int lo = c.front() < c.back() ? c.front() : c.back(), hi = c.front() < c.back() ? c.back() : c.front();
            if (lo == s && hi == e) continue;
            if (!seen.insert({lo, hi}).second) continue;
            bool in = lo >= s && hi <= e;
            put(in ? T.L(lo, hi) : 0, 1);
            put(in ? T.P(lo, hi) : 0, 1);
        }
        end_row(0, 5);
    }
    // -- bounds, objective (LGM.cpp:4708-4747)
    m.col_lo.assign(num_var, 0); m.col_up.assign(num_var, INF); m.obj.assign(num_var, 0);
    for (int c = 0; c < num_pat; c++) m.col_up[c] = 1;
    for (int c = num_pat; c < num_el; c++) m.col_up[c] = max_cn_total;
    for (int c = num_el; c < num_var - 1; c++) m.obj[c] = 1;
    m.col_lo[num_var - 1] = m.col_up[num_var - 1] = bias;
    m.obj[num_var - 1] = -1;
}

// Joint model of G graphs over the same chromosome (LocalGenomicMap::BFB_ILP_SC, LGM.cpp:4754-5093): per graph g the
// rows of BFB_ILP without the bias row and without the .juncs row, on the column block [g*numComp, (g+1)*numComp); the
// +-epsilon column of a row pair is numElements + idx/2 where idx is the RUNNING ROW COUNTER (LGM.cpp:4815, :4821, :4858,
// :4864) -- for g = 0 that is epsilon 2k / 2k+1 of segment k as in BFB_ILP, for g > 0 the counter has also run over the
// nesting rows of the graphs before, so those epsilons land further up (inside the block the reference sets aside for
// the linking epsilons; the matrix is reproduced as the reference builds it).  Then, for every pair (i, j) of the
// `evolution` lists and every element, the two rows  x_i - x_j +- eps  with eps = cnt/2, cnt starting at
// 2*(numElements + 2*n*G) (LGM.cpp:5032-5072).  Bounds: loops <= CN sum of THIS graph's segments start..end (:5012-5027).
void build_bfb_ilp_sc(int s, int e, int G, const double* seg_cn, const double* fold_cn, const std::vector<std::pair<int, int>>& evolution,
                      IlpModel& m) {
    const double INF = DBL_MAX;
    const Tri T(s, e);
    const int n = T.n, num_pat = T.num_pat, num_comp = 2 * num_pat;
    const int num_el = num_comp * G, num_eps = n * 2 * G + (G * (G - 1)) * num_comp;
    const int num_var = num_el + num_eps;
    m = IlpModel();
    m.n_cols = num_var;
    m.n_int = num_el;
    m.row_ptr.push_back(0);
    int64_t idx = 0;
    auto put = [&](int c, double v) { m.col.push_back(c); m.val.push_back(v); };
    auto end_row = [&](double lo, double up) { m.row_ptr.push_back((int64_t)m.col.size()); m.row_lo.push_back(lo); m.row_up.push_back(up); idx++; };
    m.col_lo.assign(num_var, 0); m.col_up.assign(num_var, INF); m.obj.assign(num_var, 0);
    for (int g = 0; g < G; g++) {
        const int off = g * num_comp;
        const double* cn = seg_cn + (size_t)g * n;
        const double* fold = fold_cn + (size_t)g * n;
        auto P = [&](int a, int b) { return off + T.P(a, b); };
        auto L = [&](int a, int b) { return off + T.L(a, b); };
        for (int i = s; i <= e; i++) {
            const int k = i - s;
            for (int rep = 0; rep < 2; rep++) {
                for (int a = s; a <= i; a++) for (int b = i; b <= e; b++) put(P(a, b), 1);
                for (int a = s; a <= i; a++) for (int b = i; b <= e; b++) put(L(a, b), 2);
                put(num_el + (int)(idx / 2), rep == 0 ? 1 : -1);
                if (rep == 0) end_row(cn[k], INF); else end_row(-INF, cn[k]);
            }
            for (int rep = 0; rep < 2; rep++) {
                if (i > s) for (int a = s; a < i; a++) put(P(a, i), 0.5);
                if (i < e) for (int b = i; b <= e; b++) put(P(i, b), 0.5);
                else if (i > s) put(P(i, i), 0.5);
                for (int a = s; a < i; a++) put(L(a, i), 1);
                for (int b = i; b <= e; b++) put(L(i, b), 1);
                put(num_el + (int)(idx / 2), rep == 0 ? 1 : -1);
                if (rep == 0) end_row(fold[k], INF); else end_row(-INF, fold[k]);
            }
        }
        for (int a = s; a <= e; a++) for (int b = a; b <= e; b++) {          // LGM.cpp:4867-4911
            if (a > s || b < e) {
                for (int j = s; j < a; j++) put(P(j, b), 1);
                for (int j = b + 1; j <= e; j++) put(P(a, j), 1);
                put(P(a, b), -1); end_row(0, INF);
            }
            if (a < b) {
                for (int j = a; j < b; j++) put(P(a, j), 1);
                for (int j = a + 1; j <= b; j++) put(P(j, b), 1);
                put(P(a, b), 1); end_row(0, 2);
            }
        }
        for (int a = s; a <= e; a++) for (int b = a; b <= e; b++) {          // :4914-4940
            if (a > s || b < e) {
                for (int j = s; j < a; j++) { put(P(j, b), 1); put(L(j, b), 1); }
                for (int j = b + 1; j <= e; j++) { put(P(a, j), 1); put(L(a, j), 1); }
                put(L(a, b), -1); end_row(0, INF);
            }
        }
        for (int a = s; a <= e; a++) for (int b = a + 1; b <= e; b++) {      // :4943-4974
            for (int rep = 0; rep < 2; rep++) {
                for (int j = a; j < b; j++) put(L(a, j), 1);
                for (int j = a + 1; j <= b; j++) put(L(j, b), 1);
                put(rep == 0 ? L(a, b) : P(a, b), 1); end_row(0, 2);
            }
        }
        for (int a = s; a <= e; a++) for (int b = a + 1; b <= e; b++) {      // :4977-5008
            for (int j = a; j < b; j++) put(L(a, j), 1);
            for (int j = a + 1; j <= b; j++) put(P(j, b), 1);
            put(P(a, b), 1); end_row(0, 2);
            for (int j = a; j < b; j++) put(P(a, j), 1);
            for (int j = a + 1; j <= b; j++) put(L(j, b), 1);
            put(P(a, b), 1); end_row(0, 2);
        }
        double max_cn = 0;
        for (int k = 0; k < n; k++) max_cn += cn[k];
        for (int c = 0; c < num_pat; c++) m.col_up[off + c] = 1;
        for (int c = num_pat; c < num_comp; c++) m.col_up[off + c] = max_cn;
    }
    int64_t cnt = ((int64_t)num_el + (int64_t)n * 2 * G) * 2;
    for (auto& pr : evolution) {
        for (int c = 0; c < num_comp; c++) {      // patterns k = 0..numPat-1, then loops: column order within a block
            for (int rep = 0; rep < 2; rep++) {
                put(c + num_comp * pr.first, 1); put(c + num_comp * pr.second, -1); put((int)(cnt / 2), rep == 0 ? 1 : -1);
                if (rep == 0) end_row(0, INF); else end_row(-INF, 0);
                cnt++;
            }
        }
    }
    for (int c = num_el; c < num_var; c++) m.obj[c] = 1;
}

// The same model as ROW DESCRIPTORS (ambi_ilp_rows.hpp): everything except col/val, which the caller fills entry by
// entry with ilp_row_entry -- on the GPU with one thread per non-zero.  O(rows) on the host.
void build_bfb_ilp_rows(int s, int e, const double* seg_cn, const double* fold_cn, int bias, double max_cn_total,
                        const std::vector<std::vector<int32_t>>& components, bool juncs_info, IlpModel& m,
                        std::vector<IlpRowDesc>& rows, std::vector<int32_t>& lit_col, std::vector<double>& lit_val) {
    const double INF = DBL_MAX;
    const IlpGeom G = ilp_geom(s, e);
    const int n = G.n, num_pat = G.num_pat, num_el = G.num_el, num_var = num_el + 2 * n + 1;
    m = IlpModel();
    m.n_cols = num_var;
    m.n_int = num_el;
    rows.clear(); lit_col.clear(); lit_val.clear();
    m.row_ptr.push_back(0);
    auto row = [&](int family, int a, int b, int rep, double lo, double up) {
        IlpRowDesc d{family, a, b, rep};
        rows.push_back(d);
        m.row_ptr.push_back(m.row_ptr.back() + ilp_row_len(d, G));
        m.row_lo.push_back(lo); m.row_up.push_back(up);
    };
    for (int i = s; i <= e; i++) {
        const int k = i - s;
        row(ILP_CN, i, 0, 0, seg_cn[k], INF); row(ILP_CN, i, 0, 1, -INF, seg_cn[k]);
        row(ILP_FB, i, 0, 0, fold_cn[k], INF); row(ILP_FB, i, 0, 1, -INF, fold_cn[k]);
    }
    row(ILP_BIAS, 0, 0, 0, bias, bias);
    for (int a = s; a <= e; a++) for (int b = a; b <= e; b++) {
        if (a > s || b < e) row(ILP_PA, a, b, 0, 0, INF);
        if (a < b) row(ILP_PB, a, b, 0, 0, 2);
    }
    for (int a = s; a <= e; a++) for (int b = a; b <= e; b++)
        if (a > s || b < e) row(ILP_LA, a, b, 0, 0, INF);
    for (int a = s; a <= e; a++) for (int b = a + 1; b <= e; b++) { row(ILP_LL, a, b, 0, 0, 2); row(ILP_LL, a, b, 1, 0, 2); }
    for (int a = s; a <= e; a++) for (int b = a + 1; b <= e; b++) { row(ILP_PC, a, b, 0, 0, 2); row(ILP_PC, a, b, 1, 0, 2); }
    if (!components.empty() && juncs_info) {
        std::set<std::pair<int, int>> seen;
        for (auto& c : components) {
            if (c.empty()) continue;
            int lo = c.front() < c.back() ? c.front() : c.back(), hi = c.front() < c.back() ? c.back() : c.front();
            if (lo == s && hi == e) continue;
            if (!seen.insert({lo, hi}).second) continue;
            bool in = lo >= s && hi <= e;
            lit_col.push_back(in ? G.L(lo, hi) : 0); lit_val.push_back(1);
            lit_col.push_back(in ? G.P(lo, hi) : 0); lit_val.push_back(1);
        }
        row(ILP_LIT, 0, (int)lit_col.size(), 0, 0, 5);
    }
    m.col.assign((size_t)m.row_ptr.back(), 0);
    m.val.assign((size_t)m.row_ptr.back(), 0.0);
    m.col_lo.assign(num_var, 0); m.col_up.assign(num_var, INF); m.obj.assign(num_var, 0);
    for (int c = 0; c < num_pat; c++) m.col_up[c] = 1;
    for (int c = num_pat; c < num_el; c++) m.col_up[c] = max_cn_total;
    for (int c = num_el; c < num_var - 1; c++) m.obj[c] = 1;
    m.col_lo[num_var - 1] = m.col_up[num_var - 1] = bias;
    m.obj[num_var - 1] = -1;
}

bool write_lp(const std::string& path, const IlpModel& m) {
    FILE* f = fopen(path.c_str(), "w");
    if (!f) return false;
    const double INF = DBL_MAX;
    auto term = [&](double v, int c, bool first) {
        if (v == 1) fprintf(f, first ? " x%d" : " + x%d", c);
        else if (v == -1) fprintf(f, " - x%d", c);
        else if (v < 0) fprintf(f, " - %.15g x%d", -v, c);
        else fprintf(f, first ? " %.15g x%d" : " + %.15g x%d", v, c);
    };
    fprintf(f, "\\Problem name: ambigram_bfb\n\nMinimize\nobj:");
    bool first = true; int on_line = 0;
    for (int c = 0; c < m.n_cols; c++) if (m.obj[c] != 0) { term(m.obj[c], c, first); first = false; if (++on_line % 8 == 0) fprintf(f, "\n"); }
    fprintf(f, "\nSubject To\n");
    int64_t name = 0;
    for (int64_t r = 0; r < m.n_rows(); r++) {
        const double lo = m.row_lo[r], up = m.row_up[r];
        auto body = [&]() {
            bool fst = true; int cnt = 0;
            for (int64_t k = m.row_ptr[r]; k < m.row_ptr[r + 1]; k++) { term(m.val[k], m.col[k], fst); fst = false; if (++cnt % 8 == 0) fprintf(f, "\n"); }
        };
        bool nonneg = true;
        for (int64_t k = m.row_ptr[r]; k < m.row_ptr[r + 1]; k++) if (m.val[k] < 0) nonneg = false;
        if (lo == up) { fprintf(f, "R%lld:", (long long)name++); body(); fprintf(f, " = %.15g\n", lo); }
        else if (lo <= -INF) { fprintf(f, "R%lld:", (long long)name++); body(); fprintf(f, " <= %.15g\n", up); }
        else if (up >= INF) { fprintf(f, "R%lld:", (long long)name++); body(); fprintf(f, " >= %.15g\n", lo); }
        else {
            fprintf(f, "R%lld:", (long long)name++); body(); fprintf(f, " <= %.15g\n", up);
            if (!(lo <= 0 && nonneg)) { fprintf(f, "R%lld:", (long long)name++); body(); fprintf(f, " >= %.15g\n", lo); }
        }
    }
    fprintf(f, "Bounds\n");
    for (int c = 0; c < m.n_cols; c++) {
        if (m.col_lo[c] == m.col_up[c]) fprintf(f, " x%d = %.15g\n", c, m.col_lo[c]);
        else if (m.col_up[c] >= INF) { if (m.col_lo[c] != 0) fprintf(f, " x%d >= %.15g\n", c, m.col_lo[c]); }
        else fprintf(f, " %.15g <= x%d <= %.15g\n", m.col_lo[c], c, m.col_up[c]);
    }
    fprintf(f, "Integers\n");
    for (int c = 0; c < m.n_int; c++) fprintf(f, " x%d%s", c, (c % 10 == 9) ? "\n" : "");
    fprintf(f, "\nEnd\n");
    fclose(f);
    return true;
}

bool write_mps(const std::string& path, const IlpModel& m) {
    FILE* f = fopen(path.c_str(), "w");
    if (!f) return false;
    const double INF = DBL_MAX;
    const int64_t nr = m.n_rows();
    // CSR -> CSC by counting
    std::vector<int64_t> cptr((size_t)m.n_cols + 1, 0);
    for (int64_t k = 0; k < m.nnz(); k++) cptr[(size_t)m.col[k] + 1]++;
    for (int c = 0; c < m.n_cols; c++) cptr[(size_t)c + 1] += cptr[c];
    std::vector<int32_t> crow((size_t)m.nnz());
    std::vector<double> cval((size_t)m.nnz());
    {
        std::vector<int64_t> at(cptr.begin(), cptr.end() - 1);
        for (int64_t r = 0; r < nr; r++)
            for (int64_t k = m.row_ptr[r]; k < m.row_ptr[r + 1]; k++) { const int64_t p = at[m.col[k]]++; crow[p] = (int32_t)r; cval[p] = m.val[k]; }
    }
    fprintf(f, "NAME ambigram_bfb\nROWS\n N OBJ\n");
    for (int64_t r = 0; r < nr; r++) {
        const double lo = m.row_lo[r], up = m.row_up[r];
        const char t = (lo == up) ? 'E' : (lo <= -INF ? 'L' : (up >= INF ? 'G' : 'L'));   // two-sided: L row + range
        fprintf(f, " %c R%lld\n", t, (long long)r);
    }
    fprintf(f, "COLUMNS\n");
    bool in_int = false;
    for (int c = 0; c < m.n_cols; c++) {
        if (c < m.n_int && !in_int) { fprintf(f, " MARKER MARKER INTORG\n"); in_int = true; }
        if (c >= m.n_int && in_int) { fprintf(f, " MARKER MARKER INTEND\n"); in_int = false; }
        bool any = false;
        if (m.obj[c] != 0) { fprintf(f, " x%d OBJ %.15g\n", c, m.obj[c]); any = true; }
        for (int64_t k = cptr[c]; k < cptr[(size_t)c + 1]; k++) { fprintf(f, " x%d R%d %.15g\n", c, crow[k], cval[k]); any = true; }
        if (!any) fprintf(f, " x%d OBJ 0\n", c);    // a column must appear to exist
    }
    if (in_int) fprintf(f, " MARKER MARKER INTEND\n");
    fprintf(f, "RHS\n");
    for (int64_t r = 0; r < nr; r++) {
        const double lo = m.row_lo[r], up = m.row_up[r];
        const double rhs = (lo == up) ? lo : (lo <= -INF ? up : (up >= INF ? lo : up));
        if (rhs != 0) fprintf(f, " RHS R%lld %.15g\n", (long long)r, rhs);
    }
    bool ranges = false;
    for (int64_t r = 0; r < nr; r++) {
        const double lo = m.row_lo[r], up = m.row_up[r];
        if (lo != up && lo > -INF && up < INF) {
            if (!ranges) { fprintf(f, "RANGES\n"); ranges = true; }
            fprintf(f, " RNG R%lld %.15g\n", (long long)r, up - lo);
        }
    }
    fprintf(f, "BOUNDS\n");
    for (int c = 0; c < m.n_cols; c++) {
        const double lo = m.col_lo[c], up = m.col_up[c];
        if (lo == up) fprintf(f, " FX BND x%d %.15g\n", c, lo);
        else {
            if (lo <= -INF) fprintf(f, " MI BND x%d\n", c);
            else if (lo != 0) fprintf(f, " LO BND x%d %.15g\n", c, lo);
            if (up < INF) fprintf(f, " UP BND x%d %.15g\n", c, up);
            else if (c < m.n_int) fprintf(f, " PL BND x%d\n", c);   // integer columns default to [0,1] in MPS readers
        }
    }
    fprintf(f, "ENDATA\n");
    fclose(f);
    return true;
}

}  // namespace ambi
