// ambi_ilp_rows.hpp -- the BFB ILP (LocalGenomicMap::BFB_ILP, LGM.cpp:4397-4752) as ROW DESCRIPTORS + a closed-form
// entry function: entry j of a row is a pure function of (row kind, a, b / i, j), so the 12 bytes per non-zero
// (int32 column + f64 coefficient; 56.5 M non-zeros = 0.68 GB at n = 256) can be written by one thread per entry
// with no dependence between entries.  The host lists the rows (O(rows)); `ambi_ilp_fill_kernel` writes the entries at
// HBM speed; the host form of the same function (tests/hostsim) and the O(nnz) loop generator of ambi_ilp.cpp are
// compared entry for entry in tests/test_ilp_model.py.
//
// Column numbering (localhap.cpp:117-133, LGM.cpp:4409-4410): P(a,b) = rank of (a,b) in the lexicographic list of
// s <= a <= b <= e; L(a,b) = numPat + P(a,b); then 2n epsilons; then the bias column.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#endif
#include "ambi_common.hpp"

namespace ambi {

enum IlpFamily : int32_t {
    ILP_CN = 0,     // segment CN fit: sum p + 2 sum l +- eps          (LGM.cpp:4426-4451)   a = i, rep
    ILP_FB = 1,     // fold-back CN fit: 0.5 p(ends at i) + l +- eps   (:4453-4494)          a = i, rep
    ILP_BIAS = 2,   // bias column fixed                               (:4498-4503)
    ILP_PA = 3,     // parents of pattern (a,b) - p(a,b) >= 0          (:4544-4583)
    ILP_PB = 4,     // p(a,b) + child patterns <= 2
    ILP_LA = 5,     // parents (p and l) of loop (a,b) - l(a,b) >= 0   (:4587-4612)
    ILP_LL = 6,     // child loops + l(a,b) (rep 0) / p(a,b) (rep 1) <= 2   (:4615-4646)
    ILP_PC = 7,     // p(a,b) + child loops/patterns <= 2, two mixes   (:4649-4681)
    ILP_LIT = 8     // literal row (the .juncs components row, :4684-4703): entries at lit[a ...]
};
struct IlpRowDesc { int32_t family, a, b, rep; };

// 24-bit multiply: full rate on the GPU where the 32-bit one is quarter rate; every product here stays far below 2^31 and every
// factor below 2^24 (segment counts are <= 32767)
AMBI_HD int32_t ilp_mul(int32_t x, int32_t y) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __mul24(x, y);
#else
    return x * y;
#endif
}
struct IlpGeom {   // index arithmetic of the (a,b) triangle
    int32_t s, e, n, num_pat, num_el;
    AMBI_HD int32_t P(int a, int b) const { const int da = a - s; return ilp_mul(da, n) - (ilp_mul(da, da - 1) >> 1) + (b - a); }
    AMBI_HD int32_t L(int a, int b) const { return num_pat + P(a, b); }
};
AMBI_HD IlpGeom ilp_geom(int s, int e) {
    IlpGeom G;
    G.s = s; G.e = e; G.n = e - s + 1; G.num_pat = G.n * (G.n + 1) / 2; G.num_el = 2 * G.num_pat;
    return G;
}

AMBI_HD int64_t ilp_row_len(const IlpRowDesc& d, const IlpGeom& G) {
    const int s = G.s, e = G.e, a = d.a, b = d.b;
    switch (d.family) {
        case ILP_CN: return 2ll * (a - s + 1) * (e - a + 1) + 1;
        case ILP_FB: return (int64_t)(a - s) + (a < e ? e - a + 1 : (a > s ? 1 : 0)) + (a - s) + (e - a + 1) + 1;
        case ILP_BIAS: return 1;
        case ILP_PA: return (int64_t)(a - s) + (e - b) + 1;
        case ILP_PB: return 2ll * (b - a) + 1;
        case ILP_LA: return 2ll * (a - s) + 2ll * (e - b) + 1;
        case ILP_LL: return 2ll * (b - a) + 1;
        case ILP_PC: return 2ll * (b - a) + 1;
        default: return d.b;   // ILP_LIT: b = number of literal entries
    }
}

// entry j (0-based) of the row: column and coefficient
AMBI_HD void ilp_row_entry(const IlpRowDesc& d, const IlpGeom& G, int64_t j64, const int32_t* lit_col, const double* lit_val,
                           int32_t* col, double* val) {
    const int s = G.s, e = G.e, a = d.a, b = d.b;
    const int j = (int)j64;
    switch (d.family) {
        case ILP_CN: {
            const int w = e - a + 1, c = (a - s + 1) * w;
            if (j < c) { *col = G.P(s + j / w, a + j % w); *val = 1; }
            else if (j < 2 * c) { const int q = j - c; *col = G.L(s + q / w, a + q % w); *val = 2; }
            else { *col = G.num_el + 2 * (a - s); *val = d.rep == 0 ? 1 : -1; }
            return;
        }
        case ILP_FB: {
            const int n1 = a - s, n2 = a < e ? e - a + 1 : (a > s ? 1 : 0), n3 = a - s, n4 = e - a + 1;
            int q = j;
            if (q < n1) { *col = G.P(s + q, a); *val = 0.5; return; }
            q -= n1;
            if (q < n2) { *col = G.P(a, a + q); *val = 0.5; return; }
            q -= n2;
            if (q < n3) { *col = G.L(s + q, a); *val = 1; return; }
            q -= n3;
            if (q < n4) { *col = G.L(a, a + q); *val = 1; return; }
            *col = G.num_el + 2 * (a - s) + 1; *val = d.rep == 0 ? 1 : -1;
            return;
        }
        case ILP_BIAS: *col = G.num_el + 2 * G.n; *val = 1; return;
        case ILP_PA: {
            const int n1 = a - s, n2 = e - b;
            if (j < n1) { *col = G.P(s + j, b); *val = 1; }
            else if (j < n1 + n2) { *col = G.P(a, b + 1 + (j - n1)); *val = 1; }
            else { *col = G.P(a, b); *val = -1; }
            return;
        }
        case ILP_PB: {
            const int w = b - a;
            if (j < w) { *col = G.P(a, a + j); *val = 1; }
            else if (j < 2 * w) { *col = G.P(a + 1 + (j - w), b); *val = 1; }
            else { *col = G.P(a, b); *val = 1; }
            return;
        }
        case ILP_LA: {
            const int n1 = 2 * (a - s), n2 = 2 * (e - b);
            if (j < n1) { const int q = j >> 1; *col = (j & 1) ? G.L(s + q, b) : G.P(s + q, b); *val = 1; }
            else if (j < n1 + n2) { const int t = j - n1, q = t >> 1; *col = (t & 1) ? G.L(a, b + 1 + q) : G.P(a, b + 1 + q); *val = 1; }
            else { *col = G.L(a, b); *val = -1; }
            return;
        }
        case ILP_LL: {
            const int w = b - a;
            if (j < w) { *col = G.L(a, a + j); *val = 1; }
            else if (j < 2 * w) { *col = G.L(a + 1 + (j - w), b); *val = 1; }
            else { *col = d.rep == 0 ? G.L(a, b) : G.P(a, b); *val = 1; }
            return;
        }
        case ILP_PC: {
            const int w = b - a;
            if (j < w) { *col = d.rep == 0 ? G.L(a, a + j) : G.P(a, a + j); *val = 1; }
            else if (j < 2 * w) { *col = d.rep == 0 ? G.P(a + 1 + (j - w), b) : G.L(a + 1 + (j - w), b); *val = 1; }
            else { *col = G.P(a, b); *val = 1; }
            return;
        }
        default: *col = lit_col[d.a + j]; *val = lit_val[d.a + j]; return;
    }
}


#define AMBI_UNROLL_FIXED _Pragma("unroll")
// `cnt` (1..4) CONSECUTIVE entries j0 .. j0 + cnt - 1 of one row: the family is decided once, the pieces of the row (runs of entries
// whose column follows one formula) are walked with the piece bounds and the triangle arithmetic shared between the entries, and the
// one division of the CN rows (j / w, j % w) is done for the first entry only -- the others step (q, r) forward.  Same values as
// ilp_row_entry, entry for entry (tests/test_ilp_model.py compares both with the host generator and the oracle).
template <int FIXED = 0>
AMBI_HD void ilp_row_entries(const IlpRowDesc& d, const IlpGeom& G, int j0, int cnt_rt, const int32_t* lit_col, const double* lit_val,
                             int32_t* col, double* val) {
    const int s = G.s, e = G.e, a = d.a, b = d.b;
    const int cnt = FIXED > 0 ? FIXED : cnt_rt;   // FIXED: the loops below unroll
    switch (d.family) {
        case ILP_CN: {
            const int w = e - a + 1, c = (a - s + 1) * w;
            int t = j0 < c ? j0 : (j0 < 2 * c ? j0 - c : 0);
            int q = t / w, r = t - q * w;
            AMBI_UNROLL_FIXED for (int k = 0; k < cnt; k++) {
                const int j = j0 + k;
                if (j == c) { q = 0; r = 0; }                      // from the pattern columns to the loop columns
                if (j < 2 * c) {
                    const int pa = G.P(s + q, a + r);
                    col[k] = j < c ? pa : G.num_pat + pa; val[k] = j < c ? 1.0 : 2.0;
                    if (++r == w) { r = 0; q++; }
                } else { col[k] = G.num_el + 2 * (a - s); val[k] = d.rep == 0 ? 1.0 : -1.0; }
            }
            return;
        }
        case ILP_FB: {
            const int n1 = a - s, n2 = a < e ? e - a + 1 : (a > s ? 1 : 0), n3 = a - s, n4 = e - a + 1;
            const int e1 = n1, e2 = e1 + n2, e3 = e2 + n3, e4 = e3 + n4;
            AMBI_UNROLL_FIXED for (int k = 0; k < cnt; k++) {
                const int j = j0 + k;
                if (j < e1) { col[k] = G.P(s + j, a); val[k] = 0.5; }
                else if (j < e2) { col[k] = G.P(a, a + (j - e1)); val[k] = 0.5; }
                else if (j < e3) { col[k] = G.L(s + (j - e2), a); val[k] = 1.0; }
                else if (j < e4) { col[k] = G.L(a, a + (j - e3)); val[k] = 1.0; }
                else { col[k] = G.num_el + 2 * (a - s) + 1; val[k] = d.rep == 0 ? 1.0 : -1.0; }
            }
            return;
        }
        case ILP_BIAS: AMBI_UNROLL_FIXED for (int k = 0; k < cnt; k++) { col[k] = G.num_el + 2 * G.n; val[k] = 1.0; } return;
        case ILP_PA: {
            const int n1 = a - s, n2 = e - b, pab = G.P(a, b);
            AMBI_UNROLL_FIXED for (int k = 0; k < cnt; k++) {
                const int j = j0 + k;
                if (j < n1) { col[k] = G.P(s + j, b); val[k] = 1.0; }
                else if (j < n1 + n2) { col[k] = pab + 1 + (j - n1); val[k] = 1.0; }   // P(a, b + 1 + t) = P(a, b) + 1 + t
                else { col[k] = pab; val[k] = -1.0; }
            }
            return;
        }
        case ILP_LA: {
            const int n1 = 2 * (a - s), n2 = 2 * (e - b), pab = G.P(a, b);
            AMBI_UNROLL_FIXED for (int k = 0; k < cnt; k++) {
                const int j = j0 + k;
                if (j < n1) { const int q = j >> 1; col[k] = ((j & 1) ? G.num_pat : 0) + G.P(s + q, b); val[k] = 1.0; }
                else if (j < n1 + n2) { const int t = j - n1; col[k] = ((t & 1) ? G.num_pat : 0) + pab + 1 + (t >> 1); val[k] = 1.0; }
                else { col[k] = G.num_pat + pab; val[k] = -1.0; }
            }
            return;
        }
        case ILP_PB: case ILP_LL: case ILP_PC: {
            // children of (a,b) that share its start, then those that share its end, then the element itself: the three families differ
            // in which of the two column halves (patterns / loops) each piece lives in
            const int w = b - a, paa = G.P(a, a), pab = G.P(a, b);
            const int off1 = d.family == ILP_PB ? 0 : (d.family == ILP_LL ? G.num_pat : (d.rep == 0 ? G.num_pat : 0));
            const int off2 = d.family == ILP_PB ? 0 : (d.family == ILP_LL ? G.num_pat : (d.rep == 0 ? 0 : G.num_pat));
            const int off3 = d.family == ILP_LL && d.rep == 0 ? G.num_pat : 0;
            AMBI_UNROLL_FIXED for (int k = 0; k < cnt; k++) {
                const int j = j0 + k;
                if (j < w) col[k] = off1 + paa + j;                       // P(a, a + j) = P(a, a) + j
                else if (j < 2 * w) col[k] = off2 + G.P(a + 1 + (j - w), b);
                else col[k] = off3 + pab;
                val[k] = 1.0;
            }
            return;
        }
        default: AMBI_UNROLL_FIXED for (int k = 0; k < cnt; k++) { col[k] = lit_col[d.a + j0 + k]; val[k] = lit_val[d.a + j0 + k]; } return;
    }
}

// Fills col/val of rows [row_lo, row_hi): `lanes` threads of the caller walk every row together (one wavefront per row
// on the GPU: consecutive entries -> consecutive addresses, coalesced 4- and 8-byte stores).
AMBI_HD void ilp_fill_rows(const IlpRowDesc* rows, const int64_t* row_ptr, int64_t row_lo, int64_t row_hi, int64_t row_step,
                           const IlpGeom& G, const int32_t* lit_col, const double* lit_val, int lane, int lanes, int32_t* col, double* val) {
    for (int64_t r = row_lo; r < row_hi; r += row_step) {
        const IlpRowDesc d = rows[r];
        const int64_t p0 = row_ptr[r], len = row_ptr[r + 1] - p0;
        for (int64_t j = lane; j < len; j += lanes) ilp_row_entry(d, G, j, lit_col, lit_val, col + p0 + j, val + p0 + j);
    }
}

// Balanced form: the caller's `lanes` threads fill the entries [p_lo, p_hi) of the non-zero space, whatever rows they
// belong to (one binary search for the first row, then row after row) -- rows differ in length by three orders of
// magnitude (2 n^2 / 3 entries for a segment row, 2 n for a nesting row), so the work is split by entries, not rows.
AMBI_HD void ilp_fill_span(const IlpRowDesc* rows, const int64_t* row_ptr, int64_t n_rows, int64_t p_lo, int64_t p_hi, const IlpGeom& G,
                           const int32_t* lit_col, const double* lit_val, int lane, int lanes, int32_t* col, double* val) {
    if (p_lo >= p_hi) return;
    int64_t lo = 0, hi = n_rows;   // first row whose end lies behind p_lo
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (row_ptr[mid + 1] > p_lo) hi = mid; else lo = mid + 1; }
    for (int64_t r = lo; r < n_rows && row_ptr[r] < p_hi; r++) {
        const IlpRowDesc d = rows[r];
        const int64_t r0 = row_ptr[r], r1 = row_ptr[r + 1];
        const int64_t a0 = r0 > p_lo ? r0 : p_lo, a1 = r1 < p_hi ? r1 : p_hi;
        for (int64_t p = a0 + lane; p < a1; p += lanes) ilp_row_entry(d, G, p - r0, lit_col, lit_val, col + p, val + p);
    }
}

}  // namespace ambi
