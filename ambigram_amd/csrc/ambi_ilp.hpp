// ambi_ilp.hpp -- host-side ILP model of one chromosome (LocalGenomicMap::BFB_ILP, LGM.cpp:4397-4752) in closed form.
//
// The ILP solve stays on the host (external `cbc`, localhap.cpp:179-181).  The reference ASSEMBLES the row-major model
// with an O(n * numPat^2) double loop (LGM.cpp:4464-4477: 313 s at n = 256); every coefficient is a pure function of
// (row kind, i) or (a, b), so this generator emits the same rows, in the same order with the same in-row entry order,
// in O(nnz).  Columns: [0,numPat) patterns p(a,b) in lexicographic (a,b) order, [numPat,2numPat) loops, then 2n epsilons,
// then the bias column (localhap.cpp:117-133, LGM.cpp:4409-4410).
#pragma once
#include <stdint.h>
#include <string>
#include <utility>
#include <vector>

namespace ambi {

struct IlpModel {
    int n_cols = 0, n_int = 0;
    std::vector<int64_t> row_ptr;      // CSR
    std::vector<int32_t> col;
    std::vector<double> val;
    std::vector<double> row_lo, row_up, col_lo, col_up, obj;   // +-DBL_MAX = infinity (OsiClp getInfinity())
    int64_t n_rows() const { return (int64_t)row_lo.size(); }
    int64_t nnz() const { return (int64_t)col.size(); }
};

// seg_cn: CN of segments start..end after getIndelBias (n values); junc_cn: (end+1) x 2 as returned by getJuncCN with
// ABSOLUTE ids (row i = segment id i); max_cn_total = sum of the CN of ALL segments of the graph (LGM.cpp:4708-4711);
// components: the chromosome's .juncs components (sorted absolute ids), used when juncs_info is set (LGM.cpp:4684-4703).
void build_bfb_ilp(int start_id, int end_id, const double* seg_cn, const double* junc_cn_fold /*[n] fold-back CN of id start..end*/,
                   int bias, double max_cn_total, const std::vector<std::vector<int32_t>>& components, bool juncs_info,
                   IlpModel& m);

// Joint model of G graphs sharing one chromosome (`--op sc_bfb`: LocalGenomicMap::BFB_ILP_SC, LGM.cpp:4754-5093).
// seg_cn / fold_cn: G x n, graph-major (graph 0 after its getIndelBias, localhap.cpp:497; the others as read);
// evolution: the (i, j) pairs of localhap.cpp:417-434 in its iteration order.
void build_bfb_ilp_sc(int start_id, int end_id, int n_graphs, const double* seg_cn, const double* fold_cn,
                      const std::vector<std::pair<int, int>>& evolution, IlpModel& m);

// Row-descriptor form of the same model (ambi_ilp_rows.hpp): fills everything of `m` except the entries (col/val are
// sized and zeroed); entry j of row r is ilp_row_entry(rows[r], ...), written by the device kernel or the host loop.
struct IlpRowDesc;
void build_bfb_ilp_rows(int start_id, int end_id, const double* seg_cn, const double* junc_cn_fold, int bias, double max_cn_total,
                        const std::vector<std::vector<int32_t>>& components, bool juncs_info, IlpModel& m,
                        std::vector<IlpRowDesc>& rows, std::vector<int32_t>& lit_col, std::vector<double>& lit_val);

// CPLEX-LP text readable by `cbc <file>.lp solve solu <file>.sol`; columns are named x<j> as CoinUtils names them
// (the .sol parser relies on it, localhap.cpp:204-205).  Rows whose lower bound 0 is implied (non-negative variables and
// coefficients) are written one-sided.
bool write_lp(const std::string& path, const IlpModel& m);
// The same model as an MPS file (the reference leaves <prefix>.mps beside <prefix>.lp, LGM.cpp:4749): free-format MPS with
// rows R<i>, columns x<j> (the names the .sol parser expects), integer markers around the first n_int columns, RANGES for
// two-sided rows.  Column-major, so the CSR is transposed first (O(nnz)).
bool write_mps(const std::string& path, const IlpModel& m);

}  // namespace ambi
