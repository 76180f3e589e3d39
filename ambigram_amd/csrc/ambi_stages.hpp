// ambi_stages.hpp -- the per-unit stages of the reconstruction pipeline as SPMD functions over a thread group.
// The HIP kernels (ambi_engine.hip) are thin wrappers that carve the work areas out of LDS and call these;
// the host simulation (tests/hostsim) calls the same functions with HostGroup and plain heap memory.
//
// Pipeline of one unit (reference: localhap.cpp:111-265):
//   stage_prepare   getJuncCN, bias, getIndelBias, no-FBI shortcut, targetCN, constructDAG, ideal lattice + R
//   (plan)          order-table offsets / work blocks over the whole batch
//   stage_enumerate block of consecutive topological orders -> R x K uint8 table
//   stage_first     sequential scan for the first valid order (forward pass, then the flipped orientation)
//   stage_finish_lean  bkp -> path, indelBFB, output junctions, from the runs of the breakpoint path (every unit)
//   stage_finish    the same with the path cells in group memory: units whose SVs chain or edit the path
#pragma once
#include "ambi_batch.hpp"
#include "ambi_enum_blocks.hpp"
#include "ambi_eval.hpp"
#include "ambi_eval_lane.hpp"
#include "ambi_finish.hpp"
#include "ambi_orders.hpp"
#include "ambi_prepare.hpp"
#include "ambi_wide.hpp"

namespace ambi {

// ---------------------------------------------------------------------------------------------
// stage_prepare
// ---------------------------------------------------------------------------------------------
// Group-local memory (LDS on the GPU) of the prepare stage, kept below 10 KB for the bench-sized unit (256 segments,
// 512 junctions) so that 16 units are in flight per CU:
//   * the f64 arrays (junction CN, segment CN) live in the unit's slots of the result blob in HBM, where they end up
//     anyway: every element is written / read by the one thread that owns it;
//   * fold-back and SV lists share one array of 16-bit junction indices (their phases do not overlap);
//   * the normal-junction slot counters, the target-CN difference array and the target CN itself are one array;
//   * the lattice search runs in the bytes of all of the above once their results are in the blob.
// Footprint = persistent part + max(junction phase, lattice).
struct PrepareWork {
    // persistent
    Dag* dag;
    Element* elems;       // [K]
    // junction phase
    JuncEnds* ends;       // [m]    strand-signed junction ends (the 24-byte records stay in HBM)
    int32_t* inv_junc;    // [n+1]
    int32_t* slot_cnt;    // [n+2]  contributions per normal-junction slot; then the target-CN difference array / target CN
    uint16_t* list;       // [m]    fold-back junction list, later the getIndelBias SV list
    uint8_t* taken;       // [m]
    int32_t* idx;         // [64]   node order (constructDAG), later the stack of the sort replay
    Rec3* loops;          // [64]   records permuted by the library sort
    // lattice phase (same bytes as the junction phase)
    uint8_t* lattice_mem; // [kPrepLatticeBytes]
};
// Group-local lattice search: kPrepHashSlots hash slots (at most half as many ideals) and kPrepLinks child links;
// lattices that need more use the pools in HBM.
constexpr int kPrepHashSlots = 256;
constexpr int kPrepLinks = 352;
constexpr int64_t kPrepLatticeBytes = 8ll * kPrepHashSlots /*keys*/ + 4ll * kPrepHashSlots /*pos*/ + 8ll * (kPrepHashSlots / 2) /*ikey*/ +
                                      8ll * (kPrepHashSlots / 2) /*cnt*/ + 4ll * (kPrepHashSlots / 2 + 2) /*cbase*/ + 4ll * kPrepLinks +
                                      4ll * (kMaxNodes + 3) + 16;
AMBI_HD int64_t prepare_persistent_bytes(int K) {
    return pad8(sizeof(Dag)) + pad8(int64_t(sizeof(Element)) * (K > 0 ? K : 1));
}
AMBI_HD int64_t prepare_junction_bytes(int n, int m) {
    return pad8(int64_t(sizeof(JuncEnds)) * m) + pad8(4ll * (n + 1)) + pad8(4ll * (n + 2)) + pad8(2ll * m) + pad8(m) + pad8(4 * 64) +
           pad8(sizeof(Rec3) * 64);
}
AMBI_HD int64_t prepare_work_bytes(int n, int m, int K) {
    const int64_t a = prepare_junction_bytes(n, m), b = pad8(kPrepLatticeBytes);
    return prepare_persistent_bytes(K) + (a > b ? a : b);
}
AMBI_HD PrepareWork carve_prepare(uint8_t* base, int n, int m, int K) {
    PrepareWork W;
    int64_t o = 0;
    W.dag = reinterpret_cast<Dag*>(base + o); o += pad8(sizeof(Dag));
    W.elems = reinterpret_cast<Element*>(base + o); o += pad8(int64_t(sizeof(Element)) * (K > 0 ? K : 1));
    W.lattice_mem = base + o;
    W.ends = reinterpret_cast<JuncEnds*>(base + o); o += pad8(int64_t(sizeof(JuncEnds)) * m);
    W.inv_junc = reinterpret_cast<int32_t*>(base + o); o += pad8(4ll * (n + 1));
    W.slot_cnt = reinterpret_cast<int32_t*>(base + o); o += pad8(4ll * (n + 2));
    W.list = reinterpret_cast<uint16_t*>(base + o); o += pad8(2ll * m);
    W.taken = base + o; o += pad8(m);
    W.idx = reinterpret_cast<int32_t*>(base + o); o += pad8(4 * 64);
    W.loops = reinterpret_cast<Rec3*>(base + o);
    return W;
}
AMBI_HD LatticeWork carve_prepare_lattice(uint8_t* h) {
    LatticeWork L;
    L.keys = reinterpret_cast<uint64_t*>(h); h += 8ll * kPrepHashSlots;
    L.ikey = reinterpret_cast<uint64_t*>(h); h += 8ll * (kPrepHashSlots / 2);
    L.cnt = reinterpret_cast<uint64_t*>(h); h += 8ll * (kPrepHashSlots / 2);
    L.pos = reinterpret_cast<int32_t*>(h); h += 4ll * kPrepHashSlots;
    L.cbase = reinterpret_cast<int32_t*>(h); h += 4ll * (kPrepHashSlots / 2 + 2);
    L.link = reinterpret_cast<uint32_t*>(h); h += 4ll * kPrepLinks;
    L.lvl_off = reinterpret_cast<int32_t*>(h); h += 4ll * (kMaxNodes + 3);
    L.counter = reinterpret_cast<int32_t*>(h);
    L.cap = kPrepHashSlots;
    L.link_cap = kPrepLinks;
    return L;
}

template <class G, class T>
AMBI_HD void copy_words(const G& g, T* dst, const T* src, int64_t count) {
    for (int64_t i = g.tid(); i < count; i += g.size()) dst[i] = src[i];
}

AMBI_HD IdealTable unit_ideal_table(const BatchArgs& A, int u) {
    const UnitIn& U = A.units[u];
    IdealTable T;
    T.lvl_off = A.ideal_lvl_off + int64_t(u) * (kMaxNodes + 3);
    T.counter = A.ideal_counter + 2 * int64_t(u);
    T.a_avail = A.auto_avail + U.ideal_off / 2;
    T.a_cnt = A.auto_cnt + U.ideal_off / 2;
    T.a_cbase = A.auto_cbase + U.ideal_off / 2 + u;
    T.a_child = A.auto_child + 4 * U.ideal_off;
    T.a_nblk = A.auto_nblk + U.ideal_off / 2;
    T.a_depth = A.auto_depth + U.ideal_off / 2;
    T.cap = U.ideal_cap;
    T.child_cap = 4 * U.ideal_cap;
    return T;
}
// lattice-search work areas in the HBM pools (large lattices); the level offsets, counters and child bases are the
// table's own arrays
AMBI_HD LatticeWork unit_lattice_work(const BatchArgs& A, int u, const IdealTable& T) {
    const UnitIn& U = A.units[u];
    LatticeWork W;
    W.keys = A.ideal_keys + U.ideal_off;
    W.pos = A.ideal_pos + U.ideal_off;
    W.ikey = A.ideal_cnt + U.ideal_off;
    W.cnt = A.ideal_cnt + U.ideal_off + U.ideal_cap / 2;
    W.cbase = T.a_cbase;
    W.link = A.ideal_link + 4 * U.ideal_off;
    W.lvl_off = T.lvl_off;
    W.counter = T.counter;
    W.cap = U.ideal_cap;
    W.link_cap = 4 * U.ideal_cap;
    return W;
}

// the unit's final path in run-length form (BatchArgs::run_*): nothing yet / the one run of the reference path 1+ .. P+ / what a
// finish stage found (nr < 0: more runs than slots)
AMBI_HD void runs_none(const BatchArgs& A, int u) { if (A.run_cnt) { A.run_cnt[u] = 0; A.run_cells[u] = 0; } }
AMBI_HD void runs_publish(const BatchArgs& A, int u, int nr, int P) { if (A.run_cnt) { A.run_cnt[u] = nr >= 0 ? nr : -1; A.run_cells[u] = P; } }
AMBI_HD void runs_identity(const BatchArgs& A, int u, int P, int seg_base) {
    if (!A.run_cnt) return;
    if (P > 0 && A.run_slot[u + 1] > A.run_slot[u]) { A.run_start[A.run_slot[u]] = abs_cell(1, seg_base); A.run_len[A.run_slot[u]] = P; A.run_cnt[u] = 1; }
    else A.run_cnt[u] = P > 0 ? -1 : 0;
    A.run_cells[u] = P;
}

// The prepare stage in pieces (stage_prepare runs them one after the other on one group; the express kernel runs the
// junction piece and the DAG piece on two wavefronts at the same time):
//   prep_junctions   junction ends + segment CNs staged, getJuncCN, bias, getIndelBias, the no-fold-back test
//   prep_dag         solution elements staged, targetCN, constructDAG
//   prep_copy_out    target CN, fold-back map, DAG -> result blob / HBM
//   prep_lattice     order-ideal lattice, R, the frozen automaton, the first orders
//   prep_header      the unit's result header
template <class G>
AMBI_HD void prep_junctions(const G& g, const BatchArgs& A, int u, const UnitIn& U, const PrepareWork& W, int* bias, bool* no_fbi, double* inv_sum) {
    const int n = U.n_seg, m = U.n_junc;
    uint8_t* res = A.results + U.res_off;
    const UnitLayout Lay = unit_layout(n, U.bkp_cap, U.path_cap, U.out_cap);
    // staging: junction ends (4 of the 24 bytes of a record) into group memory; the segment CNs go straight into their
    // slot of the result blob (getIndelBias edits them there)
    const JuncView J{W.ends, A.junc_cn + U.junc_off};
    double* junc_cn = reinterpret_cast<double*>(res + Lay.junc_cn);
    double* seg_cn = reinterpret_cast<double*>(res + Lay.seg_cn);
    { const JuncEnds* ge = A.junc_ends + U.junc_off; for (int j = g.tid(); j < m; j += g.size()) W.ends[j] = ge[j]; }
    copy_words(g, reinterpret_cast<uint32_t*>(seg_cn), reinterpret_cast<const uint32_t*>(A.seg_cn + U.seg_off), 2ll * (n + 1));
    g.sync();
    AMBI_MARK(A, g, u, 1);
    get_junc_cn_g(g, n, J, m, junc_cn, W.inv_junc, W.slot_cnt, W.list);                    // localhap.cpp:136-139
    AMBI_MARK(A, g, u, 2);
    *bias = compute_bias_g(g, n, J, junc_cn, W.inv_junc);                                 // :141-146
    AMBI_MARK(A, g, u, 3);
    get_indel_bias_g(g, n, J, m, seg_cn, W.list, W.taken, A.scratch_i32 + A.scratch_off[u]);   // :147
    AMBI_MARK(A, g, u, 4);
    *no_fbi = no_foldback_g(g, n, junc_cn, inv_sum);                                      // :150-153
}
// target_cn: [n+2] ints of group memory that nothing else uses while this runs
template <class G>
AMBI_HD int prep_dag(const G& g, const BatchArgs& A, int u, const UnitIn& U, const PrepareWork& W, int32_t* target_cn) {
    const int K = U.n_elem;
    copy_words(g, reinterpret_cast<uint32_t*>(W.elems), reinterpret_cast<const uint32_t*>(A.elems + U.elem_off), int64_t(sizeof(Element) / 4) * K);
    g.sync();
    if (K <= 0) return ST_ERR_NO_ELEMENTS;
    if (target_cn) target_cn_g(g, W.elems, K, U.n_seg, target_cn, target_cn);            // :222-232 (in place)
    AMBI_MARK(A, g, u, 5);
    DagScratch DS{W.idx, W.loops};
    return construct_dag_g(g, W.elems, K, U.seg_base, *W.dag, DS, A.stage_clk ? A.stage_clk + (int64_t)u * kStageSlots : nullptr);   // :236
}
// the unit's status from the pieces, in the order localhap.cpp decides it (:164 shortcut, :213 infeasible, then the DAG)
AMBI_HD int prep_status(const UnitIn& U, bool no_fbi, int dag_status) {
    if (no_fbi && !U.has_components) return ST_SHORTCUT;
    if (U.infeasible) return ST_INFEASIBLE;
    return dag_status;
}
template <class G>
AMBI_HD void prep_copy_out(const G& g, const BatchArgs& A, int u, const UnitIn& U, const PrepareWork& W, const int32_t* target_cn, bool have_target, int status) {
    const int n = U.n_seg;
    uint8_t* res = A.results + U.res_off;
    const UnitLayout Lay = unit_layout(n, U.bkp_cap, U.path_cap, U.out_cap);
    {
        int32_t* gt = reinterpret_cast<int32_t*>(res + Lay.target_cn);
        for (int i = g.tid(); i <= n; i += g.size()) gt[i] = have_target ? target_cn[i] : 0;
    }
    copy_words(g, reinterpret_cast<uint32_t*>(res + Lay.inv_junc), reinterpret_cast<const uint32_t*>(W.inv_junc), int64_t(n + 1));
    {
        int16_t* isrc = reinterpret_cast<int16_t*>(res + Lay.inv_src);
        int16_t* itgt = reinterpret_cast<int16_t*>(res + Lay.inv_tgt);
        for (int i = g.tid(); i <= n; i += g.size()) {
            const int ji = W.inv_junc[i];
            isrc[i] = (int16_t)(ji >= 0 ? iabs(W.ends[ji].s) : 0);
            itgt[i] = (int16_t)(ji >= 0 ? iabs(W.ends[ji].t) : 0);
        }
    }
    if (status == ST_OK)
        copy_words(g, reinterpret_cast<uint32_t*>(A.dags + u), reinterpret_cast<const uint32_t*>(W.dag), int64_t(sizeof(Dag) / 4));
}
// lattice of the unit's DAG (`pred`: 64 masks in group memory or HBM); lattice_mem: kPrepLatticeBytes of group memory.
// Returns the status (ST_OK, IDEALS / ORDERS capacity); *R_out = number of orders.
template <class G>
AMBI_HD int prep_lattice(const G& g, const BatchArgs& A, int u, const uint64_t* pred, int K, uint8_t* lattice_mem, uint64_t* R_out) {
    const IdealTable T = unit_ideal_table(A, u);
    int64_t* clk = A.stage_clk ? A.stage_clk + (int64_t)u * kStageSlots : nullptr;
    // fast path: search state in group memory (the frozen automaton itself always goes to HBM)
    uint8_t* first_rows = A.first_rows ? A.first_rows + (int64_t)u * A.first_budget * kFirstRowStride : nullptr;
    int st = ideal_build_and_count(g, pred, K, carve_prepare_lattice(lattice_mem), T, R_out, first_rows, A.first_budget, clk, A.block_max);
    if (st == ST_ERR_IDEALS_CAPACITY)   // large lattice
        st = ideal_build_and_count(g, pred, K, unit_lattice_work(A, u, T), T, R_out, first_rows, A.first_budget, nullptr, A.block_max);
    if (st != ST_OK) return st;
    if (*R_out >= kCountSat) return ST_ERR_ORDERS_CAPACITY;   // no table of 2^62 rows: decided here, not by the plan kernel
    return ST_OK;
}
// UnitOut::order_off before the plan stage has given the unit rows: the prepare stage says whether the unit WANTS a table
// (its status is ST_OK when the stage ends), the plan stage answers with an offset or with "no room".  The plan stage reads
// and writes this word only -- not the status, which the scan for the first valid order rewrites -- so that the scan can
// run beside it (the plan kernel is one workgroup: the chip is otherwise idle for its 25 us).
constexpr int64_t kOrderOffWanted = -1, kOrderOffNoRoom = -2, kOrderOffNone = -3;
AMBI_HD void prep_header(UnitOut* out, int status, int bias, int K, uint64_t R, double inv_sum) {
    out->status = status;
    out->bias = bias;
    out->K = K;
    out->bkp_len = 0; out->path_len = 0; out->path_indel_len = 0; out->indel_printed = 0; out->n_out_junc = 0;
    out->first_forward = -1; out->evaluated = 0; out->path_ind_stored = 0; out->reserved = 0;
    out->num_orders = (int64_t)R;
    out->first_valid = -1;
    out->order_off = status == ST_OK ? kOrderOffWanted : kOrderOffNone;
    out->inv_cn_sum = inv_sum;
}

AMBI_HD bool unit_is_wide(const BatchArgs& A, int u) { return A.wide_index != nullptr && A.wide_index[u] >= 0; }

// The prepare stage of a WIDE unit (64..127 nodes, ambi_wide.hpp): the junction side as for every unit, then the DAG, the
// order ideals and the counts in their plain two-word form in the unit's HBM working set.  No first rows are unranked: the
// scan for the first valid order of such a unit is the parallel search over its order table (stage_first hands it over).
template <class G>
AMBI_HD void stage_prepare_wide(const G& g, const BatchArgs& A, int u, uint8_t* work) {
    const UnitIn U = A.units[u];
    const int n = U.n_seg, m = U.n_junc, K = U.n_elem;
    PrepareWork W = carve_prepare(work, n, m, K);
    UnitOut* out = unit_out(A.results, u);
    int bias = 1;
    bool no_fbi = false;
    double inv_sum = 0;
    prep_junctions(g, A, u, U, W, &bias, &no_fbi, &inv_sum);
    int status = prep_status(U, no_fbi, ST_OK);
    bool have_target = false;
    WideUnit& X = A.wide[A.wide_index[u]];
    if (status == ST_OK) {
        const Element* el = A.elems + U.elem_off;
        target_cn_g(g, el, K, n, W.slot_cnt, W.slot_cnt);            // localhap.cpp:222-232
        have_target = true;
        g.sync();
        status = construct_dag_wide(g, el, K, U.seg_base, X);        // :236
    }
    g.sync();
    prep_copy_out(g, A, u, U, W, W.slot_cnt, have_target, ST_SHORTCUT /* (no 64-node DAG record to copy) */);
    g.sync();
    uint64_t R = 0;
    if (status == ST_OK) {
        status = lattice_wide(g, X, &R);
        if (status == ST_OK && (R >= kCountSat || R > (uint64_t)kWideMaxOrders)) status = ST_ERR_ORDERS_CAPACITY;
    }
    if (g.tid() == 0) { prep_header(out, status, bias, K, R, inv_sum); runs_none(A, u); }
    g.sync();
}

template <class G>
AMBI_HD void stage_prepare(const G& g, const BatchArgs& A, int u, uint8_t* work) {
    const UnitIn U = A.units[u];
    const int n = U.n_seg, m = U.n_junc, K = U.n_elem;
    if (K > kMaxNodes) { stage_prepare_wide(g, A, u, work); return; }
    PrepareWork W = carve_prepare(work, n, m, K);
    UnitOut* out = unit_out(A.results, u);
    AMBI_MARK(A, g, u, 0);
    int bias = 1;
    bool no_fbi = false;
    double inv_sum = 0;
    prep_junctions(g, A, u, U, W, &bias, &no_fbi, &inv_sum);
    int status = prep_status(U, no_fbi, ST_OK);
    bool have_target = false;
    if (status == ST_OK) {   // target CN and the DAG only for units that reach the reconstruction (the slot counters are free by now)
        status = prep_dag(g, A, u, U, W, W.slot_cnt);
        have_target = status != ST_ERR_NO_ELEMENTS;
    }
    g.sync();
    AMBI_MARK(A, g, u, 6);
    // results that still sit in group memory: target_cn, fold-back map -- before the lattice search takes the bytes
    prep_copy_out(g, A, u, U, W, W.slot_cnt, have_target, status);
    g.sync();
    AMBI_MARK(A, g, u, 7);
    uint64_t R = 0;
    if (status == ST_OK) status = prep_lattice(g, A, u, W.dag->pred, K, W.lattice_mem, &R);
    if (g.tid() == 0) { prep_header(out, status, bias, K, R, inv_sum); runs_none(A, u); }
    g.sync();
    AMBI_MARK(A, g, u, 8);
}

// ---------------------------------------------------------------------------------------------
// lattice stage alone: units whose front (junctions, DAG, header) was done by the express stage below
// ---------------------------------------------------------------------------------------------
template <class G>
AMBI_HD void stage_lattice(const G& g, const BatchArgs& A, int u, uint8_t* work /* pad8(64 * 8) + kPrepLatticeBytes */) {
    UnitOut* out = unit_out(A.results, u);
    if (out->status != ST_OK) return;
    uint64_t* pred = reinterpret_cast<uint64_t*>(work);
    for (int i = g.tid(); i < 64; i += g.size()) pred[i] = A.dags[u].pred[i];
    g.sync();
    uint64_t R = 0;
    const int st = prep_lattice(g, A, u, pred, out->K, work + 64 * 8, &R);
    if (g.tid() == 0) {
        out->num_orders = (int64_t)R;
        if (st != ST_OK) {   // as the one-piece prepare stage reports it (a path the express stage may have written is void then: the host is told)
            out->status = st; out->order_off = kOrderOffNone;
            if (A.late_flag) *A.late_flag = 1;
        }
    }
    g.sync();
}

// The lattice stage of a small batch, run BESIDE the express stage instead of behind it: it does not wait for the express
// stage's DAG but constructs the DAG itself from the solution elements (the same construct_dag_g: the same DAG), builds the
// lattice / automaton / first rows, and parks R and its status in BatchArgs::lat_R / lat_status.  It does not touch the
// unit's header (the express stage is writing it); plan_merge_lattice does, behind both.  Units the express stage ends
// without a reconstruction (shortcut, infeasible) get a lattice nobody reads.
AMBI_HD int64_t lattice_own_bytes(int K) { return prepare_persistent_bytes(K) + pad8(4 * 64) + pad8(sizeof(Rec3) * 64) + pad8(kPrepLatticeBytes); }
template <class G>
AMBI_HD void stage_lattice_own(const G& g, const BatchArgs& A, int u, uint8_t* work) {
    const UnitIn U = A.units[u];
    const int K = U.n_elem;
    int64_t o = 0;
    Dag* dag = reinterpret_cast<Dag*>(work + o); o += pad8(sizeof(Dag));
    Element* elems = reinterpret_cast<Element*>(work + o); o += pad8(int64_t(sizeof(Element)) * (K > 0 ? K : 1));
    int32_t* idx = reinterpret_cast<int32_t*>(work + o); o += pad8(4 * 64);
    Rec3* loops = reinterpret_cast<Rec3*>(work + o); o += pad8(sizeof(Rec3) * 64);
    uint8_t* lattice_mem = work + o;
    copy_words(g, reinterpret_cast<uint32_t*>(elems), reinterpret_cast<const uint32_t*>(A.elems + U.elem_off), int64_t(sizeof(Element) / 4) * K);
    g.sync();
    int st = ST_ERR_NO_ELEMENTS;
    uint64_t R = 0;
    if (K > 0) {
        DagScratch DS{idx, loops};
        st = construct_dag_g(g, elems, K, U.seg_base, *dag, DS, nullptr);
        g.sync();
        if (st == ST_OK) st = prep_lattice(g, A, u, dag->pred, K, lattice_mem, &R);
    }
    if (g.tid() == 0) { A.lat_R[u] = R; A.lat_status[u] = st; }
    g.sync();
}
// the plan stage's first step when the lattice ran beside the express stage: what stage_lattice would have written
AMBI_HD void plan_merge_lattice(const BatchArgs& A, int u) {
    UnitOut* out = unit_out(A.results, u);
    if (out->status != ST_OK) return;
    out->num_orders = (int64_t)A.lat_R[u];
    const int st = A.lat_status[u];
    if (st != ST_OK) {
        out->status = st; out->order_off = kOrderOffNone;
        if (A.late_flag) *A.late_flag = 1;
    }
}

// ---------------------------------------------------------------------------------------------
// plan: order-table offsets + enumerate work blocks (one thread block for the whole batch)
// ---------------------------------------------------------------------------------------------
// Rows each lane of the enumerate kernel produces: the rows of the whole batch are spread over about
// `target_lanes` lanes; a multiple of 4 (rows leave the registers in 16-byte groups of four) in [4, 1024].
AMBI_HD int rows_per_lane_for(int64_t total_rows, int target_lanes) {
    int64_t t = (total_rows + target_lanes - 1) / (target_lanes > 0 ? target_lanes : 1);
    t = (t + 3) & ~int64_t(3);
    if (t < 4) t = 4;
    if (t > 1024) t = 1024;
    return (int)t;
}
// bytes a unit's table takes of the arena: its rows, rounded up to `align` (a power of two >= 16) so that the next table starts aligned
AMBI_HD int64_t order_bytes(int64_t R, int K, int64_t align = 16) { return (R * row_stride(K) + align - 1) & ~(align - 1); }

// serial reference form of the plan kernel (host simulation)
AMBI_HD void plan_serial(const BatchArgs& A) {
    int64_t total_rows = 0;
    for (int i = 0; i < A.n_units; i++) {
        const UnitOut* out = unit_out(A.results, A.unit_base + i);
        if (out->order_off == kOrderOffWanted && out->num_orders < (int64_t)kCountSat) total_rows += out->num_orders;
    }
    const int T = rows_per_lane_for(total_rows, A.target_lanes);
    int64_t off = 0, blk = 0;   // off: relative to the slice's arena region
    for (int i = 0; i < A.n_units; i++) {
        const int u = A.unit_base + i;
        UnitOut* out = unit_out(A.results, u);
        A.blk_off[i] = blk;
        A.rows_per_lane[u] = T;
        if (out->order_off != kOrderOffWanted) continue;
        const int K = out->K;
        const int64_t R = out->num_orders;
        if (R >= (int64_t)kCountSat) { out->order_off = kOrderOffNoRoom; continue; }
        const int64_t bytes = order_bytes(R, K, A.order_align);
        // plain prefix sum: a unit that does not fit still advances the offset (so orders_needed is the true total)
        if (off + bytes > A.order_arena_bytes) { out->order_off = kOrderOffNoRoom; off += bytes; continue; }
        out->order_off = A.arena_base + off;
        off += bytes;
        blk += (R + 256ll * T - 1) / (256ll * T);   // one work block = 256*T rows = one workgroup (4 waves x 64*T)
    }
    A.blk_off[A.n_units] = blk;
    *A.orders_needed = off;
}

// ---------------------------------------------------------------------------------------------
// stage_enumerate: one work block = 64*T consecutive ranks of one unit = one wavefront; lane l writes rows
// [base + l*T, base + (l+1)*T) straight from registers (ambi_orders.hpp: enumerate_rows).
// Per-lane DFS stacks live in group memory; the automaton is read from a compact LDS copy when it fits
// (LdsAuto) and from HBM/L2 otherwise (GlobalAuto).
// ---------------------------------------------------------------------------------------------
AMBI_HD int64_t enum_stack_bytes(int K) { return pad8(int64_t(64) * K * 3); }

template <int NW, class M, class AUTO>
AMBI_HD void enumerate_lane(const AUTO& au, const AutoView& V, int K, int64_t R, int64_t first_rank, int T,
                            uint8_t* stack_mem, int lane, int lanes, uint8_t* unit_rows) {
    if (first_rank >= R) return;
    int64_t nrows = R - first_rank;
    if (nrows > T) nrows = T;
    LaneStacks S;
    S.stride = lanes;
    S.idx = reinterpret_cast<uint16_t*>(stack_mem) + lane;
    S.prev = stack_mem + (size_t)lanes * K * 2 + lane;
    uint32_t* out = reinterpret_cast<uint32_t*>(unit_rows + first_rank * (int64_t)row_stride(K));
    enumerate_rows<NW, AUTO>(au, V, K, (uint64_t)first_rank, (int)nrows, S, out);
}

// Block-emission form (ambi_enum_blocks.hpp) of one wave's rows [rlo, rhi); dispatch on the row width.
template <int CLS>
AMBI_HD void emit_blocks_dispatch(const uint32_t* img, int nB, int K, uint32_t rlo, uint32_t rhi, uint8_t* unit_rows,
                                  int lane_lo, int lane_hi, int part = 0, int parts = 1) {
    const int nw = row_stride(K) / 4;
    uint32_t* table = reinterpret_cast<uint32_t*>(unit_rows);
#define AMBI_EB(N) emit_blocks_wave<N>(img, nB, rlo, rhi, table, lane_lo, lane_hi, part, parts); return;
    // (5 bits per node up to 32 nodes: class 0, K <= 20, has 1..4 dwords per row, class 1, K <= 32, has 4 or 5)
    if (CLS < 0 || CLS == 0) { switch (nw) { case 1: AMBI_EB(1) case 2: AMBI_EB(2) case 3: AMBI_EB(3) case 4: AMBI_EB(4) default: break; } }
    if (CLS < 0 || CLS == 1) { switch (nw) { case 4: AMBI_EB(4) case 5: AMBI_EB(5) default: break; } }
    if (CLS < 0 || CLS == 2) { switch (nw) { case 7: AMBI_EB(7) case 8: AMBI_EB(8) case 9: AMBI_EB(9) case 10: AMBI_EB(10) case 11: AMBI_EB(11) case 12: AMBI_EB(12) default: break; } }   // (6 bits per node, 33..63 nodes)
#undef AMBI_EB
}

// A unit with no more orders than the prepare stage unranked for the scan (R <= first_budget: every chain-like DAG) has
// its whole table among those rows already: the table is a copy (node bytes, then the 0xFF padding of the row), no image.
template <class G>
AMBI_HD void copy_first_rows(const G& g, const uint8_t* first, int K, int64_t R, uint8_t* rows) {
    const int stride = row_stride(K);
    if (row_packed(K)) {   // one thread per row (at most first_budget = 64 of them): its dwords assembled from the row's bytes
        uint32_t* out = reinterpret_cast<uint32_t*>(rows);
        const int nw = stride / 4;
        for (int64_t r = g.tid(); r < R; r += g.size()) {
            uint32_t* dst = out + r * nw;
            RowBits rb;
            auto flush = [&](int wi, uint32_t w) { dst[wi] = w; };
            for (int d = 0; d < K; d++) rb.put(first[r * kFirstRowStride + d], row_bits(K), flush);
            rb.finish(nw, flush);
        }
        return;
    }
    const int64_t bytes = R * stride;
    for (int64_t x = g.tid(); x < bytes; x += g.size()) {
        const int64_t r = x / stride;
        const int d = (int)(x - r * stride);
        rows[x] = d < K ? first[r * kFirstRowStride + d] : (uint8_t)0xFF;
    }
}

// The directory-free form (emit_blocks_dfs_wave); stack / pw: the calling wave's slots in group memory.
template <int CLS>
AMBI_HD void emit_blocks_dfs_dispatch(const BuildTables& B, const uint32_t* suf, int K, int block_max, uint32_t rlo, uint32_t rhi,
                                      uint8_t* unit_rows, uint16_t* stack, uint32_t* pw, int lane_lo, int lane_hi) {
    const int nw = row_stride(K) / 4;
    uint32_t* table = reinterpret_cast<uint32_t*>(unit_rows);
#define AMBI_ED(N) emit_blocks_dfs_wave<N>(B, suf, K, block_max, rlo, rhi, table, stack, pw, lane_lo, lane_hi); return;
    if (CLS < 0 || CLS == 0) { switch (nw) { case 1: AMBI_ED(1) case 2: AMBI_ED(2) case 3: AMBI_ED(3) case 4: AMBI_ED(4) default: break; } }
    if (CLS < 0 || CLS == 1) { switch (nw) { case 4: AMBI_ED(4) case 5: AMBI_ED(5) default: break; } }
    if (CLS < 0 || CLS == 2) { switch (nw) { case 7: AMBI_ED(7) case 8: AMBI_ED(8) case 9: AMBI_ED(9) case 10: AMBI_ED(10) case 11: AMBI_ED(11) case 12: AMBI_ED(12) default: break; } }
#undef AMBI_ED
}
// group memory of the walk per wave: the ideals of the current path + prefix words with wrap copies
constexpr int kDfsWaveBytes = 2 * 64 + 4 * (16 + 3) + 4;   // 208
constexpr int kDfsWaveStride = (kDfsWaveBytes + 15) & ~15;
constexpr int kDfsStateBytes = 16 * kDfsWaveStride;   // up to sixteen waves per workgroup

// Row-width classes of the enumerate kernel (one kernel instantiation each, so that the register budget of the wide
// rows does not throttle the occupancy of the narrow ones): 0: K <= 20, 1: K <= 32, 2: K <= 63.
AMBI_HD int enum_class_of(int K) { return K <= 20 ? 0 : (K <= 32 ? 1 : (K <= kMaxNodes ? 2 : 3)); }   // 3: wide units (their own table kernel)

// dispatch on K: NW = Kpad/4 dwords per row; 32-bit masks while K <= 32.  CLS < 0: all classes (host simulation).
template <int CLS, class AUTO32, class AUTO64>
AMBI_HD void enumerate_lane_dispatch(const AUTO32& a32, const AUTO64& a64, const AutoView& V, int K, int64_t R,
                                     int64_t first_rank, int T, uint8_t* stack_mem, int lane, int lanes, uint8_t* unit_rows) {
    const int nw = row_byte_words(K);   // (register form: one byte per node)
    if (CLS < 0 || CLS == 0) {
        switch (nw) {
            case 1: enumerate_lane<1, uint32_t>(a32, V, K, R, first_rank, T, stack_mem, lane, lanes, unit_rows); return;
            case 2: enumerate_lane<2, uint32_t>(a32, V, K, R, first_rank, T, stack_mem, lane, lanes, unit_rows); return;
            case 3: enumerate_lane<3, uint32_t>(a32, V, K, R, first_rank, T, stack_mem, lane, lanes, unit_rows); return;
            case 4: enumerate_lane<4, uint32_t>(a32, V, K, R, first_rank, T, stack_mem, lane, lanes, unit_rows); return;
            case 5: enumerate_lane<5, uint32_t>(a32, V, K, R, first_rank, T, stack_mem, lane, lanes, unit_rows); return;
            default: break;
        }
    }
    if (CLS < 0 || CLS == 1) {
        switch (nw) {
            case 6: enumerate_lane<6, uint32_t>(a32, V, K, R, first_rank, T, stack_mem, lane, lanes, unit_rows); return;
            case 7: enumerate_lane<7, uint32_t>(a32, V, K, R, first_rank, T, stack_mem, lane, lanes, unit_rows); return;
            case 8: enumerate_lane<8, uint32_t>(a32, V, K, R, first_rank, T, stack_mem, lane, lanes, unit_rows); return;
            default: break;
        }
    }
    if (CLS < 0 || CLS == 2) {
        if (nw == 12) enumerate_lane<12, uint64_t>(a64, V, K, R, first_rank, T, stack_mem, lane, lanes, unit_rows);
        else if (nw == 16) enumerate_lane<16, uint64_t>(a64, V, K, R, first_rank, T, stack_mem, lane, lanes, unit_rows);
    }
}

// ---------------------------------------------------------------------------------------------
// stage_first: sequential scan for the first valid order (LGM.cpp:3519-3696 control flow)
// ---------------------------------------------------------------------------------------------
struct FirstWork {
    cell_t* bkp;        // [bkp_cap]
    int16_t* inv_src;   // [n+1]
    int16_t* inv_tgt;   // [n+1]
    Dag* dag;           // node records of an ordinary unit ...
    WideDag* wdag;      // ... or of a wide one (the same bytes; nullptr for ordinary units)
    uint8_t* ord;       // [64], [256] for a wide unit
};
// wide: the area is laid out for a unit with 64..255 nodes (larger node records, an order row of up to 256 bytes)
AMBI_HD int64_t first_work_bytes(int n, int bkp_cap, bool wide = false) {
    return pad8(2ll * bkp_cap) + 2 * pad8(2ll * (n + 1)) + pad8(wide ? sizeof(WideDag) : sizeof(Dag)) + (wide ? 256 : 64);
}
AMBI_HD FirstWork carve_first(uint8_t* base, int n, int bkp_cap, bool wide = false) {
    FirstWork W;
    int64_t o = 0;
    W.bkp = reinterpret_cast<cell_t*>(base + o); o += pad8(2ll * bkp_cap);
    W.inv_src = reinterpret_cast<int16_t*>(base + o); o += pad8(2ll * (n + 1));
    W.inv_tgt = reinterpret_cast<int16_t*>(base + o); o += pad8(2ll * (n + 1));
    W.dag = reinterpret_cast<Dag*>(base + o);
    W.wdag = wide ? reinterpret_cast<WideDag*>(base + o) : nullptr;
    o += pad8(wide ? sizeof(WideDag) : sizeof(Dag));
    W.ord = base + o;
    return W;
}
// one order through the evaluation, with the node records the work area holds
template <class G>
AMBI_HD int eval_order_w(const G& g, const FirstWork& W, const uint8_t* ord, bool forward, const InvMap& inv, cell_t* bkp, int cap, int* L_out,
                         int64_t* clk = nullptr) {
    return W.wdag ? eval_order(g, *W.wdag, ord, forward, inv, bkp, cap, L_out, clk) : eval_order(g, *W.dag, ord, forward, inv, bkp, cap, L_out, clk);
}

template <class G>
AMBI_HD void load_first_work(const G& g, const BatchArgs& A, int u, const FirstWork& W) {
    const UnitIn& U = A.units[u];
    const UnitLayout Lay = unit_layout(U.n_seg, U.bkp_cap, U.path_cap, U.out_cap);
    uint8_t* res = A.results + U.res_off;
    if (W.wdag) copy_words(g, reinterpret_cast<uint32_t*>(W.wdag), reinterpret_cast<const uint32_t*>(&A.wide[A.wide_index[u]].dag), int64_t(sizeof(WideDag) / 4));
    else
    copy_words(g, reinterpret_cast<uint32_t*>(W.dag), reinterpret_cast<const uint32_t*>(A.dags + u), int64_t(sizeof(Dag) / 4));
    copy_words(g, W.inv_src, reinterpret_cast<const int16_t*>(res + Lay.inv_src), int64_t(U.n_seg + 1));
    copy_words(g, W.inv_tgt, reinterpret_cast<const int16_t*>(res + Lay.inv_tgt), int64_t(U.n_seg + 1));
    g.sync();
}

// diagnostics hook (BatchArgs::inject_valid): the verdict of order nidx in the given orientation, or `v` itself
AMBI_HD int injected_verdict(const BatchArgs& A, int u, int64_t R, int64_t nidx, bool forwardDir, int v) {
    if (!A.inject_valid) return v;
    const int64_t off = A.inject_off[2 * (int64_t)u], len = A.inject_off[2 * (int64_t)u + 1];
    const int64_t at = (forwardDir ? 0 : R) + nidx;
    if (off < 0 || at >= len) return v;
    const int x = A.inject_valid[off + at];
    return x == 127 ? v : x;
}

template <class G>
AMBI_HD void stage_first(const G& g, const BatchArgs& A, int u, uint8_t* work) {
    UnitOut* out = unit_out(A.results, u);
    if (out->status != ST_OK || out->reserved) return;   // (reserved: reconstructed by the express stage already)
    const UnitIn U = A.units[u];
    if (U.n_elem > kMaxNodes) {   // a wide unit has no pre-unranked first rows: the parallel search over its order table takes it from order 0
        if (g.tid() == 0) { out->status = ST_PENDING; out->evaluated = 0; atomic_add_i32(A.n_pending, 1); }
        g.sync();
        return;
    }
    FirstWork W = carve_first(work, U.n_seg, U.bkp_cap);
    AMBI_MARK(A, g, u, 9);
    load_first_work(g, A, u, W);
    AMBI_MARK(A, g, u, 10);
    const UnitLayout Lay = unit_layout(U.n_seg, U.bkp_cap, U.path_cap, U.out_cap);
    uint8_t* res = A.results + U.res_off;
    const int K = out->K;
    const int64_t R = out->num_orders;
    // the first `first_budget` orders come from the rows the prepare stage unranked (when present): this scan then does
    // not depend on the order table and runs beside the enumerate kernel
    const uint8_t* rows = A.first_rows ? A.first_rows + (int64_t)u * A.first_budget * kFirstRowStride : A.order_arena + out->order_off;
    const int rstride = A.first_rows ? kFirstRowStride : row_stride(K);
    InvMap inv{W.inv_src, W.inv_tgt};
    const bool isReversed = (A.flags & FLAG_REVERSED) != 0;
    bool forwardDir = !isReversed;
    int status = ST_NO_VALID_ORDER;
    int64_t found = -1;
    int found_fwd = -1, L = 0;
    int64_t evaluated = 0;
    for (int pass = 0; pass < 2 && found < 0; pass++) {
        int64_t lim = R < A.first_budget ? R : A.first_budget;
        for (int64_t nidx = 0; nidx < lim; nidx++) {
            for (int d = g.tid(); d < K; d += g.size()) W.ord[d] = A.first_rows ? rows[nidx * rstride + d] : (uint8_t)row_node(rows + nidx * rstride, K, d);
            g.sync();
            int Lo = 0;
            int v = eval_order(g, *W.dag, W.ord, forwardDir, inv, W.bkp, U.bkp_cap, &Lo, A.stage_clk ? A.stage_clk + (int64_t)u * kStageSlots : nullptr);
            v = injected_verdict(A, u, R, nidx, forwardDir, v);
            evaluated++;
            if (v < 0) { status = v; found = -2; break; }
            if (v == 1) { found = nidx; found_fwd = forwardDir ? 1 : 0; L = Lo; status = ST_OK; break; }
        }
        if (found == -2) break;
        if (found < 0) {
            if (lim < R) { status = ST_PENDING; break; }   // budget exhausted: parallel search takes over
            forwardDir = !forwardDir;                       // LGM.cpp:3691-3695 (the last order was invalid)
        }
    }
    AMBI_MARK(A, g, u, 11);
    if (status == ST_OK) {
        cell_t* dst = reinterpret_cast<cell_t*>(res + Lay.bkp);
        for (int i = g.tid(); i < L; i += g.size()) dst[i] = W.bkp[i];
    }
    if (g.tid() == 0) {
        out->status = status;
        if (status == ST_OK) {
            out->first_valid = found; out->first_forward = found_fwd; out->bkp_len = L;
        }
        out->evaluated = (int32_t)evaluated;
        if (status == ST_PENDING) atomic_add_i32(A.n_pending, 1);
    }
    g.sync();
    AMBI_MARK(A, g, u, 12);
}

// Parallel search support: evaluate ONE order of a unit; returns 1/0/negative.  `work` as in stage_first
// (already loaded with load_first_work).
template <class G>
AMBI_HD int eval_indexed(const G& g, const BatchArgs& A, int u, const FirstWork& W, int64_t nidx, bool forwardDir, int* L) {
    UnitOut* out = unit_out(A.results, u);
    const UnitIn& U = A.units[u];
    const int K = out->K;
    const uint8_t* rows = A.order_arena + out->order_off;
    for (int d = g.tid(); d < K; d += g.size()) W.ord[d] = (uint8_t)row_node(rows + nidx * row_stride(K), K, d);
    g.sync();
    InvMap inv{W.inv_src, W.inv_tgt};
    const int v = eval_order_w(g, W, W.ord, forwardDir, inv, W.bkp, U.bkp_cap, L);
    return injected_verdict(A, u, out->num_orders, nidx, forwardDir, v);
}

// ---------------------------------------------------------------------------------------------
// Parallel search for the first valid order (units whose sequential scan ran out of budget): the orders of a unit are
// cut into chunks, one group per chunk, in any order and concurrently.  The reference's scan (LGM.cpp:3519-3696) stops
// at the FIRST order that is valid -- or on which it reads out of bounds -- so the result is the minimum over the
// chunks of {index of a valid order} and of {index of an undefined order}, whichever is smaller; orders behind a known
// hit are skipped.  One pass = one orientation.
// ---------------------------------------------------------------------------------------------
struct SearchSlot {
    int64_t found;      // least index of a valid order seen so far (kSearchNone: none)
    int64_t err_key;    // least {index * 256 + (-status)} of an order that ended in an error status (kSearchNone: none)
};
constexpr int64_t kSearchNone = 0x7fffffffffffffffll;
AMBI_HD void atomic_min_i64(int64_t* p, int64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMin(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v);   // values are non-negative
#else
    if (v < *p) *p = v;
#endif
}
AMBI_HD int64_t load_now_i64(const int64_t* p) { return *reinterpret_cast<const volatile int64_t*>(p); }
// first index that can no longer be the answer of this pass
AMBI_HD int64_t search_limit(int64_t found, int64_t err_key) {
    const int64_t e = err_key == kSearchNone ? kSearchNone : (err_key >> 8);
    return found < e ? found : e;
}

template <class G>
AMBI_HD void stage_search_chunk(const G& g, const BatchArgs& A, int u, const FirstWork& W, int64_t first, int chunk, bool forward, SearchSlot* slot) {
    const int64_t R = unit_out(A.results, u)->num_orders;
    for (int64_t n = first; n < first + chunk && n < R; n++) {
        const int64_t lim = (int64_t)g.bcast_u64((uint64_t)search_limit(load_now_i64(&slot->found), load_now_i64(&slot->err_key)), 0);
        if (n >= lim) break;
        int L = 0;
        const int v = eval_indexed(g, A, u, W, n, forward, &L);
        if (v < 0) { if (g.tid() == 0) atomic_min_i64(&slot->err_key, n * 256 + (int64_t)(-v)); break; }
        if (v == 1) { if (g.tid() == 0) atomic_min_i64(&slot->found, n); break; }
    }
    g.sync();
}

// after a pass of the search: the unit's verdict, or nothing when the pass found nothing and the other orientation is
// still to come.  `work`: first_work_bytes of group memory.
template <class G>
AMBI_HD void stage_resolve(const G& g, const BatchArgs& A, int u, uint8_t* work, const SearchSlot* slot, bool forward, int pass) {
    UnitOut* out = unit_out(A.results, u);
    if (out->status != ST_PENDING) return;
    const UnitIn& U = A.units[u];
    const int64_t R = out->num_orders;
    const int64_t f = slot->found, ek = slot->err_key;
    if (ek != kSearchNone && (ek >> 8) < f) {   // the reference meets the undefined order before any valid one
        if (g.tid() == 0) { out->status = -(int32_t)(ek & 255); out->evaluated = (int32_t)(pass * R + (ek >> 8) + 1); }
        return;
    }
    if (f == kSearchNone) {
        if (pass == 1 && g.tid() == 0) { out->status = ST_NO_VALID_ORDER; out->evaluated = (int32_t)(2 * R); }
        return;
    }
    FirstWork W = carve_first(work, U.n_seg, U.bkp_cap, U.n_elem > kMaxNodes);
    load_first_work(g, A, u, W);
    int L = 0;
    const int v = eval_indexed(g, A, u, W, f, forward, &L);   // materialise the winner's breakpoints
    const UnitLayout Lay = unit_layout(U.n_seg, U.bkp_cap, U.path_cap, U.out_cap);
    cell_t* dst = reinterpret_cast<cell_t*>(A.results + U.res_off + Lay.bkp);
    for (int i = g.tid(); i < L; i += g.size()) dst[i] = W.bkp[i];
    if (g.tid() == 0) {
        out->status = (v == 1) ? ST_OK : ST_ERR_REF_UB;
        out->first_valid = f; out->first_forward = forward ? 1 : 0; out->bkp_len = L;
        out->evaluated = (int32_t)(pass * R + f + 1);
    }
    g.sync();
}

// ---------------------------------------------------------------------------------------------
// --all (LGM.cpp:3672-3695): every order of a unit is evaluated in one orientation (pass 0: the first orientation,
// pass 1: the flipped one, run only when the LAST order of pass 0 is invalid, :3691-3695).  Fused enumerate + evaluate:
// a group takes the 64 consecutive orders [64c, 64c+64) of a unit, its threads UNRANK them from the order-ideal automaton
// into group memory (the order table in HBM is not read), the group assembles them one after the other and leaves ONE
// 64-bit word of the unit's validity bitmap.  `rows`: 64 * kFirstRowStride bytes of group memory.
// ---------------------------------------------------------------------------------------------
AMBI_HD int64_t all_words(int64_t R) { return (R + 63) >> 6; }
AMBI_HD bool all_pass0_last_valid(const BatchArgs& A, int u, int64_t R) {
    return R > 0 && ((A.all_bits[A.all_off[u] + ((R - 1) >> 6)] >> ((R - 1) & 63)) & 1ull) != 0;
}
template <class G>
AMBI_HD void stage_all_chunk(const G& g, const BatchArgs& A, int u, const FirstWork& W, uint8_t* rows, int64_t c, int pass) {
    const UnitOut* out = unit_out(A.results, u);
    const UnitIn& U = A.units[u];
    const int K = out->K;
    const int64_t R = out->num_orders;
    const bool fwd0 = !(A.flags & FLAG_REVERSED), forward = pass == 0 ? fwd0 : !fwd0;
    const int64_t first = c * 64;
    const int cnt = (int)(R - first < 64 ? R - first : 64);
    const bool wide = W.wdag != nullptr;    // a wide unit's orders are read from its (small) order table, one at a time
    if (!wide) {
        const AutoView V = auto_view(unit_ideal_table(A, u));
        for (int i = g.tid(); i < cnt; i += g.size()) (void)order_unrank(V, K, (uint64_t)(first + i), rows + (int64_t)i * kFirstRowStride);
    }
    g.sync();
    const InvMap inv{W.inv_src, W.inv_tgt};
    uint64_t word = 0;
    int undefined = 0;
    for (int i = 0; i < cnt; i++) {
        int L = 0;
        const uint8_t* ord = rows + (int64_t)i * kFirstRowStride;
        if (wide) {
            const uint8_t* trow = A.order_arena + out->order_off + (first + i) * row_stride(K);
            for (int d = g.tid(); d < K; d += g.size()) W.ord[d] = (uint8_t)row_node(trow, K, d);
            g.sync();
            ord = W.ord;
        }
        int v = eval_order_w(g, W, ord, forward, inv, W.bkp, U.bkp_cap, &L);
        v = injected_verdict(A, u, R, first + i, forward, v);
        if (v == 1) word |= 1ull << i;
        else if (v < 0) undefined = 1;
        g.sync();
    }
    if (g.tid() == 0) {
        A.all_bits[A.all_off[u] + (int64_t)pass * all_words(R) + c] = word;
        if (undefined) atomic_add_i32(A.all_flags + u, 1);
    }
}
// The same chunk with ONE THREAD PER ORDER (ambi_eval_lane.hpp): every thread unranks its order and walks it through the
// scalar algorithm; the cells of the 64 orders are interleaved in group memory.  rows_t: [64 positions][64 lanes] bytes,
// cells: [bkp_cap][64 lanes] cells.  Units whose breakpoint path is too long for that take stage_all_chunk.
constexpr int kAllLaneMaxCells = 192;   // breakpoint cells per order the lane form holds (64 lanes x 192 cells x 2 bytes = 24 KB per wavefront)
template <class G>
AMBI_HD void stage_all_chunk_lanes(const G& g, const BatchArgs& A, int u, const FirstWork& W, uint8_t* rows_t, cell_t* cells, int64_t c, int pass,
                                   const AutoView* staged = nullptr /* the unit's automaton in group memory, if the caller staged it */) {
    const UnitOut* out = unit_out(A.results, u);
    const UnitIn& U = A.units[u];
    const int K = out->K;
    const int64_t R = out->num_orders;
    const bool fwd0 = !(A.flags & FLAG_REVERSED), forward = pass == 0 ? fwd0 : !fwd0;
    const AutoView V = staged ? *staged : auto_view(unit_ideal_table(A, u));
    const int64_t first = c * 64;
    const int cnt = (int)(R - first < 64 ? R - first : 64);
    const InvMap inv{W.inv_src, W.inv_tgt};
    uint64_t mine = 0;
    int undefined = 0;
    for (int i = g.tid(); i < 64; i += g.size()) {
        int v = 0;
        if (i < cnt) {
            if (A.all_rows_from_table && out->order_off >= 0) {   // experiment switch: the orders from the table the enumerate kernel wrote
                const uint8_t* row = A.order_arena + out->order_off + (first + i) * row_stride(K);
                for (int d = 0; d < K; d++) rows_t[d * 64 + i] = (uint8_t)row_node(row, K, d);
            } else
                (void)order_unrank(V, K, (uint64_t)(first + i), rows_t + i, 64);
            int L = 0;
            v = eval_order_lane(*W.dag, rows_t + i, 64, forward, inv, LaneCells{cells + i, 64}, U.bkp_cap, &L);
            v = injected_verdict(A, u, R, first + i, forward, v);
        }
        if (v == 1) mine |= 1ull << i;
        else if (v < 0) undefined = 1;
    }
    const uint64_t word = g.size() >= 64 ? g.ballot_u64(mine != 0) : mine;
    const bool any_undefined = g.any(undefined != 0);
    if (g.tid() == 0) {
        A.all_bits[A.all_off[u] + (int64_t)pass * all_words(R) + c] = word;
        if (any_undefined) atomic_add_i32(A.all_flags + u, 1);
    }
}
// header of a unit after both passes: all orders of the executed passes were evaluated; undefined orders refuse the unit
// does this rank evaluate chunk `cl` (index inside the unit) = global chunk `c` ?
AMBI_HD bool all_chunk_is_mine(const BatchArgs& A, int64_t c, int64_t cl, int64_t R) {
    if (A.all_world <= 1) return true;
    return (c % A.all_world) == A.all_rank || cl == all_words(R) - 1;
}
AMBI_HD void all_finalize_unit(const BatchArgs& A, int u) {
    UnitOut* out = unit_out(A.results, u);
    if (A.all_off[u + 1] == A.all_off[u]) return;   // unit without a map (not reconstructed)
    const int64_t R = out->num_orders;
    // counts from the bitmaps (complete once the ranks' maps are merged; a single rank's are complete at once)
    const int64_t nw = all_words(R);
    for (int ps = 0; ps < 2; ps++) {
        int cnt = 0;
        for (int64_t w = 0; w < nw; w++) cnt += popc64(A.all_bits[A.all_off[u] + ps * nw + w]);
        A.all_count[2 * (int64_t)u + ps] = cnt;
    }
    out->evaluated = (int32_t)(all_pass0_last_valid(A, u, R) ? R : 2 * R);
    if (A.all_flags[u]) out->status = ST_ERR_REF_UB;
}

// ---------------------------------------------------------------------------------------------
// stage_finish: bkp -> path (LGM.cpp:3661-3670), indelBFB (:3746-3837), output junctions (localhap.cpp:267-289)
// ---------------------------------------------------------------------------------------------
struct FinishWork {
    cell_t* path;       // [path_cap]
    cell_t* bkp;        // [bkp_cap]
    int32_t* offs;      // [bkp_cap/2 + 2]
    JuncEnds* ends;     // [m]   strand-signed junction ends
    int32_t* sv;        // [m]
    int32_t* cand;      // [finish_cand_cap]  candidate junction steps, their representatives and edge keys
    int32_t* first;     // [2n+1] first occurrence of every vertex in the path
    int32_t* last;      // [2n+1] last occurrence
    uint8_t* taken;     // [m]
    uint8_t* has_ext;   // [m]
};
// (the deque of the general indelBFB path, 2m+4 ints, lives in the unit's HBM scratch: it is touched only when SVs chain)
AMBI_HD int finish_cand_cap(int bkp_cap, int out_cap) { return out_cap + 2 * bkp_cap + 32; }
// the full finish stage holds the path cells in group memory: at most kPathLdsCells of them (longer paths are served by
// the lean stage only; if their SVs chain or edit the path the unit ends with ST_ERR_PATH_CAPACITY)
constexpr int kPathLdsCells = 65536;
AMBI_HD int lds_path_cap(const BatchArgs& A, int path_cap) {
    const int lim = A.finish_path_cells > 0 && A.finish_path_cells < kPathLdsCells ? A.finish_path_cells : kPathLdsCells;
    return path_cap < lim ? path_cap : lim;
}
// path_cap: already limited by lds_path_cap
AMBI_HD int64_t finish_work_bytes(int n, int m, int bkp_cap, int path_cap, int out_cap) {
    return pad8(2ll * path_cap + 16) + pad8(2ll * bkp_cap) + pad8(4ll * (bkp_cap / 2 + 2)) + pad8(int64_t(sizeof(JuncEnds)) * m) +
           pad8(4ll * m) + pad8(4ll * finish_cand_cap(bkp_cap, out_cap)) + 2 * pad8(4ll * (2 * n + 1)) + 2 * pad8(m);
}
AMBI_HD FinishWork carve_finish(uint8_t* base, int n, int m, int bkp_cap, int path_cap, int out_cap) {
    FinishWork W;
    int64_t o = 0;
    W.path = reinterpret_cast<cell_t*>(base + o); o += pad8(2ll * path_cap + 16);   // 16-byte aligned, 16 bytes of slack behind
    W.bkp = reinterpret_cast<cell_t*>(base + o); o += pad8(2ll * bkp_cap);
    W.offs = reinterpret_cast<int32_t*>(base + o); o += pad8(4ll * (bkp_cap / 2 + 2));
    W.ends = reinterpret_cast<JuncEnds*>(base + o); o += pad8(int64_t(sizeof(JuncEnds)) * m);
    W.sv = reinterpret_cast<int32_t*>(base + o); o += pad8(4ll * m);
    W.cand = reinterpret_cast<int32_t*>(base + o); o += pad8(4ll * finish_cand_cap(bkp_cap, out_cap));
    W.first = reinterpret_cast<int32_t*>(base + o); o += pad8(4ll * (2 * n + 1));
    W.last = reinterpret_cast<int32_t*>(base + o); o += pad8(4ll * (2 * n + 1));
    W.taken = base + o; o += pad8(m);
    W.has_ext = base + o;
    return W;
}

// A unit the plan stage had no room for ends with ORDERS_CAPACITY whatever the scan made of it (the scan may have run
// beside the plan stage).  Every unit passes through a finish stage after both: this is where the verdict lands.
template <class G>
AMBI_HD bool plan_refused(const G& g, UnitOut* out) {
    if (out->order_off != kOrderOffNoRoom) return false;
    g.sync();
    if (g.tid() == 0) {   // ... and nothing of what the scan (or the express stage) found stays in the header
        out->status = ST_ERR_ORDERS_CAPACITY;
        out->bkp_len = 0; out->path_len = 0; out->path_indel_len = 0; out->indel_printed = 0; out->n_out_junc = 0;
        out->first_forward = -1; out->evaluated = 0; out->path_ind_stored = 0; out->first_valid = -1;
    }
    g.sync();
    return true;
}

// EXT: the path cells live in `ext_path` (device memory of the caller: 2 * path_cap + 16 bytes, 16-byte aligned) instead of
// group memory -- the work area then holds everything but the cells (13 KB for the bench unit instead of 82 KB), so that
// the workgroup fits where a lean finish workgroup fits.  Same code, same results; the cells are reached through global
// loads / stores that stay in the L2 of the workgroup's XCD (28 KB per unit).
template <bool EXT = false, class G>
AMBI_HD void stage_finish(const G& g, const BatchArgs& A, int u, uint8_t* work, cell_t* ext_path = nullptr) {
    UnitOut* out = unit_out(A.results, u);
    if (plan_refused(g, out)) return;
    const UnitIn U = A.units[u];
    const int n = U.n_seg, m = U.n_junc;
    const UnitLayout Lay = unit_layout(n, U.bkp_cap, U.path_cap, U.out_cap);
    uint8_t* res = A.results + U.res_off;
    rcell_t* gpath = reinterpret_cast<rcell_t*>(res + Lay.path);
    rcell_t* gpath2 = reinterpret_cast<rcell_t*>(res + Lay.path_ind);
    OutJunc* gout = reinterpret_cast<OutJunc*>(res + Lay.out_junc);
    const int base = U.seg_base;
    int status = out->status;
    if (status == ST_REFINISH) {   // handed over by the lean stage: an ordinary reconstructed unit
        status = ST_OK;
        g.sync();
        if (g.tid() == 0) out->status = ST_OK;
    }
    if (status == ST_SHORTCUT || status == ST_INFEASIBLE) {
        // reference path 1+ .. n+ (localhap.cpp:165-169 / :214-219); no indelBFB on this branch
        int P = n <= U.path_cap ? n : U.path_cap;
        for (int i = g.tid(); i < P; i += g.size()) gpath[i] = (rcell_t)(i + 1);
        if (g.tid() == 0) {
            out->path_len = P; out->path_indel_len = P; out->indel_printed = 0; out->n_out_junc = 0; out->path_ind_stored = 0;
            if (n > U.path_cap) out->status = ST_ERR_PATH_CAPACITY;
            runs_identity(A, u, P, base);
        }
        g.sync();
        return;
    }
    if (status != ST_OK) return;
    const int pcap = EXT ? U.path_cap : lds_path_cap(A, U.path_cap);
    FinishWork W = carve_finish(work, n, m, U.bkp_cap, EXT ? 0 : pcap, U.out_cap);
    if (EXT) W.path = ext_path;
    const int L = out->bkp_len;
    AMBI_MARK(A, g, u, 16);
    copy_words(g, W.bkp, reinterpret_cast<const cell_t*>(res + Lay.bkp), int64_t(L));
    {
        const JuncEnds* ge = A.junc_ends + U.junc_off;   // 4 bytes per junction instead of the 24-byte record
        for (int j = g.tid(); j < m; j += g.size()) W.ends[j] = ge[j];
    }
    g.sync();
    AMBI_MARK(A, g, u, 17);
    // a launch with a reduced path area (finish_retry): what does not fit it goes to the list kernel behind this one
    const bool may_retry = A.finish_retry && A.refin_list && pcap < U.path_cap;
    auto hand_over = [&]() {
        g.sync();
        if (g.tid() == 0) { out->status = ST_REFINISH; A.refin_list[atomic_add_i32(A.refin_count, 1)] = u; }
        g.sync();
    };
    int P = expand_bkp(g, W.bkp, L, W.path, pcap, W.offs, gpath, base);
    if (P == ST_ERR_PATH_CAPACITY && may_retry) { hand_over(); return; }
    if (P < 0) { if (g.tid() == 0) out->status = P; g.sync(); return; }
    int P2 = P;
    AMBI_MARK(A, g, u, 18);
    IndelScratch S{W.sv, W.taken, W.has_ext, A.scratch_i32 + A.scratch_off[u], W.first, W.last};
    bool edited = false;
    int printed = indel_bfb(g, n, W.ends, m, W.path, &P2, pcap, S, &edited);
    if (printed == ST_ERR_PATH_CAPACITY && may_retry) { hand_over(); return; }
    if (printed < 0) { if (g.tid() == 0) { out->status = printed; out->path_len = P; } g.sync(); return; }
    AMBI_MARK(A, g, u, 19);
    // the edited path is materialised only when indelBFB changed something; otherwise readers take `path`
    if (edited) for (int i = g.tid(); i < P2; i += g.size()) gpath2[i] = W.path[i];
    // output junctions of the final path; records go straight into the blob (absolute ids)
    int nout = synth_out_juncs(g, W.path, P2, gout, U.out_cap, W.cand, finish_cand_cap(U.bkp_cap, U.out_cap), base);
    AMBI_MARK(A, g, u, 20);
    int nruns = 0;
    if (A.run_cnt) nruns = emit_runs_cells(g, W.path, P2, base, A.run_start + A.run_slot[u], A.run_len + A.run_slot[u], (int)(A.run_slot[u + 1] - A.run_slot[u]));
    if (g.tid() == 0) {
        out->path_len = P; out->path_indel_len = P2; out->indel_printed = printed; out->path_ind_stored = edited ? 1 : 0;
        out->n_out_junc = nout >= 0 ? nout : 0;
        if (nout < 0) out->status = nout;
        runs_publish(A, u, nruns, P2);
    }
    g.sync();
    AMBI_MARK(A, g, u, 21);
}

// ---------------------------------------------------------------------------------------------
// stage_finish_lean: the same results from the runs of the breakpoint path alone (ambi_finish.hpp, "Lean finish").
// ~11 KB of group memory for a 256-segment unit instead of ~38 KB, so its workgroups fit beside the resident
// enumerate workgroups instead of taking their places.  Units whose SVs chain or edit the path get ST_REFINISH, are
// counted in n_pending and go through stage_finish afterwards.
// ---------------------------------------------------------------------------------------------
struct FinishLeanWork {
    cell_t* bkp;        // [bkp_cap]
    int32_t* offs;      // [bkp_cap/2 + 2]
    JuncEnds* ends;     // [m]
    int32_t* sv;        // [m]
    int32_t* cand;      // [3 * (bkp_cap/2 + 1)]
    int32_t* first;     // [2n+1]
    int32_t* last;      // [2n+1]
    uint8_t* taken;     // [m]
    uint8_t* has_ext;   // [m]
    int32_t* htab;      // [lean_hash_ints(bkp_cap)]  hash table of the output-junction synthesis (synth_classes)
};
AMBI_HD int lean_hash_ints(int bkp_cap) { int h = 64; while (h < 2 * (bkp_cap / 2 + 1)) h <<= 1; return 2 * h; }
AMBI_HD int64_t finish_lean_work_bytes(int n, int m, int bkp_cap) {
    return pad8(2ll * bkp_cap) + pad8(4ll * (bkp_cap / 2 + 2)) + pad8(int64_t(sizeof(JuncEnds)) * m) + pad8(4ll * m) +
           pad8(12ll * (bkp_cap / 2 + 1)) + 2 * pad8(4ll * (2 * n + 1)) + 2 * pad8(m) + pad8(4ll * lean_hash_ints(bkp_cap));
}
AMBI_HD FinishLeanWork carve_finish_lean(uint8_t* base, int n, int m, int bkp_cap) {
    FinishLeanWork W;
    int64_t o = 0;
    W.bkp = reinterpret_cast<cell_t*>(base + o); o += pad8(2ll * bkp_cap);
    W.offs = reinterpret_cast<int32_t*>(base + o); o += pad8(4ll * (bkp_cap / 2 + 2));
    W.ends = reinterpret_cast<JuncEnds*>(base + o); o += pad8(int64_t(sizeof(JuncEnds)) * m);
    W.sv = reinterpret_cast<int32_t*>(base + o); o += pad8(4ll * m);
    W.cand = reinterpret_cast<int32_t*>(base + o); o += pad8(12ll * (bkp_cap / 2 + 1));
    W.first = reinterpret_cast<int32_t*>(base + o); o += pad8(4ll * (2 * n + 1));
    W.last = reinterpret_cast<int32_t*>(base + o); o += pad8(4ll * (2 * n + 1));
    W.taken = base + o; o += pad8(m);
    W.has_ext = base + o; o += pad8(m);
    W.htab = reinterpret_cast<int32_t*>(base + o);
    return W;
}

// mirror: a second place for the cells of the final path (the express stage's slot of the pinned result mailbox).
// pre_nsv >= 0: the junction ends are in the work area and indel_collect has run on them already (the express stage does that
// on its junction-side wavefront while the DAG side is still busy): its result.
#if defined(AMBI_EDIT_MARKS)   // diagnostic build: marks inside the one-wavefront tail, in the slots of the prepare stage's marks
#define LEAN_DIAG_MARK(w, x) AMBI_MARK(A, w, u, x)
#else
#define LEAN_DIAG_MARK(w, x) ((void)0)
#endif
template <class G>
AMBI_HD void stage_finish_lean(const G& g, const BatchArgs& A, int u, uint8_t* work, rcell_t* mirror = nullptr, int pre_nsv = -1) {
    UnitOut* out = unit_out(A.results, u);
    const UnitIn U = A.units[u];
    const int n = U.n_seg, m = U.n_junc;
    const UnitLayout Lay = unit_layout(n, U.bkp_cap, U.path_cap, U.out_cap);
    uint8_t* res = A.results + U.res_off;
    rcell_t* gpath = reinterpret_cast<rcell_t*>(res + Lay.path);
    OutJunc* gout = reinterpret_cast<OutJunc*>(res + Lay.out_junc);
    const int base = U.seg_base;
    if (plan_refused(g, out)) return;
    const int status = out->status;
    if (out->reserved) return;                        // reconstructed by the express stage already
    if (U.direct_full && A.direct_full_on) return;   // taken by the full stage from the start (its kernel runs beside this one)
    if (status == ST_SHORTCUT || status == ST_INFEASIBLE) {
        // reference path 1+ .. n+ (localhap.cpp:165-169 / :214-219); no indelBFB on this branch
        int P = n <= U.path_cap ? n : U.path_cap;
        for (int i = g.tid(); i < P; i += g.size()) gpath[i] = (rcell_t)(i + 1);
        if (g.tid() == 0) {
            out->path_len = P; out->path_indel_len = P; out->indel_printed = 0; out->n_out_junc = 0; out->path_ind_stored = 0;
            if (n > U.path_cap) out->status = ST_ERR_PATH_CAPACITY;
            runs_identity(A, u, P, base);
        }
        g.sync();
        return;
    }
    if (status != ST_OK) return;
    FinishLeanWork W = carve_finish_lean(work, n, m, U.bkp_cap);
    const int L = out->bkp_len, np = L / 2;
    AMBI_MARK(A, g, u, 16);
    copy_words(g, W.bkp, reinterpret_cast<const cell_t*>(res + Lay.bkp), int64_t(L));
    if (pre_nsv < 0) {
        const JuncEnds* ge = A.junc_ends + U.junc_off;   // 4 bytes per junction instead of the 24-byte record
        for (int j = g.tid(); j < m; j += g.size()) W.ends[j] = ge[j];
    }
    g.sync();
    AMBI_MARK(A, g, u, 17);
    const int P = run_offsets(g, W.bkp, L, W.offs);
    LEAN_DIAG_MARK(g, 1);
    if (P > U.path_cap) { if (g.tid() == 0) out->status = ST_ERR_PATH_CAPACITY; g.sync(); return; }
    IndelScratch S{W.sv, W.taken, W.has_ext, nullptr, W.first, W.last};
    const int nsv = pre_nsv >= 0 ? pre_nsv : indel_collect(g, n, W.ends, m, S);
    LEAN_DIAG_MARK(g, 2);
    if (nsv > 0) {
        for (int i = g.tid(); i < 2 * n + 1; i += g.size()) { W.first[i] = 0x7fffffff; W.last[i] = -1; }
        g.sync();
    }
    LEAN_DIAG_MARK(g, 3);
    expand_runs(g, W.bkp, np, W.offs, gpath, base, n, nsv > 0 ? W.first : nullptr, W.last, mirror);
    AMBI_MARK(A, g, u, 18);
    auto refinish = [&]() {   // chaining or editing SVs: the full stage redoes this unit
        if (g.tid() == 0) { out->status = ST_REFINISH; if (A.refin_list) A.refin_list[atomic_add_i32(A.refin_count, 1)] = u; }   // (no list: the caller looks at the status itself)
        g.sync();
    };
    int printed = 0, nout, nruns = 0;
    const RunPath RP{W.bkp, W.offs, np};
#if defined(__HIP_DEVICE_COMPILE__)
    if (G::kIsBlock && np <= 256 && g.size() >= 128) {
        // few breakpoint pairs: the output-junction synthesis is a chain of short dependent phases -- ONE wavefront goes through
        // it without a workgroup barrier, while the other wavefronts do the SV look-ups, which are independent of it (if the
        // look-ups hand the unit to the full stage, that stage writes the output junctions again)
        int v = 0, stop = 0;
#if defined(AMBI_LEAN_SKIP) && (AMBI_LEAN_SKIP & 8)
        if (g.tid() < 64) {}
#else
        if (g.tid() < 64) {
            WaveGroup w;
            v = synth_out_juncs_runs(w, W.bkp, np, W.offs, gout, U.out_cap, W.cand, base, W.htab, lean_hash_ints(U.bkp_cap));
            LEAN_DIAG_MARK(w, 7);
        }
#endif
#if !defined(AMBI_LEAN_SKIP) || !(AMBI_LEAN_SKIP & 4)
        else {
            // the second wavefront: the run-length form of the path first (reads the pairs only, as the synthesis does)
            if (g.tid() < 128 && A.run_cnt) {
                WaveGroup w;
                nruns = emit_runs_pairs(w, W.bkp, np, W.offs, P, base, A.run_start + A.run_slot[u], A.run_len + A.run_slot[u], (int)(A.run_slot[u + 1] - A.run_slot[u]));
#if defined(AMBI_EDIT_MARKS)
                if (g.tid() == 64 && A.stage_clk) A.stage_clk[(int64_t)u * kStageSlots + 8] = stage_clock();
#endif
            }
            if (nsv > 0) stop = indel_lookups_thread(g.tid() - 64, g.size() - 64, n, W.ends, nsv, RP, P, S);
#if defined(AMBI_EDIT_MARKS)
            if (g.tid() == 64 && A.stage_clk) A.stage_clk[(int64_t)u * kStageSlots + 6] = stage_clock();
#endif
        }
#endif
        nout = g.bcast_i32(v, 0);
        nruns = g.bcast_i32(nruns, 64);
        if (nsv > 0) { printed = g.any(stop != 0) ? 0 : 1; if (!printed) { refinish(); return; } }
        AMBI_MARK(A, g, u, 19);
    } else
#endif
    {
        if (nsv > 0) {
            printed = indel_lookups_only(g, n, W.ends, nsv, RP, P, S);
            if (!printed) { refinish(); return; }
        }
        AMBI_MARK(A, g, u, 19);
        nout = synth_out_juncs_runs(g, W.bkp, np, W.offs, gout, U.out_cap, W.cand, base, W.htab, lean_hash_ints(U.bkp_cap));
        if (A.run_cnt) nruns = emit_runs_pairs(g, W.bkp, np, W.offs, P, base, A.run_start + A.run_slot[u], A.run_len + A.run_slot[u], (int)(A.run_slot[u + 1] - A.run_slot[u]));
    }
    AMBI_MARK(A, g, u, 20);
    if (g.tid() == 0) {
        out->path_len = P; out->path_indel_len = P; out->indel_printed = printed; out->path_ind_stored = 0;
        out->n_out_junc = nout >= 0 ? nout : 0;
        if (nout < 0) out->status = nout;
        runs_publish(A, u, nruns, P);
    }
    g.sync();
    AMBI_MARK(A, g, u, 21);
}

// ---------------------------------------------------------------------------------------------
// stage_finish_edit: units whose SVs may EDIT the path (UnitIn::direct_full) without the cells of the path in any memory of
// the workgroup -- the lean stage's runs, indelBFB on the run list (ambi_finish.hpp: indel_bfb_runs), the edited path written
// once from the final list.  Same results as stage_finish.  What this stage does not do -- run lists or paths that outgrow
// their room -- is handed on with ST_REFINISH: the caller runs stage_finish on the unit (`hand_list` / `hand_count`: the device list of the
// direct full-stage launch behind this one; nullptr: the caller looks at the status).
// Measured (profiles/r04_notes.md): 195 k shader cycles per bench unit with two deletions and a duplication in stage_finish with
// the cells in device memory (125 k of them inside indelBFB: three shifts of 14 000 cells and three table fills).
// ---------------------------------------------------------------------------------------------
struct FinishEditWork {
    FinishLeanWork L;       // (cand is not used: the lists below can be longer than the breakpoint path)
    cell_t* val[2];         // [2 * cap2] two run lists: an edit reads one and writes the other
    int32_t* off[2];        // [cap2 + 2]
    int32_t* cand;          // [3 * (cap2 + 1)]
    int32_t* misc;          // [16]  loc[8] of runs_build
    int32_t* htab; int htab_ints;   // hash table of the output-junction synthesis: the occurrence tables, which indelBFB is done with by then (the run lists can be longer than the lean stage's own table allows for)
    int32_t* grp;           // [2m + 4]  the deque of chaining SVs (stage_finish keeps it in device memory: there it is touched rarely
                            //           compared with the cells; here every access would be the longest wait of its step)
};
AMBI_HD int edit_run_cap(int bkp_cap) { return 2 * (bkp_cap / 2) + 32; }
AMBI_HD int64_t finish_edit_work_bytes(int n, int m, int bkp_cap) {
    const int64_t c = edit_run_cap(bkp_cap);
    return pad8(finish_lean_work_bytes(n, m, bkp_cap)) + 2 * pad8(4 * c) + 2 * pad8(4 * (c + 2)) + pad8(12 * (c + 1)) + 64 + pad8(4ll * (2 * m + 4));
}
AMBI_HD FinishEditWork carve_finish_edit(uint8_t* base, int n, int m, int bkp_cap) {
    FinishEditWork W;
    W.L = carve_finish_lean(base, n, m, bkp_cap);
    const int64_t c = edit_run_cap(bkp_cap);
    int64_t o = pad8(finish_lean_work_bytes(n, m, bkp_cap));
    for (int k = 0; k < 2; k++) { W.val[k] = reinterpret_cast<cell_t*>(base + o); o += pad8(4 * c); }
    for (int k = 0; k < 2; k++) { W.off[k] = reinterpret_cast<int32_t*>(base + o); o += pad8(4 * (c + 2)); }
    W.cand = reinterpret_cast<int32_t*>(base + o); o += pad8(12 * (c + 1));
    W.misc = reinterpret_cast<int32_t*>(base + o); o += 64;
    W.grp = reinterpret_cast<int32_t*>(base + o);
    W.htab = W.L.first; W.htab_ints = (int)(2 * pad8(4ll * (2 * n + 1)) / 4);   // (first and last lie one behind the other, carve_finish_lean)
    return W;
}

template <class G>
AMBI_HD void stage_finish_edit(const G& g, const BatchArgs& A, int u, uint8_t* work, int32_t* hand_list = nullptr, int32_t* hand_count = nullptr) {
    UnitOut* out = unit_out(A.results, u);
    const UnitIn U = A.units[u];
    const int n = U.n_seg, m = U.n_junc;
    const UnitLayout Lay = unit_layout(n, U.bkp_cap, U.path_cap, U.out_cap);
    uint8_t* res = A.results + U.res_off;
    rcell_t* gpath = reinterpret_cast<rcell_t*>(res + Lay.path);
    rcell_t* gpath2 = reinterpret_cast<rcell_t*>(res + Lay.path_ind);
    OutJunc* gout = reinterpret_cast<OutJunc*>(res + Lay.out_junc);
    const int base = U.seg_base;
    if (plan_refused(g, out)) return;
    int status = out->status;
    if (out->reserved) return;                        // reconstructed by the express stage already
    if (status == ST_REFINISH) {   // handed over by the lean stage (host simulation): an ordinary reconstructed unit
        status = ST_OK;
        g.sync();
        if (g.tid() == 0) out->status = ST_OK;
    }
    auto hand_on = [&]() {
        g.sync();
        if (g.tid() == 0) { out->status = ST_REFINISH; if (hand_list) hand_list[atomic_add_i32(hand_count, 1)] = u; }
        g.sync();
    };
    if (status != ST_OK) {   // shortcut / infeasible units and errors: the full stage's own branches (rare among these units)
        if (status == ST_SHORTCUT || status == ST_INFEASIBLE) {
            int P = n <= U.path_cap ? n : U.path_cap;
            for (int i = g.tid(); i < P; i += g.size()) gpath[i] = (rcell_t)(i + 1);
            if (g.tid() == 0) {
                out->path_len = P; out->path_indel_len = P; out->indel_printed = 0; out->n_out_junc = 0; out->path_ind_stored = 0;
                if (n > U.path_cap) out->status = ST_ERR_PATH_CAPACITY;
                runs_identity(A, u, P, base);
            }
            g.sync();
        }
        return;
    }
    FinishEditWork W = carve_finish_edit(work, n, m, U.bkp_cap);
    const int cap2 = (A.edit_cap_limit > 0 && A.edit_cap_limit < edit_run_cap(U.bkp_cap)) ? A.edit_cap_limit : edit_run_cap(U.bkp_cap);
    const int L = out->bkp_len, np = L / 2;
    AMBI_MARK(A, g, u, 16);
    copy_words(g, W.L.bkp, reinterpret_cast<const cell_t*>(res + Lay.bkp), int64_t(L));
    {
        const JuncEnds* ge = A.junc_ends + U.junc_off;
        for (int j = g.tid(); j < m; j += g.size()) W.L.ends[j] = ge[j];
    }
    g.sync();
    AMBI_MARK(A, g, u, 17);
    const int P = run_offsets(g, W.L.bkp, L, W.L.offs);
    if (P > U.path_cap) { hand_on(); return; }        // (the full stage reports it)
    expand_runs(g, W.L.bkp, np, W.L.offs, gpath, base, n, (int32_t*)nullptr, (int32_t*)nullptr);
    AMBI_MARK(A, g, u, 18);
    IndelScratch S{W.L.sv, W.L.taken, W.L.has_ext, W.grp, W.L.first, W.L.last};
    RunList cur{W.L.bkp, W.L.offs, np, P};
    RunList buf[2] = {{W.val[0], W.off[0], 0, 0}, {W.val[1], W.off[1], 0, 0}};
    bool edited = false;
    const int printed = indel_bfb_runs(g, n, W.L.ends, m, cur, buf, cap2, U.path_cap, S, W.misc, &edited, A.stage_clk ? A.stage_clk + (int64_t)u * kStageSlots : nullptr);
    if (printed < 0) { hand_on(); return; }           // no room for the runs, or an error the full stage reports
    AMBI_MARK(A, g, u, 19);
    int nout = 0, nruns = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    if (G::kIsBlock && cur.n <= 256 && g.size() >= 128) {
        // as in the lean stage: ONE wavefront goes through the chain of short phases of the output-junction synthesis and the run
        // emission without a workgroup barrier, the other wavefronts write the cells of the edited path meanwhile
        if (g.tid() < 64) {
            WaveGroup w;
            nout = synth_out_juncs_runs(w, cur.val, cur.n, cur.off, gout, U.out_cap, W.cand, base, W.htab, W.htab_ints);
        } else {
            if (g.tid() < 128 && A.run_cnt) {
                WaveGroup w;
                nruns = emit_runs_pairs(w, cur.val, cur.n, cur.off, cur.P, base, A.run_start + A.run_slot[u], A.run_len + A.run_slot[u], (int)(A.run_slot[u + 1] - A.run_slot[u]));
            }
            if (edited) {
                const int lane = g.tid() & 63, sub = (g.tid() >> 6) - 1, nsub = (g.size() >> 6) - 1;
                for (int j = sub; j < cur.n; j += nsub) {
                    const int a = cur.val[2 * j], o0 = cur.off[j], len = cur.off[j + 1] - o0;
                    for (int k = lane; k < len; k += 64) gpath2[o0 + k] = (rcell_t)(a + k);
                }
            }
        }
        nout = g.bcast_i32(nout, 0);
        nruns = g.bcast_i32(nruns, 64);
        AMBI_MARK(A, g, u, 20);
    } else
#endif
    {
        if (edited) expand_runs(g, cur.val, cur.n, cur.off, gpath2, base, n, (int32_t*)nullptr, (int32_t*)nullptr);
        nout = synth_out_juncs_runs(g, cur.val, cur.n, cur.off, gout, U.out_cap, W.cand, base, W.htab, W.htab_ints);
        AMBI_MARK(A, g, u, 20);
        if (A.run_cnt) nruns = emit_runs_pairs(g, cur.val, cur.n, cur.off, cur.P, base, A.run_start + A.run_slot[u], A.run_len + A.run_slot[u], (int)(A.run_slot[u + 1] - A.run_slot[u]));
    }
    if (g.tid() == 0) {
        out->path_len = P; out->path_indel_len = cur.P; out->indel_printed = printed; out->path_ind_stored = edited ? 1 : 0;
        out->n_out_junc = nout >= 0 ? nout : 0;
        if (nout < 0) out->status = nout;
        runs_publish(A, u, nruns, cur.P);
    }
    g.sync();
    AMBI_MARK(A, g, u, 21);
}

// ---------------------------------------------------------------------------------------------
// stage_express: the whole reconstruction of a unit whose FIRST order is valid, in one workgroup -- for small batches,
// where the latency of the kernel chain (prepare -> plan -> scan -> finish, each a single wavefront working through
// order-dependent steps) is what a caller waits for.  allTopologicalOrders emits the orders in lexicographic order
// (LGM.cpp:3380-3409), so order 0 is "always the lowest-numbered node whose predecessors are placed" -- K steps over the
// DAG, no lattice.  Two wavefronts work at the same time: one on the junction side (getJuncCN, bias, getIndelBias), one
// on the DAG side (targetCN, constructDAG, order 0, placement of its elements); then imperfectFBI (needs both), then the
// finish stage on the whole workgroup.  If order 0 does not assemble, the unit is left to the ordinary scan kernel; the
// lattice / order table are built behind this kernel either way (stage_lattice).
// W0 / W1: the junction-side and DAG-side wavefront groups (the same 1-thread group in the host simulation, which runs
// the two sides one after the other); gb: the workgroup; role: 0 / 1 = this thread belongs to W0 / W1, 2 = neither,
// -1 = every role in turn (host).
// ---------------------------------------------------------------------------------------------
struct ExpressWork {
    PrepareWork P;
    int32_t* target;      // [n+2]   target CN (the DAG side's own array: the slot counters belong to the junction side)
    FirstWork F;
    int32_t* flags;       // [8]     cross-wave scalars
    uint8_t* finish;      // finish-stage work area
};
AMBI_HD int64_t express_finish_bytes(int n, int m, int bkp_cap, int path_cells, int out_cap) {
    const int64_t a = finish_work_bytes(n, m, bkp_cap, path_cells, out_cap), b = finish_lean_work_bytes(n, m, bkp_cap);
    return a > b ? a : b;
}
AMBI_HD int64_t express_work_bytes(int n, int m, int K, int bkp_cap, int path_cells, int out_cap) {
    return pad8(prepare_work_bytes(n, m, K)) + pad8(4ll * (n + 2)) + pad8(first_work_bytes(n, bkp_cap)) + 64 + express_finish_bytes(n, m, bkp_cap, path_cells, out_cap);
}
AMBI_HD ExpressWork carve_express(uint8_t* base, int n, int m, int K, int bkp_cap) {
    ExpressWork W;
    int64_t o = 0;
    W.P = carve_prepare(base, n, m, K); o += pad8(prepare_work_bytes(n, m, K));
    W.target = reinterpret_cast<int32_t*>(base + o); o += pad8(4ll * (n + 2));
    W.F = carve_first(base + o, n, bkp_cap); o += pad8(first_work_bytes(n, bkp_cap));
    W.flags = reinterpret_cast<int32_t*>(base + o); o += 64;
    W.finish = base + o;
    return W;
}
// order 0: repeatedly the lowest-numbered node all of whose predecessors are placed
template <class G>
AMBI_HD bool first_order(const G& g, const Dag& D, uint8_t* ord) {
    const int K = D.K;
    if (g.size() >= 64) {
        // node j in thread j: "all my predecessors are placed and I am not" is one ballot per step
        const int j = g.tid();
        uint64_t waiting = j < K ? D.pred[j & 63] : ~0ull;   // my predecessors not yet placed
        bool open_ = j < K;                                  // I am not placed yet
        bool ok = K > 0;
        for (int d = 0; d < K; d++) {
            const uint64_t av = g.ballot_u64(open_ && waiting == 0);
            if (!av) { ok = false; break; }
            const int v = ctz64(av);
            if (j == 0) ord[d] = (uint8_t)v;
            waiting &= ~(1ull << v);
            open_ = open_ && j != v;
        }
        if (!ok && j == 0) ord[0] = 0xFF;
        g.sync();
        return ok;
    }
    if (g.tid() == 0) {
        uint64_t placed = 0;
        for (int d = 0; d < K; d++) {
            const uint64_t av = avail_mask(D.pred, K, placed);
            if (!av) { ord[0] = 0xFF; break; }   // cyclic relation: no order at all
            const int v = ctz64(av);
            ord[d] = (uint8_t)v;
            placed |= 1ull << v;
        }
    }
    g.sync();
    return K > 0 && ord[0] != 0xFF;
}

template <class GW, class GB>
AMBI_HD bool stage_express(const GW& gw, const GB& gb, int role, const BatchArgs& A, int u, uint8_t* work) {
    const UnitIn U = A.units[u];
    const int n = U.n_seg, m = U.n_junc, K = U.n_elem;
    ExpressWork W = carve_express(work, n, m, K, U.bkp_cap);
    UnitOut* out = unit_out(A.results, u);
    int32_t* fl = W.flags;   // [0] bias, [1] no_fbi, [2] dag status, [3] elements placed by order 0 (or negative status), [4] L, [5] verdict
    double* inv_sum_slot = reinterpret_cast<double*>(fl + 6);
    const bool forward = !(A.flags & FLAG_REVERSED);
    AMBI_MARK(A, gb, u, 9);   // (diagnostics: marks 9-12 are the scan stage's; that stage leaves an express unit alone)
    if (role == 0 || role < 0) {
        int bias = 1; bool no_fbi = false; double inv_sum = 0;
        prep_junctions(gw, A, u, U, W.P, &bias, &no_fbi, &inv_sum);
        if (gw.tid() == 0) { fl[0] = bias; fl[1] = no_fbi ? 1 : 0; *inv_sum_slot = inv_sum; }
        // front part of indelBFB (needs the junction ends only) into the finish stage's work area, which nothing uses yet:
        // this wavefront is done long before the DAG side
        int nsv = -1;
#if defined(__HIP_DEVICE_COMPILE__)
        if (role == 0) {
            FinishLeanWork FL = carve_finish_lean(W.finish, n, m, U.bkp_cap);
            for (int j = gw.tid(); j < m; j += gw.size()) FL.ends[j] = W.P.ends[j];
            gw.sync();
            nsv = indel_collect(gw, n, FL.ends, m, IndelScratch{FL.sv, FL.taken, FL.has_ext, nullptr, FL.first, FL.last});
        }
#endif
        if (gw.tid() == 0) fl[8] = nsv;
        AMBI_MARK(A, gw, u, 6);
    }
    if (role == 1 || role < 0) {
        int dst = prep_dag(gw, A, u, U, W.P, role < 0 ? W.target : nullptr);   // (a third wavefront fills the target CN, below)
        int placed = -1, L = 0;
        AMBI_MARK(A, gw, u, 7);
        if (dst == ST_OK) {
            gw.sync();
            if (first_order(gw, *W.P.dag, W.F.ord)) {
                AMBI_MARK(A, gw, u, 8);
                placed = kRegsGiveUp;
#if defined(__HIP_DEVICE_COMPILE__)
                if constexpr (GW::kLaneArrays) placed = eval_place_regs(*W.P.dag, W.F.ord, forward, W.F.bkp, U.bkp_cap, &L);   // breakpoint cells in registers (up to 256 of them)
#endif
                if (placed == kRegsGiveUp) placed = eval_place(gw, *W.P.dag, W.F.ord, forward, W.F.bkp, U.bkp_cap, &L);
            }
            else placed = 0;   // no order: nothing assembles here; the lattice stage reports R = 0
        }
        if (gw.tid() == 0) { fl[2] = dst; fl[3] = placed; fl[4] = L; }
    }
    if (role == 2 && K > 0 && gw.tid() < 64 && (gb.tid() >> 6) == 2) {
        // targetCN (localhap.cpp:222-232) straight from the elements in HBM, beside the other two sides
        target_cn_g(gw, A.elems + U.elem_off, K, n, W.target, W.target);
    }
    gb.sync();
    AMBI_MARK(A, gb, u, 10);
    const int bias = fl[0], dag_status = fl[2];
    const bool no_fbi = fl[1] != 0;
    const int status = prep_status(U, no_fbi, dag_status);
    const bool have_target = !(no_fbi && !U.has_components) && !U.infeasible && dag_status != ST_ERR_NO_ELEMENTS;
    prep_copy_out(gb, A, u, U, W.P, W.target, have_target, status);
    AMBI_MARK(A, gb, u, 2);
    // the fold-back map in the form the evaluation reads (as load_first_work would fetch it from the blob)
    for (int i = gb.tid(); i <= n; i += gb.size()) {
        const int ji = W.P.inv_junc[i];
        W.F.inv_src[i] = (int16_t)(ji >= 0 ? iabs(W.P.ends[ji].s) : 0);
        W.F.inv_tgt[i] = (int16_t)(ji >= 0 ? iabs(W.P.ends[ji].t) : 0);
    }
    if (gb.tid() == 0) { prep_header(out, status, bias, K, 0, *inv_sum_slot); runs_none(A, u); }   // R comes from the lattice stage
    gb.sync();
    // finish: the lean stage (runs of the breakpoint path; any path length), the full stage when the lean one hands the
    // unit over or would leave it to the direct full-stage launch
    bool mirrored = false;   // the final path went to the mailbox slot while it was written (uniform over the workgroup)
    auto finish = [&]() {
        bool need_full = U.direct_full && A.direct_full_on;
        if (!need_full) {
            rcell_t* mirror = A.mail ? reinterpret_cast<rcell_t*>(A.mail + A.mail_off[u] + mail_layout(U.path_cap, U.out_cap).path) : nullptr;
            stage_finish_lean(gb, A, u, W.finish, mirror, fl[8]);
            gb.sync();
            need_full = out->status == ST_REFINISH;
            mirrored = mirror != nullptr && !need_full && out->status == ST_OK;
        }
        if (need_full) stage_finish(gb, A, u, W.finish);
        gb.sync();
        if (gb.tid() == 0) out->reserved = 1;   // done: the scan / finish kernels behind leave the unit alone
        gb.sync();
    };
    if (status == ST_SHORTCUT || status == ST_INFEASIBLE) { finish(); return false; }   // (that branch writes the path itself: ordinary copy)   // the reference path: the finish stage writes it
    if (status != ST_OK) return false;
    const int placed = fl[3], L = fl[4];
    AMBI_MARK(A, gb, u, 3);
    if (role == 1 || role < 0) {
        int v = placed < 0 ? placed : eval_finish(gw, placed, K, W.F.bkp, L, InvMap{W.F.inv_src, W.F.inv_tgt}, true);
        if (A.inject_valid && A.inject_off[2 * (int64_t)u] >= 0) v = 0;   // injected verdicts (diagnostics) are indexed with R: the scan kernel applies them
        if (gw.tid() == 0) fl[5] = v;
    }
    gb.sync();
    AMBI_MARK(A, gb, u, 11);
    if (fl[5] != 1) return false;   // order 0 does not assemble (or ends in an error): the ordinary scan takes the unit, from order 0
    {
        const UnitLayout Lay = unit_layout(n, U.bkp_cap, U.path_cap, U.out_cap);
        cell_t* dst = reinterpret_cast<cell_t*>(A.results + U.res_off + Lay.bkp);
        for (int i = gb.tid(); i < L; i += gb.size()) dst[i] = W.F.bkp[i];
        if (gb.tid() == 0) { out->first_valid = 0; out->first_forward = forward ? 1 : 0; out->bkp_len = L; out->evaluated = 1; }
    }
    gb.sync();
    finish();
    AMBI_MARK(A, gb, u, 12);
    return mirrored;
}

}  // namespace ambi
