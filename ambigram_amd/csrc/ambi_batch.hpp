// ambi_batch.hpp -- HBM layout of a batch of units and the kernel argument block.
//
// Inputs (uploaded once, resident in HBM while a batch is run any number of times):
//   UnitIn[U]           fixed-size descriptor per unit: sizes + offsets into the pools below
//   seg_cn   f64        (n+1) per unit, slot 0 unused, indexed by LOCAL segment id      ("8n"  of SURVEY 8d)
//   junc_ends, junc_cn  strand-signed ends (4 bytes) + copy number (f64) per junction, reference junction order preserved
//                       (12 of the 24 bytes of the host's Junction record -- SURVEY 8d counts "24m" -- are all the device reads)
//   elems    Element    16-byte records (solution columns with value > 0)               ("16K")
// Working set (device only): per-unit Dag, ideal tables, the order-table arena (R x K uint8 rows, 16-byte aligned
// per unit), bkp cells.
// Results: one contiguous blob  [UnitOut[U]] [per-unit variable part ...]  so that a whole batch is fetched by a
// single D2H copy or gathered by a single RCCL collective.
#pragma once
#include "ambi_common.hpp"
#include "ambi_finish.hpp"

namespace ambi {

struct WideUnit;

struct UnitIn {
    int32_t n_seg;        // n: local segments 1..n
    int32_t seg_base;     // absolute id = local + seg_base
    int32_t n_junc;       // m
    int32_t n_elem;       // K
    int32_t infeasible;   // .sol said Infeasible
    int32_t bkp_cap;      // cells
    int32_t path_cap;     // cells
    int32_t out_cap;      // output junction records
    int64_t seg_off;      // into seg_cn pool (doubles)
    int64_t junc_off;     // into juncs pool (records)
    int64_t elem_off;     // into elems pool (records)
    int64_t res_off;      // byte offset of this unit's variable part inside the result blob
    int64_t ideal_off;    // slot offset into the ideal-table pools
    int32_t ideal_cap;    // power of two
    int32_t has_components;   // .juncs components exist for this chromosome (localhap.cpp:158-164)
    int32_t direct_full;      // the unit has same-strand non-adjacent junctions (deletion / duplication / insertion candidates of indelBFB,
                              // LGM.cpp:3746-3837): it goes straight to the full finish stage, the lean stage leaves it alone
    int32_t pad_;
};

// fixed-size result header of one unit
struct UnitOut {
    int32_t status;          // ambi::Status
    int32_t bias;            // localhap.cpp:141-146
    int32_t K;
    int32_t bkp_len;         // L of the first valid order after imperfectFBI
    int32_t path_len;        // P  (getBFB result, LGM.cpp:3660-3671)
    int32_t path_indel_len;  // P' after indelBFB
    int32_t indel_printed;   // reference prints the indel caption + path (LGM.cpp:3835-3836)
    int32_t n_out_junc;
    int32_t first_forward;   // orientation of the first valid assembly: 1 forward seed, 0 reversed seed, -1 none
    int32_t evaluated;       // number of order evaluations performed by the reference's sequential scan (E)
    int64_t num_orders;      // R
    int64_t first_valid;     // index of the first valid order or -1
    int64_t order_off;       // byte offset of this unit's rows in the order-table arena (-1: not materialised)
    double inv_cn_sum;       // localhap.cpp:150-153
    int32_t path_ind_stored; // 1: indelBFB edited the path and path_ind holds it; 0: path_ind equals path and was not written
    int32_t reserved;
};

// Variable part of a unit inside the result blob, in this order (each array padded to 8 bytes):
//   junc_cn   f64  2*(n+1)
//   seg_cn    f64  (n+1)         after getIndelBias
//   target_cn i32  (n+1)
//   inv_src   i16  (n+1)   inv_tgt i16 (n+1)   inv_junc i32 (n+1)
//   bkp       i16  bkp_cap
//   path      i16  path_cap      LOCAL signed ids (getBFB result); absolute id = abs_cell(id, seg_base)
//   path_ind  i16  path_cap      after indelBFB
//   out_junc  OutJunc out_cap
AMBI_HD int64_t pad8(int64_t b) { return (b + 7) & ~int64_t(7); }
// A path cell of the result blob: the local signed segment id (2 bytes; round 1 stored absolute ids in 4 -- the paths are
// 0.2 GB per step of the bench batch, written beside the store-bound enumerate kernel).  Readers add the unit's base.
typedef int16_t rcell_t;
AMBI_HD int32_t abs_cell(int v, int seg_base) { return v > 0 ? v + seg_base : v - seg_base; }
struct UnitLayout {
    int64_t junc_cn, seg_cn, target_cn, inv_src, inv_tgt, inv_junc, bkp, path, path_ind, out_junc, total;
};
AMBI_HD UnitLayout unit_layout(int n, int bkp_cap, int path_cap, int out_cap) {
    UnitLayout L;
    int64_t o = 0;
    L.junc_cn = o; o += pad8(int64_t(16) * (n + 1));
    L.seg_cn = o; o += pad8(int64_t(8) * (n + 1));
    L.target_cn = o; o += pad8(int64_t(4) * (n + 1));
    L.inv_src = o; o += pad8(int64_t(2) * (n + 1));
    L.inv_tgt = o; o += pad8(int64_t(2) * (n + 1));
    L.inv_junc = o; o += pad8(int64_t(4) * (n + 1));
    L.bkp = o; o += pad8(int64_t(2) * bkp_cap);
    L.path = o; o += pad8(int64_t(sizeof(rcell_t)) * path_cap);
    L.path_ind = o; o += pad8(int64_t(sizeof(rcell_t)) * path_cap);
    L.out_junc = o; o += pad8(int64_t(sizeof(OutJunc)) * out_cap);
    L.total = o;
    return L;
}

// kernel argument block (device pointers)
// A batch is run as one or more SLICES (contiguous unit ranges) whose kernel chains are issued on different HIP streams,
// so that the HBM-bound enumerate kernel of one slice overlaps the latency-bound kernels of the others.  Per-unit
// arrays are indexed by the GLOBAL unit index  unit_base + local index; blk_off / orders_needed are per slice.
struct BatchArgs {
    int32_t n_units;             // units of this slice
    int32_t unit_base;           // first unit of this slice
    int64_t arena_base;          // byte offset of this slice's region inside the order arena
    uint32_t flags;
    int32_t first_budget;        // orders the first-valid kernel tries per orientation before declaring PENDING
    int32_t target_lanes;        // enumerate kernel: lanes to spread the rows of the batch over (sets rows per lane)
    int32_t enum_stack_lds;      // enumerate kernel: LDS bytes per wave for the per-lane DFS stacks
    int32_t enum_auto_lds;       // enumerate kernel: LDS bytes per wave for the compact automaton copy
    int32_t block_lds;           // block emission: LDS bytes per workgroup for a unit's image (directory + suffix rows)
    int32_t block_scratch_lds;   // ambi_blocks_build_kernel: extra LDS bytes for the automaton copy used while building
    int32_t block_max;           // block emission: largest block (rows); <= kBlockMaxLimit
    int32_t build_in_emit;       // 1: units with one work block get their image built in LDS by the enumerate workgroup
    int32_t emit_interleave;     // 1: the blocks of a work block are dealt round-robin to the workgroup's waves (one compact store window)
    int32_t finish_path_cells;   // path cells the full finish stage can hold in group memory (<= kPathLdsCells; 0: that limit)
    int32_t order_align;         // every unit's table starts at a multiple of this many bytes of the arena (power of two >= 16)
    int32_t block_dfs;           // 1: units whose directory does not fit get the directory-free image (block walk at emission)
    int32_t* unit_fallback;      // [U] set by ambi_blocks_build_kernel when a unit's image does not fit block_lds
    uint8_t* block_img;          // [U][block_lds] images (built once per unit, copied to LDS by the emitting workgroups)
    int32_t* block_hdr;          // [U][8]  BlockImageHeader
    const UnitIn* units;
    const double* seg_cn;
    const double* junc_cn;      // [sum m] copy number of every junction (the 24-byte records stay on the host: the device reads 12 bytes per junction)
    const JuncEnds* junc_ends;  // [sum m] the strand-signed ends of every junction (4 bytes; the finish stages need nothing else of a junction)
    const Element* elems;
    Dag* dags;                   // [U]
    uint8_t* results;            // result blob; UnitOut[U] at the front
    // ideal tables (pools indexed by UnitIn::ideal_off; cap = UnitIn::ideal_cap slots per unit)
    uint64_t* ideal_keys;        // [slots]    lattice search in HBM (large lattices): hash keys
    uint64_t* ideal_cnt;         // [slots]    first half of a unit's range: ideal masks by index, second half: counts
    int32_t* ideal_pos;          // [slots]    hash slot -> ideal index
    uint32_t* ideal_link;        // [4*slots]  child links
    int32_t* ideal_lvl_off;      // [U][kMaxNodes+3]
    int32_t* ideal_counter;      // [U][2]
    uint64_t* auto_avail;        // [slots/2]
    uint64_t* auto_cnt;          // [slots/2]
    int32_t* auto_cbase;         // [slots/2 + U]
    uint16_t* auto_child;        // [U * 4 * ideal_cap]
    uint32_t* auto_nblk;         // [slots/2]  emission blocks below every ideal (for this run's block_max)
    uint8_t* auto_depth;         // [slots/2]  level of every ideal
    // order table
    uint8_t* first_rows;         // [U][first_budget][kFirstRowStride] the first orders of every unit, unranked by the prepare stage (nullptr: the scan reads the order table)
    uint8_t* order_arena;
    int64_t order_arena_bytes;   // bytes of this slice's region
    int64_t* blk_off;            // [n_units+1] enumerate work-block prefix of this slice (local index)
    int32_t* rows_per_lane;      // [U]   T of the unit (global index)
    int32_t* n_pending;          // [1]   whole batch
    int64_t* orders_needed;      // [1] bytes the order tables of this slice need (for arena sizing)
    // scratch for indel grouping (per unit: sv[m], grp[2m+4] ints, taken[m] bytes)
    int32_t* scratch_i32;
    int64_t* scratch_off;        // [U] offset (ints) into scratch_i32
    // single-slice runs hand the two host-visible scalars over without copy commands: the prepare kernel zeroes
    // n_pending, the plan kernel stores orders_needed and the finish kernel n_pending straight into pinned host memory
    int32_t zero_pending;        // 1: block 0 of the prepare kernel zeroes *n_pending
    int32_t* host_pending;       // device address of a pinned host int32 (nullptr: the host copies n_pending itself)
    int32_t* blocks_done;        // [1] finished workgroups of the lean finish kernel (the last one reports n_pending and resets it)
    int32_t* refin_list;         // [U] units the lean finish stage hands to the full stage (SVs that chain or edit the path), in any order
    int32_t* express_seq;        // pinned host int: the express kernel's last workgroup stores run_seq here (the host spins on it)
    int32_t run_seq;
    int32_t* express_left;       // pinned host int: set to 1 by the express kernel when a unit is left to the ordinary scan / finish kernels
    int32_t finish_retry;        // 1 (a full-stage launch whose path area, finish_path_cells, is smaller than the units' capacities): a path that
                                 // does not fit the area is handed to the list kernel behind (refin_list), which has the full area
    int32_t direct_full_on;      // 1: units with UnitIn::direct_full are served by a full-stage launch of their own (the lean stage skips them)
    int32_t edit_cap_limit;      // tests (env AMBI_EDIT_RUN_CAP): upper limit of the edit stage's run lists, so that its hand-over to the full stage can be reached; 0: none
    int32_t* refin_count;        // [1] entries of refin_list; zeroed before the lean kernel, read by the full-stage kernel behind it
    int64_t* host_needed;        // device address of a pinned host int64 (nullptr: the host copies orders_needed itself)
    int64_t* stage_clk;          // [U][kStageSlots] shader-clock marks inside the per-unit stages (nullptr: off; env AMBI_STAGE_PROFILE)
    // diagnostics hook (ambi_batch_debug_inject_validity): verdicts that REPLACE the outcome of evaluating an order, so that
    // the control flow around the evaluation (scan budget, parallel search, minimum index, orientation flip, --all) can be
    // driven to places no known input reaches.  nullptr in every ordinary run.
    const int8_t* inject_valid;  // pool: per unit 2*R verdicts, first orientation "forward seed" then "reversed seed"; 127 = evaluate
    const int64_t* inject_off;   // [U][2] {offset into the pool or -1, number of verdicts}
    // --all (ambi_all_kernel): validity bitmaps of both passes, one bit per order
    uint64_t* all_bits;          // pool of 64-order words
    const int64_t* all_off;      // [U+1] first word of every unit's pass-0 map; its pass-1 map follows (ceil(R/64) words each)
    int32_t* all_count;          // [U][2] valid orders per pass
    int32_t* all_flags;          // [U]    != 0: an order on which the reference's behaviour is undefined was met (behind the bitmaps in the same pool)
    int32_t all_rows_from_table; // experiment (env AMBI_ALL_TABLE=1): the lane kernel reads its orders from the order table instead of unranking them
    int32_t all_rank, all_world; // --all over several ranks (one wide sample): a rank evaluates the chunks c with c % world == rank,
                                 // plus the LAST chunk of every unit (every rank must know whether the orientation flips)
    // small batches (express path): what a caller of ONE sample wants on the host -- header, final path(s), output junctions --
    // is mirrored by the express kernel into a pinned host "mailbox" (one slot per unit, MailLayout), so that no copy command
    // follows the kernel; and the two late verdicts that can still void an express result reach the host through pinned words
    // wide units (64..127 DAG nodes, ambi_wide.hpp): their DAG / lattice working sets, and which unit has which
    WideUnit* wide;       // [number of wide units]
    const int32_t* wide_index;   // [U] index into `wide`, -1 for ordinary units (nullptr: the batch has no wide unit)
    uint8_t* mail;               // device address of the pinned mailbox (nullptr: none)
    const int64_t* mail_off;     // [U] byte offset of every unit's slot
    int32_t* plan_seq;           // pinned host int: the plan kernel stores run_seq here once orders_needed / late_flag are final
    int32_t* late_flag;          // pinned host int: != 0 when the lattice or plan stage refused a unit AFTER the express stage published it
    // small batches: the lattice of every unit is built by a kernel of its own BESIDE the express kernel (stage_lattice_own: it
    // constructs the DAG a second time instead of waiting for the express kernel's); its outcome is parked here and merged into
    // the headers by the plan kernel, which runs behind both (nullptr: the lattice stage writes the headers itself)
    int32_t* lat_seq;            // pinned host int: the last wave of the side lattice kernel stores run_seq here ...
    int32_t* lat_unsure;         // pinned host int: ... after setting this to 1 unless EVERY unit's lattice is fine and all order tables together
                                 // (every unit counted, also those the express stage ends without a reconstruction) fit the arena:
                                 // 0 = nothing behind the express kernel can void what it published
    int64_t* lat_sum;            // [2] device: {bytes of the order tables so far, waves done} of the side lattice kernel
    uint64_t* lat_R;             // [U] number of orders
    int32_t* lat_status;         // [U] ST_OK or the status the DAG / lattice stage ends with
    // The FINAL path of every unit (the path after indelBFB) in run-length form, written by the finish stage that produces the path --
    // what ambi_batch_runs_to_host copies to the host (a run = a stretch of cells counting up by one: `3+4+5+` = start 3, length 3).
    // run_cnt[u] = its runs (0: no path; -1: more runs than the unit's slots -- the pack kernels serve such a batch), run_cells[u] its
    // cells; unit u's runs sit at run_slot[u] .. run_slot[u+1] of run_start (absolute signed ids) / run_len.  nullptr: not collected.
    int32_t* run_cnt;
    int32_t* run_cells;
    int32_t* run_start;
    int32_t* run_len;
    const int64_t* run_slot;
};

// Mailbox slot of one unit: [UnitOut, 128 bytes] [path: path_cap cells] [path after indelBFB: path_cap cells] [output junctions:
// out_cap records], each part on a 16-byte boundary; only the used prefix of every part is written.
struct MailLayout { int64_t path, path_ind, out_junc, total; };
AMBI_HD int64_t pad16(int64_t b) { return (b + 15) & ~int64_t(15); }
AMBI_HD MailLayout mail_layout(int path_cap, int out_cap) {
    MailLayout M;
    M.path = 128;
    M.path_ind = M.path + pad16(int64_t(sizeof(rcell_t)) * path_cap);
    M.out_junc = M.path_ind + pad16(int64_t(sizeof(rcell_t)) * path_cap);
    M.total = M.out_junc + pad16(int64_t(sizeof(OutJunc)) * out_cap);
    return M;
}

// stage-level timing marks (diagnostics only; one predictable branch per mark when off)
constexpr int kStageSlots = 32;
#define AMBI_MARK(A, g, u, slot) \
    do { if ((A).stage_clk && (g).tid() == 0) (A).stage_clk[(int64_t)(u) * kStageSlots + (slot)] = stage_clock(); } while (0)

AMBI_HD UnitOut* unit_out(uint8_t* results, int u) { return reinterpret_cast<UnitOut*>(results) + u; }

}  // namespace ambi
