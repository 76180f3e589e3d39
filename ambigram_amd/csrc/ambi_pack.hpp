// ambi_pack.hpp -- host-side packing of units into the flat batch layout (ambi_batch.hpp).
// Pure host C++ (no HIP): shared by the HIP engine (ambi_engine.hip) and the CPU host simulation.
#pragma once
#include <string>
#include <vector>

#include "ambi_batch.hpp"
#include "lh_graph.hpp"

namespace ambi {

constexpr int kPathCapLimit = 1 << 22;   // cells per unit in the result blob (the lean finish stage streams them; only the full stage,
                                         // which edits the path in LDS, is limited to kPathLdsCells)
constexpr int kDefaultIdealCap = 4096; // slots per unit (holds up to 2048 order ideals)

struct HostBatch {
    std::vector<UnitIn> units;
    std::vector<double> seg_cn;
    std::vector<Junction> juncs;
    std::vector<JuncEnds> junc_ends;   // junc_ends(juncs[i]) of every record
    std::vector<double> junc_cn;       // juncs[i].cn: ends + copy number (12 bytes) are all the device reads of a junction
    std::vector<Element> elems;
    std::vector<int64_t> scratch_off;       // per unit, ints
    std::vector<int64_t> run_slot;          // [U+1] slots of the units' final paths in run-length form (BatchArgs::run_slot); filled by finalize()
    std::vector<std::vector<int32_t>> junc_global;   // per unit: local junction index -> index in the sample's graph
    int64_t result_bytes = 0, ideal_slots = 0, scratch_ints = 0;
    int max_n = 0, max_m = 0, max_k = 0, max_bkp = 0, max_path = 0, max_out = 0;
    int ideal_cap = kDefaultIdealCap;
    int n_wide = 0;                         // units with 64..127 DAG nodes (ambi_wide.hpp)
    std::vector<int32_t> wide_index;        // [U] index among the wide units or -1; filled by finalize()
    bool any_sv = false;                    // some unit has a junction indelBFB would look at (neither adjacency nor fold-back): the full finish stage may be needed
    // diagnostics hook (ambi_batch_debug_inject_validity): verdict overrides per unit, see BatchArgs::inject_valid
    std::vector<int8_t> inject;
    std::vector<int64_t> inject_off;        // [U][2] {offset or -1, count}; filled by finalize()
    std::vector<std::vector<int8_t>> inject_unit;

    // Raw unit: local ids 1..n_seg, junctions already restricted to the unit.  Returns unit index or negative Status.
    int add_unit(int n_seg, int seg_base, const double* cn_local /*[n_seg], id 1 first*/, int n_junc, const int32_t* j_src,
                 const int32_t* j_tgt, const int8_t* j_sdir, const int8_t* j_tdir, const double* j_cn, int n_elem,
                 const int32_t* e_is_loop, const int32_t* e_a, const int32_t* e_b, const int32_t* e_cn, int infeasible,
                 int has_components);
    // One chromosome of a parsed .lh plus the .sol columns of that chromosome (localhap.cpp:111-232).
    // block / n_blocks: `--op sc_bfb` (localhap.cpp:540-566): the solution of graph `block` sits in the columns
    // [block*numComp, (block+1)*numComp) of a joint .sol; such a unit never takes the no-fold-back shortcut (that decision
    // is the caller's, from the first graph alone, localhap.cpp:505-512).
    int add_graph_chr(const LhGraph& g, int chr, const SolFile* sol, int block = 0, int n_blocks = 0);
    // Appends a copy of unit `u` of another batch (its packed inputs, capacities, junction map, injected verdicts): how a batch
    // is dealt over several devices (ambi_batch_run_sharded).  Returns the new unit's index.
    int add_unit_from(const HostBatch& src, int u);
    void finalize();                          // computes result/ideal/scratch offsets
    int64_t header_bytes() const { return int64_t(sizeof(UnitOut)) * (int64_t)units.size(); }
};

}  // namespace ambi
