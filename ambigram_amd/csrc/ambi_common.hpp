// ambi_common.hpp -- shared definitions of the MI355X BFB reconstruction engine.
//
// Vocabulary (follows the reference, deepomicslab/Ambigram):
//   unit      = one chromosome of one sample: the independent work item of `--op bfb`
//               (loop body localhap.cpp:111-265).
//   element   = a selected BFB pattern p(a,b) or loop l(a,b,cn) of the ILP solution (localhap.cpp:204-211).
//   node      = DAG node (LocalGenomicMap.cpp:3276-3378), order = one topological order (:3380-3409).
//   bkp       = breakpoint path assembled by getBFB (:3514-3697): pairs (first,last) of signed vertex ids.
//   vertex    = signed LOCAL segment id: +i = (i,'+'), -i = (i,'-'); local ids are 1..n inside the unit,
//               absolute id = local + seg_base.
//
// All algorithm code in the ambi_*.hpp headers is SPMD code over a thread-group policy `G` (ambi_group.hpp):
// the HIP kernels instantiate it with a wavefront (64 lanes) or a workgroup, the CPU host-simulation used by
// `pytest -m "not gpu"` instantiates the very same source with a 1-thread group.
#pragma once
#include <stdint.h>
#include <cstdlib>
#include <cstring>

#if defined(__HIPCC__)
#define AMBI_HD __host__ __device__ inline
#else
#define AMBI_HD inline
#endif

namespace ambi {

// ---- limits of the device engine (documented in DESIGN.md; exceeded -> unit status, never silent) ----
constexpr int kMaxNodes = 63;        // K: DAG nodes per unit (64-bit masks; all-ones is the empty hash key)
constexpr int kMaxSegLocal = 32767;  // local segment ids are stored as int16 in bkp/path cells

// ---- unit status codes (also the negative return codes of the C-ABI, see include/ambigram_hip.h) ----
enum Status : int32_t {
    ST_OK = 0,
    ST_SHORTCUT = 1,            // no fold-back inversion: reference path 1+..n+ (localhap.cpp:164-170)
    ST_INFEASIBLE = 2,          // .sol said Infeasible (localhap.cpp:213-220)
    ST_NO_VALID_ORDER = 3,      // no topological order assembles in either orientation (path stays empty)
    ST_PENDING = 4,             // internal: first-valid search budget exhausted, parallel search needed
    ST_REFINISH = 5,            // internal: the lean finish stage met SVs that chain or edit the path; the full stage takes the unit
    ST_ERR_TOO_MANY_NODES = -10,   // K > kMaxNodes
    ST_ERR_NO_ELEMENTS = -11,      // K == 0: the reference indexes an empty order (UB)
    ST_ERR_REF_UB = -12,           // the reference would read out of bounds on this input
    ST_ERR_BKP_CAPACITY = -13,
    ST_ERR_PATH_CAPACITY = -14,
    ST_ERR_ORDERS_CAPACITY = -15,  // R*K does not fit the order-table arena
    ST_ERR_IDEALS_CAPACITY = -16,
    ST_ERR_BAD_INPUT = -17,
    ST_ERR_OUTJUNC_CAPACITY = -18,
};

// run flags
constexpr uint32_t FLAG_REVERSED = 1u;   // --reversed (localhap.cpp:37)
constexpr uint32_t FLAG_ALL = 2u;        // --all      (localhap.cpp:38)
constexpr uint32_t FLAG_LAZY_ORDERS = 4u; // the order tables (allTopologicalOrders' by-product) are not written by the run: on demand only

// An element of the ILP solution, 16 bytes ("16K" term of SURVEY.md 8d).
struct Element {
    int32_t is_loop;   // 0 = pattern p(a,b), 1 = loop l(a,b)
    int32_t a, b;      // LOCAL segment ids, a <= b
    int32_t cn;        // copy number (> 0)
};

// Junction record inside a unit (both ends inside [start,end]); dirs: +1 / -1.
struct Junction {
    int32_t src, tgt;      // LOCAL segment ids
    int8_t sdir, tdir;
    int8_t same_chr;       // always 1 inside a unit; kept for the record layout (24 bytes, SURVEY.md 8d "24m")
    int8_t pad0;
    int32_t pad1;
    double cn;
};
static_assert(sizeof(Junction) == 24, "junction record is 24 bytes");

// Strand-signed ends of a junction (edge A: sign = strand, Junction.cpp:27-39): all that the per-unit scans need
// besides the copy number; 4 bytes per junction in group memory instead of the 24-byte record.
struct JuncEnds { int16_t s, t; };
struct JuncView {
    const JuncEnds* e;       // [m] group memory
    const double* cn;        // [m] the unit's junction copy numbers in HBM
};
AMBI_HD JuncEnds junc_ends(const Junction& j) {
    JuncEnds E;
    E.s = (int16_t)(j.sdir > 0 ? j.src : -j.src);
    E.t = (int16_t)(j.tdir > 0 ? j.tgt : -j.tgt);
    return E;
}

AMBI_HD int a_src(const Junction& j) { return j.sdir > 0 ? j.src : -j.src; }   // edge A (Junction.cpp:27-39)
AMBI_HD int a_tgt(const Junction& j) { return j.tdir > 0 ? j.tgt : -j.tgt; }
AMBI_HD int b_src(const Junction& j) { return -a_tgt(j); }                      // edge B = complement
AMBI_HD int b_tgt(const Junction& j) { return -a_src(j); }

AMBI_HD int iabs(int v) { return v < 0 ? -v : v; }

// DAG of one unit after constructDAG (LGM.cpp:3276-3378), flat.
struct Dag {
    int32_t K;
    int32_t pat[64][3];       // node2pat[i]  = (a,b,cn) or a==0 when the slot is empty
    int32_t loop[64][3];     // node2loop[i] = (a,b,cn) after the libstdc++ sort (:3303), a==0 when empty
    uint64_t succ[64];     // adj[i] as a bit set (duplicates in the reference's lists do not matter)
    uint64_t pred[64];     // transposed
};

// shader-clock marks inside the per-unit stages (diagnostics, env AMBI_STAGE_PROFILE): clk = the unit's mark array or nullptr
AMBI_HD int64_t stage_clock() {
#if defined(__HIP_DEVICE_COMPILE__)
    return (int64_t)__builtin_readcyclecounter();
#else
    return 0;
#endif
}
template <class G>
AMBI_HD void clk_mark(const G& g, int64_t* clk, int slot) { if (clk && g.tid() == 0) clk[slot] = stage_clock(); }


// Environment switches of the engine (host code).  They are experiments and diagnostics, not a user interface: a switch that changes
// WHICH kernels run or how they are launched is honoured only when AMBI_EXPERIMENTS=1 is set as well (the tests, profiles/tools and the A/B
// scripts set it); without it the engine runs its measured defaults whatever else is in the environment.  Always honoured: the switches that
// only print or record (AMBI_DEBUG*, AMBI_STAGE_PROFILE) and the memory budget of the order-table arena.
inline const char* ambi_env(const char* name) {
    static const bool experiments = [] { const char* e = std::getenv("AMBI_EXPERIMENTS"); return e && std::atoi(e) != 0; }();
    if (experiments) return std::getenv(name);
    static const char* const always[] = {"AMBI_DEBUG", "AMBI_DEBUG_SIZES", "AMBI_DEBUG_LATENCY", "AMBI_DEBUG_QUARANTINE", "AMBI_STAGE_PROFILE", "AMBI_ARENA_MAX_BYTES"};
    for (const char* a : always) if (std::strcmp(a, name) == 0) return std::getenv(name);
    return nullptr;
}

}  // namespace ambi
