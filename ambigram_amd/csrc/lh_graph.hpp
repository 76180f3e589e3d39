// lh_graph.hpp -- host-side flat breakpoint-graph model and the text formats around the BFB path.
//
// Replaces the pointer graph of the reference (Graph/Segment/Vertex/Edge/Junction/Weight)
// with SoA vectors: a vertex is a signed segment id, a junction is (src, sdir, tgt, tdir, cn) and its two
// complementary edges are derived on the fly.
#pragma once
#include <memory>
#include <unordered_map>
#include <stdint.h>
#include <string>
#include <vector>

namespace ambi {

struct TrxBefore;
struct LhGraph {
    // PROP I1 / C1 (TRX-BFB): this graph is the REBUILT one and `trx` holds what leads back to the graph of the file (below)
    std::shared_ptr<TrxBefore> trx;
    // header keys (Graph.cpp:140-167)
    std::string sample_name, ploidy;
    std::vector<double> avg_coverages;
    double avg_cov_raw = -1, avg_virus_dp = -1, avg_cov_junc = 0, purity = -1;
    double avg_tumor_ploidy = -1, avg_ploidy = 0;      // defaults of Graph::Graph(const char*) Graph.cpp:36-41
    int virus_seg_start = 0; bool virus_seg_start_set = false;
    int expected_ploidy = 0;
    double ratio = 0; bool ratio_set = false;
    double haploid_depth = 0, avg_coverage = 0;
    // segments (file order; ids must be 1..N in file order, localhap.cpp:96 / LGM.cpp:3734 rely on it)
    std::vector<int32_t> seg_id, seg_chr, seg_start, seg_end, seg_partition;
    std::vector<std::string> seg_chrom;
    std::vector<double> seg_cov, seg_cn;
    // junctions (file order, duplicates dropped as Graph.cpp:592-595 does)
    std::vector<int32_t> j_src, j_tgt;
    std::vector<int8_t> j_sdir, j_tdir;     // +1 / -1
    std::vector<double> j_cov, j_cn;
    std::vector<uint8_t> j_inferred, j_bounded;
    // chromosomes
    std::vector<int32_t> source_ids, sink_ids;
    // PROP line (LGM.cpp:3941-3987)
    std::string main_chr;
    int ins_mode = 0, con_mode = 0;
    std::vector<std::string> ins_chr, con_chr;
    std::vector<int32_t> start_segs;
    // .juncs components (LGM.cpp:5096-5156)
    std::vector<std::vector<int32_t>> components;
    // lines the reference prints to stdout while loading (progress, SEG echoes, WARNs, .juncs breakpoints)
    std::vector<std::string> log;

    int n_seg() const { return (int)seg_id.size(); }
    int n_junc() const { return (int)j_src.size(); }
    int n_chr() const { return (int)source_ids.size(); }
    int find_junction(int src, int sdir, int tgt, int tdir) const;   // Graph.cpp:501-511, -1 if absent
    // indices behind find_junction / add_junction (the reference compares printed edge strings pair by pair, O(m^2)):
    // an edge and its complement share one key; segments by id
    std::unordered_map<uint64_t, int32_t> junc_index;
    std::unordered_map<int32_t, int32_t> seg_index;
    size_t segs_indexed = 0;
    bool add_junction(int src, int sdir, int tgt, int tdir, double cov, double cn, bool inferred, bool bounded);
};

// Error codes of the host layer (negative; 0 = ok).  Text via lh_error_string().
enum LhError {
    LH_OK = 0,
    LH_ERR_OPEN = -1,          // reference: "Cannot open file" + exit(1)          (Graph.cpp:111-114)
    LH_ERR_MALFORMED = -2,     // reference dereferences a NULL strtok result (segfault)
    LH_ERR_UNKNOWN_SEG = -3,   // reference: uncaught SegmentDoesNotExistException  (Graph.cpp:519)
    LH_ERR_SOURCE_SINK = -4,   // reference: assert sourceIds.size()==sinkIds.size() (Graph.cpp:227)
    LH_ERR_PLOIDY = -5,        // reference: "input error: ..." + exit(1)           (Graph.cpp:318-330)
    LH_ERR_SEG_IDS = -6,       // ids not 1..N in file order (reference indexes segs[id-1])
    LH_ERR_SOL_OPEN = -7,      // reference: "ILP error: cannot open file" + exit(1) (localhap.cpp:187-190)
    LH_ERR_LINE_TOO_LONG = -8, // > 8191 bytes: the reference's getline(line, 8192) never terminates
    LH_ERR_UNSUPPORTED = -9,   // TRX-BFB (PROP I1 / C1) where the reference reads what nothing has set: no junction between the listed
                               // chromosomes, a one-vertex path, a .juncs file together with these modes
};
const char* lh_error_string(int code);

int read_lh(const std::string& path, LhGraph& g);                 // Graph.cpp:109-237 + calculateHapDepth/CopyNum + PROP
int hap_depth(LhGraph& g);                                        // Graph.cpp:312-367 (part of read_lh; `--op sc_bfb` runs it a second time on its first graph)
void copy_num(LhGraph& g);                                        // Graph.cpp:369-405
int read_juncs(LhGraph& g, const std::string& path);
int write_lh(LhGraph& g, const std::string& path);               // Graph.cpp:239-266 Graph::writeGraph              // LGM.cpp:5096-5156 (needs partitions set)
void set_partitions(LhGraph& g);                                   // localhap.cpp:94-98

// ---- TRX-BFB, PROP I1 / C1 (localhap.cpp:79-88, :263): a translocation that happened BEFORE the BFB cycles -----------------------
// insertBeforeBFB / concatBeforeBFB (LGM.cpp:4195-4395) rebuild the graph -- the inserted segments spliced into the main
// chromosome, or the two chromosomes joined at their junction into one -- the BFB stages run on the rebuilt graph, and virusBFB
// (LGM.cpp:3839-3939) takes every path back to the segments of the file.  The reference ends the rebuild with
// `new Graph(mSegs, mJuncs, mSources, mSinks)`, a constructor that assigns through pointers it never initialises (Graph.cpp:25-34):
// its evident intent -- a graph made of copies of the four vectors -- is what is implemented here (the reference holds outputs of
// both modes: README.md:128-134, :148-157).  read_lh does the rebuild where localhap.cpp does it, between the PROP line and the
// partitions, so every caller sees the rebuilt graph; g.trx keeps the way back.
struct TrxBefore {
    LhGraph original;                    // the graph of the file (after the copy-number maths)
    std::vector<int32_t> original_of;    // [rebuilt segment id] -> id in the file; [0] unused
    std::vector<int32_t> unused_sv;      // junctions of the file (indices) whose segments are not both in the rebuilt graph, in file order
};
int trx_rebuild(LhGraph& g);             // by g.ins_mode / g.con_mode == 1; the lines the reference prints go to g.log
// virusBFB: `path` (rebuilt ids) -> ids of the file; lines = what the reference prints (caption + path of the first stage, and of
// the second stage when one of the unused junctions cuts the path).  LH_ERR_UNSUPPORTED where the reference aborts or leaves a vertex
// of the rebuilt graph in the path.
int trx_restore_path(const LhGraph& rebuilt, std::vector<int32_t>& path, std::vector<std::string>& lines);

struct SolFile {                                                   // localhap.cpp:192-212
    bool infeasible = false;
    double objective = 0;
    std::vector<int32_t> col, val;                                 // in file order (a later line overrides an earlier one)
};
int read_sol(const std::string& path, SolFile& s);

// printBFB (LGM.cpp:3411-3429): "1+2+|2-1-" with "||" between chromosomes
std::string format_path(const LhGraph& g, const int32_t* path, int len);

// column index <-> element (localhap.cpp:117-133): columns [0,numPat) are patterns in lexicographic (a,b) order,
// [numPat, 2 numPat) the loops.
bool column_to_element(int col, int start_id, int end_id, int* is_loop, int* a, int* b);

// BFB-TRX stitching (LGM.cpp:4052-4193); paths[chr] may be reverse-complemented in place, as in the reference.
void translocation_bfb(const LhGraph& g, std::vector<std::vector<int32_t>>& paths, std::vector<int32_t>& res);

struct OutJunction { int32_t u, v, count; };
// localhap.cpp:267-316: merge the junction steps of `path` into `out` (increase=false for the TRX pass)
void merge_out_junctions(std::vector<OutJunction>& out, const int32_t* path, int len, bool increase);

}  // namespace ambi
