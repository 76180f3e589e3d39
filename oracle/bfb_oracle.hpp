// oracle/bfb_oracle.hpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the `Ambigram --op bfb` hot path of deepomicslab/Ambigram
// (reference @ /root/reference, citations below are relative to it; "LGM.cpp" =
// src/LocalGenomicMap.cpp).  It exists to CHECK the HIP engine in
// ambigram_amd/csrc; only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may link or call it.  Nothing under ambigram_amd/ includes
// this file and the product never falls back to it.
//
// Style: deliberately close to the reference's own data flow (STL vectors,
// std::find / reverse iterators / std::sort with the reference comparator) so
// that iterator-arithmetic quirks are inherited from libstdc++ itself rather
// than re-derived.  A vertex (segment id, strand) is a signed int: +id / -id;
// the complement vertex is the negation.
//
// Pinning status (see DESIGN.md "Oracle"):
//   * .lh reader + copy-number maths (#1-#3): checked against the REAL reference
//     Graph.cpp compiled from /root/reference (oracle/_ref, Makefile target
//     `ref`), fixtures in tests/golden/graph_*.json.
//   * BFB stages (#6-#18,#20): LocalGenomicMap.cpp is unbuildable here (needs the
//     COIN-OR Cbc/Osi headers, absent from the image), so these are pinned by the
//     reference's own known answers: README.md:85-122 (6-seg example) and the
//     reference outputs recorded in SURVEY.md Appendix B.4/B.5.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

namespace oracle {

struct Seg {
    int id = 0;
    int chrId = -1;          // Graph.cpp:193-197 (index into SOURCE/SINK lists)
    std::string chrom;
    int start = 0, end = 0;
    double cov = 0, cn = 0;  // Weight: coverage + copy number (Weight.cpp:3-13)
    int partition = 0;       // localhap.cpp:94-98
};

struct Junc {
    int src = 0, tgt = 0;    // segment ids
    char sdir = '+', tdir = '+';
    double cov = 0, cn = 0;
    bool inferred = false, bounded = false;
    // edge A: (src,sdir)->(tgt,tdir); edge B: (tgt,!tdir)->(src,!sdir)  (Junction.cpp:27-39)
    int a_src() const { return sdir == '+' ? src : -src; }
    int a_tgt() const { return tdir == '+' ? tgt : -tgt; }
    int b_src() const { return -a_tgt(); }
    int b_tgt() const { return -a_src(); }
};

struct Graph {
    // header (Graph.cpp:140-167)
    std::string sampleName, ploidy;
    std::vector<double> avgCoverages;
    double avgCoverageRaw = -1, avgVirusDP = -1, avgCoverageJunc = 0, purity = -1;
    double avgTumorPloidy = -1, avgPloidy = 0;   // ctor Graph.cpp:36-41
    int virusSegStart = 0;  bool virusSegStartSet = false;
    int expectedPloidy = 0;
    double ratio = 0;  bool ratioSet = false;     // mRatio is uninitialised in the reference unless set (Graph.cpp:339-341)
    double haploidDepth = 0, avgCoverage = 0;
    std::vector<Seg> segs;
    std::vector<Junc> juncs;
    std::vector<int> sourceIds, sinkIds;
    std::vector<std::string> log;                 // lines the reference writes to stdout while loading

    Seg* segById(int id);                          // Graph.cpp:513-520 (first match, linear scan)
    bool hasSeg(int id) const;
};

// Graph.cpp:109-237.  Returns false (and sets err) where the reference would exit(1)/crash.
bool readGraph(const std::string& path, Graph& g, std::string& err);
// Graph.cpp:312-367 + 369-405
bool calculateHapDepth(Graph& g, std::string& err);
void calculateCopyNum(Graph& g);
// Graph.cpp:579-610 incl. duplicate test Graph.cpp:489-499
bool addJunction(Graph& g, int src, char sdir, int tgt, char tdir, double cov, double cn, bool inferred, bool bounded);
int findJunction(const Graph& g, int src, char sdir, int tgt, char tdir);   // Graph.cpp:501-511, -1 if none

struct Props {                                    // LGM.cpp:3941-3987
    std::string mainChr;
    int insMode = 0, conMode = 0;
    std::vector<std::string> insChr, conChr;
    std::vector<int> startSegs;
};
void readBFBProps(const std::string& lhPath, Props& p);

// LGM.cpp:5096-5156 (originalSegs is always empty on the supported modes)
void readComponents(Graph& g, const std::string& juncsPath, std::vector<std::vector<int>>& components,
                    std::vector<std::string>& log);

// ---- per-chromosome stages ----------------------------------------------------------------
using Inversions = std::unordered_map<int, int>;   // segment id -> junction index (LGM.cpp:3989-4050)

// LGM.cpp:3989-4050; juncCN is (endID+1) x 2 row-major
void getJuncCN(const Graph& g, int startID, int endID, Inversions& inv, std::vector<double>& juncCN);
// localhap.cpp:141-146
int computeBias(const Graph& g, int startID, int endID, const Inversions& inv, const std::vector<double>& juncCN);
// LGM.cpp:3699-3744 (mutates segment CN)
void getIndelBias(Graph& g, int startID, int endID);

// LGM.cpp:3254-3264 + localhap.cpp:117-133: key -> column index; std::map keeps string order
std::map<std::string, int> makeVariableIdx(int startID, int endID, int* numPat);

struct Dag {                                       // LGM.cpp:3276-3378
    std::vector<std::vector<int>> adj, node2pat, node2loop;
};
void constructDAG(const std::map<std::string, int>& variableIdx, const std::vector<int>& elementCN, Dag& dag);
// localhap.cpp:237-254 + LGM.cpp:3380-3409
void allTopologicalOrders(const Dag& dag, std::vector<std::vector<int>>& orders, size_t maxOrders = SIZE_MAX);

struct BfbResult {
    std::vector<int> path;                         // first valid assembly (LGM.cpp:3660-3671)
    std::vector<std::vector<int>> allPaths;        // --all: every valid order, in print order
    std::vector<long> allEvalIdx;                  // --all: for every entry of allPaths, the 0-based count of evaluations before it (pass * R + order index)
    std::vector<int> bkpFirst;                     // breakpoint path of the first valid order after imperfectFBI
    long firstValidOrder = -1;                     // index in `orders`
    int firstValidOrientationForward = -1;         // 1 forward seed, 0 reversed seed
    long evaluated = 0;                            // number of order evaluations performed (E in SURVEY 8d)
    bool undefinedBehaviour = false;               // the reference would have read out of bounds on some evaluated order
    bool undefinedOnValid = false;                 // the run has no defined result: a null dereference on some evaluated order, or a stray read
                                                   // on an order that placed all its elements (the scan stopped there)
};
// LGM.cpp:3431-3512
void imperfectFBI(const Graph& g, std::vector<int>& bkp, const Inversions& inv, bool* ub);
// LGM.cpp:3514-3697.  Appends printed lines to log.
void getBFB(const Graph& g, const std::vector<std::vector<int>>& orders, const Dag& dag, const Inversions& inv,
            bool isReversed, bool printAll, BfbResult& res, std::vector<std::string>& log);
// evaluate ONE order in one orientation (the body of the loop LGM.cpp:3519-3658); returns validity
// *ub: imperfectFBI touched the cell behind the end of bkp (stray read / write); *crash: a null dereference
bool evalOrder(const Graph& g, const std::vector<int>& order, const Dag& dag, const Inversions& inv, bool forwardDir,
               std::vector<int>& bkp, bool* ub, bool* crash);
void expandBkp(const std::vector<int>& bkp, std::vector<int>& path);   // LGM.cpp:3661-3670

// LGM.cpp:3746-3837; returns true if the caption+path were printed
bool indelBFB(const Graph& g, std::vector<int>& path, int startID, int endID, std::vector<std::string>& log);
// LGM.cpp:3411-3429
std::string formatPath(const Graph& g, const std::vector<int>& path);
// LGM.cpp:4052-4193
// `trace` (optional, test diagnostics): which branch every junction group took: "concat", "concat-skip", "insert",
// "insert-retry" (second attempt with the reverse-complemented group succeeded), "insert-skip"
void translocationBFB(const Graph& g, std::vector<std::vector<int>>& paths, std::vector<int>& res,
                      const std::string& mainChr, std::vector<std::string>& log, std::vector<std::string>* trace = nullptr);

// ---- TRX-BFB (PROP I1 / C1): the graph is rebuilt BEFORE the BFB stages and the path mapped back afterwards --------------
// insertBeforeBFB / concatBeforeBFB end in `new Graph(mSegs, mJuncs, mSources, mSinks)` (LGM.cpp:4293, 4393), a constructor that
// assigns through pointers it never initialises (Graph.cpp:25-34): undefined behaviour as written.  Its INTENT is unambiguous --
// a graph whose four vectors are copies of the arguments, every other member as that constructor sets or leaves it -- and the
// reference holds outputs of both modes (README.md:128-134, :148-157), so the restatement gives the constructor that meaning
// and says so here.  What stays undefined and is refused: a `.juncs` file together with these modes (readComponents reads the
// rebuilt graph's mAvgCoverage, which nothing ever sets).
struct TrxMap {
    Graph original;                      // the graph as read: its vertices are what virusBFB puts into the path
    std::vector<int> originalOf;         // [new segment id] -> original id (`originalSegs`, LGM.cpp:4286-4291 / :4386-4391); [0] unused
    std::vector<Junc> unusedSV;          // junctions of the original graph without a place in the rebuilt one (:4267-4270 / :4372-4375)
};
// LGM.cpp:4195-4295 / :4297-4395.  g is replaced by the rebuilt graph; the lines the reference prints go to log.
// false (+ err): the reference reads an empty vector / an unset variable there.
bool insertBeforeBFB(Graph& g, const std::vector<std::string>& insChr, TrxMap& map, std::vector<std::string>& log, std::string& err);
bool concatBeforeBFB(Graph& g, const std::vector<std::string>& conChr, TrxMap& map, std::vector<std::string>& log, std::string& err);
// LGM.cpp:3839-3939: path holds vertices of the rebuilt graph on entry and (mostly) of the original graph on return; a vertex the
// reference leaves untouched (no junction of the original graph leads to its segment) stays a vertex of the REBUILT graph:
// rebuiltVertex[i] tells which graph path[i] belongs to.  false (+ err): path->at(1) on a one-vertex path (the reference aborts).
bool virusBFB(const TrxMap& map, const Graph& rebuilt, std::vector<int>& path, std::vector<char>& rebuiltVertex,
              std::vector<std::string>& log, std::string& err);
// printBFB over such a mixed path
std::string formatPathMixed(const Graph& original, const Graph& rebuilt, const std::vector<int>& path, const std::vector<char>& rebuiltVertex);

struct OutJunc { int u, v; int count; };           // localhap.cpp:267-293
void synthesizeOutputJuncs(const std::vector<int>& path, std::vector<OutJunc>& out, bool increase);

// .sol token scan (localhap.cpp:192-212)
struct Sol { bool infeasible = false; double objective = 0; std::vector<std::pair<int, int>> cols; };
bool readSol(const std::string& path, Sol& sol);

// ---- whole `--op bfb` run (localhap.cpp:49-388) with the cbc call replaced by a given .sol ---
struct ChrStage {                                  // stage dumps for parity tests
    int startID = 0, endID = 0, bias = 0;
    bool shortcut = false, infeasible = false;
    std::vector<double> juncCN;
    std::vector<int> invSeg, invJunc;             // inversions map as sorted (segment, junction index) pairs
    std::vector<double> segCNAfterIndelBias;
    Dag dag;
    long numOrders = 0;
    std::vector<std::vector<int>> orders;          // kept only if keepOrders
    BfbResult bfb;
    std::vector<int> pathAfterIndel;
    bool indelPrinted = false;
};
struct RunOptions {
    std::string lh, juncs, lpPrefix = "oracle";
    std::vector<std::string> solPerChr;            // .sol path for each chromosome that reaches the ILP, in order
    bool juncInfo = false, reversed = false, all = false;
    bool keepOrders = false;
    size_t maxOrders = SIZE_MAX;
};
struct RunResult {
    bool ok = false; std::string err;
    std::vector<std::string> log;                  // stdout lines, in order
    std::vector<ChrStage> chr;
    std::vector<std::vector<int>> paths;
    std::vector<int> trxPath; bool trxRun = false; std::vector<std::string> trxTrace;
    bool trxBefore = false;                        // PROP I1 / C1: `paths` hold ORIGINAL segment ids (after virusBFB), `chr` the stages on the rebuilt graph
    std::vector<int> originalOf;                   // [rebuilt id] -> original id
    std::vector<OutJunc> outJuncs;
    std::vector<int> targetCN;
    int pathLen = 0, cnSum = 0, maxCN = 0, numInv = 0;
    double ilpError = 0;
    double reconSeconds = 0;   // wall time of stages #7,#8,#11-#16,#20 only (the region the GPU step covers)
};
RunResult runBfb(const RunOptions& opt);

// ---- whole `--op sc_bfb` run (localhap.cpp:390-679): several .lh files (single cells / sub-clones with the same
// segmentation) share one joint ILP per chromosome; solPerChr = the joint .sol of every chromosome that reaches the ILP.
struct ScResult {
    bool ok = false; std::string err;
    std::vector<std::string> log;                              // stdout lines, in order
    std::vector<std::vector<std::vector<int>>> paths;          // [graph][chromosome]
    std::vector<std::vector<int>> trxPaths;                    // [graph], when PROP asks for BFB-TRX
    std::vector<std::vector<ChrStage>> chr;                    // [chromosome][graph] stage dumps of reconstructed units
    int pathLen = 0, cnSum = 0, maxCN = 0, nSeg = 0, nJunc = 0;
};
ScResult runScBfb(const std::vector<std::string>& lhs, const RunOptions& opt);

// ---- ILP rows (LGM.cpp:4397-4752), semantic form ------------------------------------------
struct IlpModel {
    int numCols = 0;
    std::vector<int64_t> rowPtr; std::vector<int> colIdx; std::vector<double> val;
    std::vector<double> rowLo, rowUp, colLo, colUp, obj;
    int numInt = 0;
};
void buildBfbIlp(const Graph& g, int startID, int endID, const std::vector<double>& juncCN,
                 const std::vector<std::vector<int>>& components, bool juncsInfo, int bias, IlpModel& m,
                 bool literalHotLoop = false);
// LGM.cpp:4754-5093 (`--op sc_bfb`): the joint model of several graphs over one chromosome, literal restatement
void buildBfbIlpSc(const std::vector<const Graph*>& graphs, int startID, int endID, const std::vector<std::vector<double>>& juncCNs,
                   const std::vector<std::vector<int>>& evolution, IlpModel& m);


}  // namespace oracle
