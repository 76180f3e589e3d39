// oracle/oracle_capi.cpp -- TEST INFRASTRUCTURE ONLY.
// extern "C" surface of the CPU oracle for ctypes (tests/, smoke(), bench.py cpu_baseline leg).
// Results are returned as one JSON document so the Python side needs no struct mirroring.
#include <chrono>
#include <cstring>
#include <sstream>
#include <string>

#include "bfb_oracle.hpp"

using namespace oracle;

namespace {
void jstr(std::ostringstream& o, const std::string& s) {
    o << '"';
    for (char c : s) {
        if (c == '"' || c == '\\') o << '\\' << c;
        else if (c == '\n') o << "\\n";
        else o << c;
    }
    o << '"';
}
template <class T> void jarr(std::ostringstream& o, const std::vector<T>& v) {
    o << '[';
    for (size_t i = 0; i < v.size(); i++) { if (i) o << ','; o << v[i]; }
    o << ']';
}
void jarr2(std::ostringstream& o, const std::vector<std::vector<int>>& v) {
    o << '[';
    for (size_t i = 0; i < v.size(); i++) { if (i) o << ','; jarr(o, v[i]); }
    o << ']';
}
void jdarr(std::ostringstream& o, const std::vector<double>& v) {
    o << '[';
    o.precision(17);
    for (size_t i = 0; i < v.size(); i++) { if (i) o << ','; o << v[i]; }
    o << ']';
}
char* dup(const std::string& s) { char* p = (char*)malloc(s.size() + 1); memcpy(p, s.c_str(), s.size() + 1); return p; }

std::vector<std::string> split(const char* s, char d) {
    std::vector<std::string> out; if (!s || !*s) return out;
    std::string cur;
    for (const char* p = s; *p; p++) { if (*p == d) { out.push_back(cur); cur.clear(); } else cur += *p; }
    out.push_back(cur);
    return out;
}
}  // namespace

extern "C" {

void oracle_free(char* p) { free(p); }

// Runs the whole `--op bfb` flow (localhap.cpp:49-388) with the cbc step replaced by the given .sol files
// (comma separated, one per chromosome that reaches the ILP).  flags: bit0 reversed, bit1 all, bit2 junc_info,
// bit3 keep orders in the dump.  Returns a malloc'ed JSON string (free with oracle_free).
char* oracle_run_bfb(const char* lh, const char* juncs, const char* sols, int flags, long long maxOrders, double* seconds) {
    RunOptions opt;
    opt.lh = lh; opt.juncs = juncs ? juncs : "";
    opt.solPerChr = split(sols, ',');
    opt.reversed = flags & 1; opt.all = flags & 2; opt.juncInfo = flags & 4; opt.keepOrders = flags & 8;
    if (maxOrders > 0) opt.maxOrders = (size_t)maxOrders;
    auto t0 = std::chrono::steady_clock::now();
    RunResult R;
    try { R = runBfb(opt); }
    catch (const std::exception& e) {   // (the reference would terminate here: std::stoi on a malformed token, say)
        R = RunResult(); R.ok = false; R.err = std::string("exception: ") + e.what();
    }
    auto t1 = std::chrono::steady_clock::now();
    if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
    std::ostringstream o;
    o << "{\"ok\":" << (R.ok ? "true" : "false") << ",\"err\":"; jstr(o, R.err);
    o << ",\"log\":[";
    for (size_t i = 0; i < R.log.size(); i++) { if (i) o << ','; jstr(o, R.log[i]); }
    o << "],\"paths\":"; jarr2(o, R.paths);
    o << ",\"trx_run\":" << (R.trxRun ? "true" : "false") << ",\"trx_path\":"; jarr(o, R.trxPath);
    o << ",\"trx_trace\":["; for (size_t i = 0; i < R.trxTrace.size(); i++) { if (i) o << ','; jstr(o, R.trxTrace[i]); } o << "]";
    o << ",\"trx_before\":" << (R.trxBefore ? "true" : "false") << ",\"original_of\":"; jarr(o, R.originalOf);
    o << ",\"target_cn\":"; jarr(o, R.targetCN);
    o << ",\"recon_seconds\":" << R.reconSeconds << ",\"path_len\":" << R.pathLen << ",\"cn_sum\":" << R.cnSum << ",\"max_cn\":" << R.maxCN << ",\"num_inv\":" << R.numInv;
    o << ",\"out_juncs\":[";
    for (size_t i = 0; i < R.outJuncs.size(); i++) { if (i) o << ','; o << '[' << R.outJuncs[i].u << ',' << R.outJuncs[i].v << ',' << R.outJuncs[i].count << ']'; }
    o << "],\"chr\":[";
    for (size_t c = 0; c < R.chr.size(); c++) {
        const ChrStage& s = R.chr[c];
        if (c) o << ',';
        o << "{\"start\":" << s.startID << ",\"end\":" << s.endID << ",\"bias\":" << s.bias
          << ",\"shortcut\":" << (s.shortcut ? "true" : "false") << ",\"infeasible\":" << (s.infeasible ? "true" : "false");
        o << ",\"junc_cn\":"; jdarr(o, s.juncCN);
        o << ",\"inv_seg\":"; jarr(o, s.invSeg);
        o << ",\"inv_junc\":"; jarr(o, s.invJunc);
        o << ",\"seg_cn\":"; jdarr(o, s.segCNAfterIndelBias);
        o << ",\"node2pat\":"; jarr2(o, s.dag.node2pat);
        o << ",\"node2loop\":"; jarr2(o, s.dag.node2loop);
        o << ",\"adj\":"; jarr2(o, s.dag.adj);
        o << ",\"num_orders\":" << s.numOrders;
        o << ",\"orders\":"; jarr2(o, s.orders);
        o << ",\"first_valid\":" << s.bfb.firstValidOrder << ",\"first_forward\":" << s.bfb.firstValidOrientationForward
          << ",\"evaluated\":" << s.bfb.evaluated << ",\"ub\":" << (s.bfb.undefinedBehaviour ? "true" : "false")
          << ",\"ub_valid\":" << (s.bfb.undefinedOnValid ? "true" : "false");
        o << ",\"bkp\":"; jarr(o, s.bfb.bkpFirst);
        o << ",\"path\":"; jarr(o, s.bfb.path);
        o << ",\"all_paths\":"; jarr2(o, s.bfb.allPaths);
        o << ",\"all_eval_idx\":"; jarr(o, s.bfb.allEvalIdx);
        o << ",\"indel_printed\":" << (s.indelPrinted ? "true" : "false");
        o << ",\"path_indel\":"; jarr(o, s.pathAfterIndel);
        o << "}";
    }
    o << "]}";
    return dup(o.str());
}

// ILP model of chromosome `chr` exactly as main() would hand it to BFB_ILP (localhap.cpp:111-173): graph loaded,
// partitions set, .juncs read, getJuncCN / bias / getIndelBias applied for chromosomes 0..chr.  literal != 0 runs the
// reference's O(numPat^2) coefficient loop (LGM.cpp:4464-4477) instead of its closed form.  JSON: CSR + bounds.
char* oracle_ilp_json(const char* lh, const char* juncs, int chr, int junc_info, int literal, double* seconds) {
    Graph g; std::string err;
    std::ostringstream o;
    o.precision(17);
    bool ok = readGraph(lh, g, err) && calculateHapDepth(g, err);
    if (!ok || chr < 0 || chr >= (int)g.sinkIds.size()) { o << "{\"ok\":false}"; return dup(o.str()); }
    calculateCopyNum(g);
    for (size_t i = 0; i < g.sourceIds.size(); i++)
        for (int j = g.sourceIds[i]; j <= g.sinkIds[i]; j++) g.segs[j - 1].partition = (int)i;
    std::vector<std::vector<int>> components;
    std::vector<std::string> log;
    readComponents(g, juncs ? juncs : "", components, log);
    std::vector<double> juncCN; int bias = 1;
    for (int c = 0; c <= chr; c++) {
        Inversions inv;
        getJuncCN(g, g.sourceIds[c], g.sinkIds[c], inv, juncCN);
        bias = computeBias(g, g.sourceIds[c], g.sinkIds[c], inv, juncCN);
        getIndelBias(g, g.sourceIds[c], g.sinkIds[c]);
    }
    std::vector<std::vector<int>> valid;
    for (auto& comp : components) if (g.segById(comp[0])->partition == chr) valid.push_back(comp);
    IlpModel m;
    auto t0 = std::chrono::steady_clock::now();
    buildBfbIlp(g, g.sourceIds[chr], g.sinkIds[chr], juncCN, valid, junc_info != 0, bias, m, literal != 0);
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    o << "{\"ok\":true,\"n_cols\":" << m.numCols << ",\"n_int\":" << m.numInt << ",\"bias\":" << bias << ",\"row_ptr\":";
    { std::vector<long long> rp(m.rowPtr.begin(), m.rowPtr.end()); jarr(o, rp); }
    o << ",\"col\":"; jarr(o, m.colIdx);
    o << ",\"val\":"; jdarr(o, m.val);
    o << ",\"row_lo\":"; jdarr(o, m.rowLo); o << ",\"row_up\":"; jdarr(o, m.rowUp);
    o << ",\"col_lo\":"; jdarr(o, m.colLo); o << ",\"col_up\":"; jdarr(o, m.colUp);
    o << ",\"obj\":"; jdarr(o, m.obj);
    o << "}";
    return dup(o.str());
}

// translocationBFB (LGM.cpp:4052-4193) alone on given per-chromosome paths: `paths` = chromosomes separated by ';', signed
// segment ids separated by ','.  The graph's PROP line names the main chromosome.  JSON: result path, printed line, trace,
// and the per-chromosome paths as the call left them (they may have been reverse-complemented in place).
char* oracle_translocation_json(const char* lh, const char* paths) {
    Graph g; std::string err;
    std::ostringstream o;
    bool ok = readGraph(lh, g, err) && calculateHapDepth(g, err);
    if (!ok) { o << "{\"ok\":false}"; return dup(o.str()); }
    calculateCopyNum(g);
    Props props;
    readBFBProps(lh, props);
    std::vector<std::vector<int>> pp;
    for (auto& chr : split(paths, ';')) {
        pp.emplace_back();
        for (auto& t : split(chr.c_str(), ',')) if (!t.empty()) pp.back().push_back(atoi(t.c_str()));
    }
    std::vector<int> res; std::vector<std::string> log, trace;
    translocationBFB(g, pp, res, props.mainChr, log, &trace);
    o << "{\"ok\":true,\"path\":"; jarr(o, res);
    o << ",\"line\":"; jstr(o, log.empty() ? std::string() : log.back());
    o << ",\"paths\":"; jarr2(o, pp);
    o << ",\"trace\":["; for (size_t i = 0; i < trace.size(); i++) { if (i) o << ','; jstr(o, trace[i]); } o << "]}";
    return dup(o.str());
}

// `--op sc_bfb` (localhap.cpp:390-679): lhs and sols comma separated; flags as oracle_run_bfb.
char* oracle_run_sc_bfb(const char* lhs, const char* sols, int flags, long long maxOrders) {
    RunOptions opt;
    opt.solPerChr = split(sols, ',');
    opt.reversed = flags & 1; opt.all = flags & 2;
    if (maxOrders > 0) opt.maxOrders = (size_t)maxOrders;
    ScResult R = runScBfb(split(lhs, ','), opt);
    std::ostringstream o;
    o << "{\"ok\":" << (R.ok ? "true" : "false") << ",\"err\":"; jstr(o, R.err);
    o << ",\"log\":[";
    for (size_t i = 0; i < R.log.size(); i++) { if (i) o << ','; jstr(o, R.log[i]); }
    o << "],\"paths\":[";
    for (size_t k = 0; k < R.paths.size(); k++) { if (k) o << ','; jarr2(o, R.paths[k]); }
    o << "],\"trx_paths\":"; jarr2(o, R.trxPaths);
    o << ",\"path_len\":" << R.pathLen << ",\"cn_sum\":" << R.cnSum << ",\"max_cn\":" << R.maxCN << ",\"n_seg\":" << R.nSeg << ",\"n_junc\":" << R.nJunc;
    o << ",\"chr\":[";
    for (size_t c = 0; c < R.chr.size(); c++) {
        if (c) o << ',';
        o << '[';
        for (size_t k = 0; k < R.chr[c].size(); k++) {
            const ChrStage& s = R.chr[c][k];
            if (k) o << ',';
            o << "{\"shortcut\":" << (s.shortcut ? "true" : "false") << ",\"infeasible\":" << (s.infeasible ? "true" : "false")
              << ",\"num_orders\":" << s.numOrders << ",\"first_valid\":" << s.bfb.firstValidOrder << ",\"first_forward\":" << s.bfb.firstValidOrientationForward
              << ",\"evaluated\":" << s.bfb.evaluated << ",\"ub\":" << (s.bfb.undefinedBehaviour ? "true" : "false");
            o << ",\"node2pat\":"; jarr2(o, s.dag.node2pat);
            o << ",\"node2loop\":"; jarr2(o, s.dag.node2loop);
            o << ",\"bkp\":"; jarr(o, s.bfb.bkpFirst);
            o << ",\"path\":"; jarr(o, s.bfb.path);
            o << ",\"path_indel\":"; jarr(o, s.pathAfterIndel);
            o << ",\"indel_printed\":" << (s.indelPrinted ? "true" : "false") << "}";
        }
        o << ']';
    }
    o << "]}";
    return dup(o.str());
}

// Joint ILP of chromosome `chr` as main() hands it to BFB_ILP_SC (localhap.cpp:464-515): every graph loaded, the first
// graph's getIndelBias applied for chromosomes 0..chr.
char* oracle_ilp_sc_json(const char* lhs, int chr) {
    std::vector<std::string> files = split(lhs, ',');
    std::vector<Graph> graphs(files.size());
    std::ostringstream o;
    o.precision(17);
    std::string err;
    for (size_t k = 0; k < files.size(); k++) {
        if (!readGraph(files[k], graphs[k], err) || !calculateHapDepth(graphs[k], err)) { o << "{\"ok\":false}"; return dup(o.str()); }
        calculateCopyNum(graphs[k]);
    }
    Graph& g = graphs[0];
    calculateHapDepth(g, err); calculateCopyNum(g);
    if (chr < 0 || chr >= (int)g.sinkIds.size()) { o << "{\"ok\":false}"; return dup(o.str()); }
    for (int c = 0; c <= chr; c++) getIndelBias(g, g.sourceIds[c], g.sinkIds[c]);
    std::vector<const Graph*> gp; std::vector<std::vector<double>> jcn(graphs.size());
    for (size_t k = 0; k < graphs.size(); k++) { gp.push_back(&graphs[k]); Inversions inv; getJuncCN(graphs[k], g.sourceIds[chr], g.sinkIds[chr], inv, jcn[k]); }
    std::vector<std::vector<int>> evolution(graphs.size());
    for (size_t i = 0; i < graphs.size(); i++) for (size_t j = i + 1; j < graphs.size(); j++) evolution[i].push_back((int)j);
    IlpModel m;
    buildBfbIlpSc(gp, g.sourceIds[chr], g.sinkIds[chr], jcn, evolution, m);
    o << "{\"ok\":true,\"n_cols\":" << m.numCols << ",\"n_int\":" << m.numInt << ",\"row_ptr\":";
    { std::vector<long long> rp(m.rowPtr.begin(), m.rowPtr.end()); jarr(o, rp); }
    o << ",\"col\":"; jarr(o, m.colIdx);
    o << ",\"val\":"; jdarr(o, m.val);
    o << ",\"row_lo\":"; jdarr(o, m.rowLo); o << ",\"row_up\":"; jdarr(o, m.rowUp);
    o << ",\"col_lo\":"; jdarr(o, m.colLo); o << ",\"col_up\":"; jdarr(o, m.colUp);
    o << ",\"obj\":"; jdarr(o, m.obj);
    o << "}";
    return dup(o.str());
}

// Parsed-graph dump (after calculateHapDepth/calculateCopyNum), same JSON shape as oracle/_ref's ref_graph_dump.
static char* graph_dump_impl(const char* lh, const char* juncs);
char* oracle_graph_dump(const char* lh) { return graph_dump_impl(lh, nullptr); }
// the same after readComponents(juncs) (LGM.cpp:5096-5156); the components come as an extra "components" member
char* oracle_graph_dump_juncs(const char* lh, const char* juncs) { return graph_dump_impl(lh, juncs); }
static char* graph_dump_impl(const char* lh, const char* juncs) {
    Graph g; std::string err;
    std::ostringstream o;
    o.precision(17);
    bool ok = readGraph(lh, g, err) && calculateHapDepth(g, err);
    if (ok) calculateCopyNum(g);
    std::vector<std::vector<int>> components;
    if (ok && juncs) {
        for (size_t i = 0; i < g.sourceIds.size(); i++)
            for (int j = g.sourceIds[i]; j <= g.sinkIds[i]; j++) g.segs[j - 1].partition = (int)i;
        std::vector<std::string> clog;
        readComponents(g, juncs, components, clog);
    }
    o << "{\"ok\":" << (ok ? "true" : "false") << ",\"err\":"; jstr(o, err);
    o << ",\"segs\":[";
    for (size_t i = 0; i < g.segs.size(); i++) {
        auto& s = g.segs[i];
        if (i) o << ',';
        o << '[' << s.id << ',' << s.chrId << ','; jstr(o, s.chrom); o << ',' << s.start << ',' << s.end << ',' << s.cov << ',' << s.cn << ']';
    }
    o << "],\"juncs\":[";
    for (size_t i = 0; i < g.juncs.size(); i++) {
        auto& j = g.juncs[i];
        if (i) o << ',';
        o << '[' << j.src << ",\"" << j.sdir << "\"," << j.tgt << ",\"" << j.tdir << "\"," << j.cov << ',' << j.cn << ','
          << (j.inferred ? 1 : 0) << ',' << (j.bounded ? 1 : 0) << ']';
    }
    o << "],\"sources\":"; jarr(o, g.sourceIds);
    o << ",\"sinks\":"; jarr(o, g.sinkIds);
    if (juncs) { o << ",\"components\":"; jarr2(o, components); }
    o << ",\"log\":[";
    for (size_t i = 0; i < g.log.size(); i++) { if (i) o << ','; jstr(o, g.log[i]); }
    o << "]}";
    return dup(o.str());
}

}  // extern "C"
