// ambigram_cli.cpp -- drop-in for `Ambigram --op bfb` (localhap.cpp:19-388) on top of the C ABI of the HIP engine.
//
// Same flags (localhap.cpp:22-40), same stdout lines in the same order, same side files (<prefix>.lp, <prefix>.sol via
// the external `cbc`, appended simulation_sv.txt and time.csv), same exit codes for unreadable .lh / missing .sol.
// The per-chromosome stages run on the GPU through libambigram_hip.so; the ILP model is built on the host
// (ambi_ilp_build) and solved by whatever `cbc` is on PATH, exactly as the reference shells out (localhap.cpp:179-181).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <sys/wait.h>
#include <string>
#include <vector>

#include "../../include/ambigram_hip.h"

namespace {

struct Args { std::map<std::string, std::string> kv; bool help = false; };

Args parse(int argc, char** argv) {
    Args a;
    for (int i = 1; i < argc; i++) {
        std::string t = argv[i];
        if (t == "--help") { a.help = true; continue; }
        if (t.rfind("--", 0) != 0) continue;
        std::string k = t.substr(2), v;
        size_t eq = k.find('=');
        if (eq != std::string::npos) { v = k.substr(eq + 1); k = k.substr(0, eq); }
        else if (i + 1 < argc && std::string(argv[i + 1]).rfind("--", 0) != 0) v = argv[++i];
        else v = "true";
        a.kv[k] = v;
    }
    return a;
}
bool truthy(const std::string& s) { return s == "true" || s == "1" || s == "True"; }

void print_log(const ambi_graph_t* g, size_t* printed) {
    int64_t n = ambi_graph_log(g, nullptr, 0);
    std::string buf((size_t)n + 1, '\0');
    ambi_graph_log(g, &buf[0], n + 1);
    buf.resize((size_t)n);
    std::cout << buf.substr(*printed);
    *printed = buf.size();
}

std::string path_text(const ambi_graph_t* g, const std::vector<int32_t>& p) {
    int64_t n = ambi_format_path(g, p.data(), (int32_t)p.size(), nullptr, 0);
    std::string s((size_t)n + 1, '\0');
    ambi_format_path(g, p.data(), (int32_t)p.size(), &s[0], n + 1);
    s.resize((size_t)n);
    return s;
}

struct OutJ { int u, v; double cn; };
void merge_steps(std::vector<OutJ>& acc, int u, int v, double c, bool increase) {
    for (auto& j : acc)
        if ((j.u == u && j.v == v) || (j.u == -v && j.v == -u)) { if (increase) j.cn += c; return; }
    acc.push_back({u, v, c});
}

int die(const std::string& msg) { std::cerr << msg << std::endl; return 1; }

// `--op sc_bfb` (localhap.cpp:390-679): --in_lh is a comma-separated list of .lh files over the same segmentation; one
// joint ILP per chromosome (BFB_ILP_SC), the block k*numComp..(k+1)*numComp of its solution belongs to graph k.  All
// graphs x chromosomes are ONE reconstruct batch.  --juncdb / --junc_info are accepted and ignored, as in the reference.
int run_sc_bfb(Args& A) {
    auto t_begin = std::chrono::steady_clock::now();
    const std::string lh_list = A.kv["in_lh"], prefix = A.kv["lp_prefix"];
    const bool reversed = truthy(A.kv["reversed"]), all = truthy(A.kv["all"]);
    std::vector<std::string> files;
    { std::stringstream ss(lh_list); std::string t; while (std::getline(ss, t, ',')) if (!t.empty()) files.push_back(t); }   // strtok_r(.., ","): empty fields skipped
    if (files.empty()) return die("Cannot open file ");
    const int G = (int)files.size();
    std::vector<ambi_graph_t*> gs(G, nullptr);
    int rc;
    for (int k = 0; k < G; k++) {
        rc = ambi_graph_read_lh(files[k].c_str(), &gs[k]);
        if (rc == AMBI_ERR_OPEN) return die("Cannot open file " + files[k]);
        if (rc != 0) return die(std::string("input error: ") + ambi_error_string(rc));
        size_t printed = 0;
        print_log(gs[k], &printed);
    }
    {   // localhap.cpp:438-439: calculateHapDepth / calculateCopyNum a second time on the first graph
        int64_t before = ambi_graph_log(gs[0], nullptr, 0);
        if ((rc = ambi_graph_recalculate(gs[0])) != 0) return die(ambi_error_string(rc));
        size_t printed = (size_t)before;
        print_log(gs[0], &printed);
    }
    ambi_graph_t* g0 = gs[0];
    int32_t n_seg, n_junc, n_chr, ins_mode, con_mode;
    char main_chr[256];
    ambi_graph_sizes(g0, &n_seg, &n_junc, &n_chr);
    ambi_graph_props(g0, &ins_mode, &con_mode, main_chr, sizeof(main_chr));   // PROP of the first file (lhRawFn is cut at the first comma)
    for (int k = 1; k < G; k++) {
        int32_t ns, nj, nc;
        ambi_graph_sizes(gs[k], &ns, &nj, &nc);
        if (ns != n_seg || nc != n_chr) return die("sc_bfb: the graphs do not share one segmentation");
    }
    // probe batch: every graph x chromosome -- fold-back CNs of every graph (BFB_ILP_SC calls getJuncCN per graph,
    // LGM.cpp:4794), the first graph's getIndelBias (localhap.cpp:497) and its shortcut decision (:505)
    ambi_batch_t* probe; ambi_batch_create(&probe);
    for (int c = 0; c < n_chr; c++) for (int k = 0; k < G; k++)
        if ((rc = ambi_batch_add_chromosome(probe, gs[k], c, 0, nullptr, nullptr, 0)) < 0) return die(ambi_error_string(rc));
    if ((rc = ambi_batch_upload(probe)) != 0 || (rc = ambi_batch_run(probe, 0, nullptr)) != 0 || (rc = ambi_batch_download(probe)) != 0)
        return die(std::string("engine: ") + ambi_error_string(rc));
    std::vector<std::vector<double>> cn(G, std::vector<double>(n_seg));
    for (int k = 0; k < G; k++) ambi_graph_segments(gs[k], nullptr, nullptr, nullptr, nullptr, nullptr, cn[k].data());
    ambi_batch_t* b; ambi_batch_create(&b);
    std::vector<std::string> head(n_chr);
    std::vector<int> first_unit(n_chr, -1);   // unit of (chromosome c, graph 0) in the reconstruct batch; graph k follows at +k
    std::vector<char> plain(n_chr, 0);
    const double solver_timeout = A.kv.count("solver_timeout") ? atof(A.kv["solver_timeout"].c_str()) : 0;
    int units = 0;
    for (int c = 0; c < n_chr; c++) {
        int32_t s, e;
        ambi_graph_chromosome(g0, c, &s, &e);
        const int n = e - s + 1;
        std::vector<double> seg(G * (size_t)n), fold(G * (size_t)n), junc_cn(2 * (n + 1)), seg_cn(n + 1);
        ambi_unit_result_t pr0{};
        for (int k = 0; k < G; k++) {
            ambi_unit_result_t pr; ambi_batch_unit_result(probe, c * G + k, &pr);
            if (k == 0) pr0 = pr;
            ambi_batch_unit_prepare(probe, c * G + k, junc_cn.data(), seg_cn.data(), nullptr, nullptr);
            if (k == 0) for (int i = 1; i <= n; i++) cn[0][s - 1 + i - 1] = seg_cn[i];   // getIndelBias edits the first graph only
            for (int i = 0; i < n; i++) { seg[(size_t)k * n + i] = cn[k][s - 1 + i]; fold[(size_t)k * n + i] = junc_cn[2 * (i + 1) + 1]; }
        }
        if (pr0.status == AMBI_ST_SHORTCUT) { plain[c] = 1; continue; }   // no fold-back in the first graph: reference paths, nothing printed (:505-512)
        ambi_ilp_t* ilp = nullptr;
        if ((rc = ambi_ilp_build_sc(g0, c, G, seg.data(), fold.data(), &ilp)) != 0) return die(ambi_error_string(rc));
        head[c] = "Declare done\n";
        for (int k = 0; k < G; k++) head[c] += "ILP formula done\n";
        head[c] += "Variable constrains done\n";
        ambi_ilp_write_mps(ilp, (prefix + ".mps").c_str());
        ambi_ilp_write_lp(ilp, (prefix + ".lp").c_str());
        ambi_ilp_destroy(ilp);
        (void)remove(("./" + prefix + ".sol").c_str());
        std::string cmd = "cbc " + prefix + ".lp solve solu " + prefix + ".sol";   // localhap.cpp:527-529
        if (solver_timeout > 0) { char t[64]; snprintf(t, sizeof(t), "timeout -k 5 %.0f ", solver_timeout); cmd = t + cmd; }
        std::cout.flush();
        const int solver_rc = system(cmd.c_str());
        if (solver_rc != 0) std::cerr << "ILP warning: `" << cmd << "` ended with status " << (WIFEXITED(solver_rc) ? WEXITSTATUS(solver_rc) : -1) << std::endl;
        first_unit[c] = units;
        for (int k = 0; k < G; k++) {
            rc = ambi_batch_add_chromosome_sol_block(b, gs[k], c, ("./" + prefix + ".sol").c_str(), k, G);
            if (rc == AMBI_ERR_SOL_OPEN) return die("ILP error: cannot open file ./" + prefix + ".sol");   // localhap.cpp:533-536
            if (rc < 0) return die(ambi_error_string(rc));
            units++;
        }
    }
    ambi_batch_destroy(probe);
    if (units > 0 && ((rc = ambi_batch_upload(b)) != 0 ||
                      (rc = ambi_batch_run(b, (reversed ? AMBI_FLAG_REVERSED : 0u) | (all ? AMBI_FLAG_ALL : 0u), nullptr)) != 0 ||
                      (rc = ambi_batch_download(b)) != 0))
        return die(std::string("engine: ") + ambi_error_string(rc));
    std::vector<std::vector<std::vector<int32_t>>> paths(G, std::vector<std::vector<int32_t>>(n_chr));
    int refused = 0;
    for (int c = 0; c < n_chr; c++) {
        int32_t s, e;
        ambi_graph_chromosome(g0, c, &s, &e);
        auto reference_path = [&](int k) { paths[k][c].clear(); for (int i = s; i <= e; i++) paths[k][c].push_back(i); };
        if (plain[c]) { for (int k = 0; k < G; k++) reference_path(k); continue; }
        std::cout << head[c];
        ambi_unit_result_t r0; ambi_batch_unit_result(b, first_unit[c], &r0);
        if (r0.status == AMBI_ST_INFEASIBLE) {   // localhap.cpp:543-551
            std::cout << "ILP is unsolvable.\n";
            for (int k = 0; k < G; k++) reference_path(k);
            continue;
        }
        for (int k = 0; k < G; k++) {
            const int u = first_unit[c] + k;
            ambi_unit_result_t r; ambi_batch_unit_result(b, u, &r);
            if (r.status != AMBI_ST_OK) {
                std::cout.flush();
                std::cerr << "sc_bfb: graph " << k << " chromosome " << c << ": " << ambi_error_string(r.status) << std::endl;
                refused++;
                continue;
            }
            std::vector<int32_t> p(r.path_len), q(r.path_indel_len);
            ambi_batch_unit_path(b, u, 0, p.data(), r.path_len);
            ambi_batch_unit_path(b, u, 1, q.data(), r.path_indel_len);
            if (all) {
                const int64_t stride = 2ll * r.path_len + 64;
                for (int pass = 0; pass < 2; pass++) {
                    int64_t nv = 0;
                    ambi_batch_all_count(b, u, pass, &nv);
                    for (int64_t lo = 0; lo < nv; lo += 64) {
                        const int64_t cnt = nv - lo < 64 ? nv - lo : 64;
                        std::vector<int32_t> len((size_t)cnt), cells((size_t)(cnt * stride));
                        if ((rc = ambi_batch_all_paths(b, u, pass, lo, cnt, len.data(), cells.data(), stride)) != 0) return die(std::string("sc_bfb --all: ") + ambi_error_string(rc));
                        for (int64_t j = 0; j < cnt; j++) {
                            if (len[j] < 0) return die(std::string("sc_bfb --all: ") + ambi_error_string(len[j]));
                            std::vector<int32_t> pj(cells.begin() + j * stride, cells.begin() + j * stride + len[j]);
                            std::cout << path_text(gs[k], pj) << std::endl;
                        }
                    }
                }
            } else
                std::cout << path_text(gs[k], p) << std::endl;
            if (r.indel_printed) std::cout << "BFB path with insertion, deletion, or duplication:\n" << path_text(gs[k], q) << std::endl;
            paths[k][c] = q;
        }
    }
    ambi_batch_destroy(b);
    if (refused) return die("sc_bfb: " + std::to_string(refused) + " unit(s) without a path");
    int path_len = 0, cn_sum = 0, max_cn_i = 0;
    for (int k = 0; k < G; k++) {   // localhap.cpp:654-660
        for (auto& p : paths[k]) path_len += (int)p.size();
        for (double v : cn[k]) { cn_sum += v; max_cn_i = (max_cn_i > v) ? max_cn_i : v; }
    }
    if (ins_mode == 2 || con_mode == 2) {   // :661-664: every graph
        if (!main_chr[0]) return die("BFB-TRX needs PROP M:<chr>");
        for (int k = 0; k < G; k++) {
            std::cout << "BFB with translocation:\n";
            std::vector<int64_t> offs(n_chr + 1, 0);
            for (int c = 0; c < n_chr; c++) offs[c + 1] = offs[c] + (int64_t)paths[k][c].size();
            std::vector<int32_t> flat((size_t)offs[n_chr] + 1), out((size_t)offs[n_chr] * 2 + 16);
            for (int c = 0; c < n_chr; c++) std::copy(paths[k][c].begin(), paths[k][c].end(), flat.begin() + offs[c]);
            int len = ambi_translocation_bfb(gs[k], flat.data(), offs.data(), n_chr, out.data(), (int32_t)out.size());
            if (len < 0) return die(ambi_error_string(len));
            out.resize(len);
            std::cout << path_text(gs[k], out) << std::endl;
        }
    }
    {   // time.csv (localhap.cpp:666-678): name of the first file up to its first '.', first graph's sizes, sums over all graphs
        auto t_end = std::chrono::steady_clock::now();
        std::ofstream tf("time.csv", std::ios_base::app);
        tf << files[0].substr(0, files[0].find(".")) << "," << n_seg << "," << 0 << "," << n_junc << "," << cn_sum << "," << path_len << ","
           << max_cn_i << "," << std::chrono::duration_cast<std::chrono::microseconds>(t_end - t_begin).count() / 1000000.0 << "\n";
    }
    for (auto* g : gs) ambi_graph_destroy(g);
    return 0;
}


// `--op bfb` (localhap.cpp:49-388).  One sample = the reference's behaviour, line for line.  Extensions (no counterpart in the
// reference, whose --in_lh names one file for this op): --in_lh may be a comma-separated LIST of samples -- every sample goes
// through its own ILP / solver stage in turn, then the chromosomes of ALL samples are reconstructed as ONE batch -- and
// --devices N|all deals that batch over several GPUs (ambi_batch_run_sharded: one host thread per device, round-robin).
// Every sample's stdout lines and side-file rows are what a run on that sample alone prints, in sample order.
struct Sample {
    std::string lh;
    ambi_graph_t* g = nullptr;
    int32_t n_seg = 0, n_junc = 0, n_chr = 0, ins_mode = 0, con_mode = 0;
    char main_chr[256] = {0};
    std::vector<double> cn_all;
    std::vector<int32_t> seg_start, seg_end;
    std::vector<std::string> head;       // lines the reference prints before a chromosome's path lines
    std::vector<int> unit;               // chromosome -> unit of the reconstruct batch
    int num_inv = 0;
    bool trx_before = false;             // PROP I1 / C1: g is the rebuilt graph, g_file the graph of the file (chromosome names / positions of restored paths)
    ambi_graph_t* g_file = nullptr;
    std::chrono::steady_clock::time_point t_begin;
};

int run_bfb(Args& A) {
    const std::string prefix = A.kv["lp_prefix"], juncs = A.kv.count("juncdb") ? A.kv["juncdb"] : "";
    const bool junc_info = truthy(A.kv["junc_info"]), reversed = truthy(A.kv["reversed"]), all = truthy(A.kv["all"]);
    const double solver_timeout = A.kv.count("solver_timeout") ? atof(A.kv["solver_timeout"].c_str()) : 0;   // extension: seconds, 0 = none
    int n_devices = 0;                   // 0: one device, the classic path; > 0: that many; -1: all visible
    std::vector<int32_t> device_list;    // or an explicit list of ordinals "0,1,1" (an ordinal may repeat: several shares on one device)
    if (A.kv.count("devices")) {
        const std::string d = A.kv["devices"];
        if (d.find(',') != std::string::npos) {
            std::stringstream ss(d);
            std::string t;
            while (std::getline(ss, t, ',')) if (!t.empty()) device_list.push_back(atoi(t.c_str()));
            n_devices = (int)device_list.size();
        } else n_devices = (d == "all") ? -1 : atoi(d.c_str());
    }
    std::vector<std::string> files;
    {
        std::stringstream ss(A.kv["in_lh"]);
        std::string f;
        while (std::getline(ss, f, ',')) if (!f.empty()) files.push_back(f);
        if (files.empty()) files.push_back(A.kv["in_lh"]);
    }
    int rc;
    std::vector<Sample> samples(files.size());
    ambi_batch_t* b; ambi_batch_create(&b);
    int n_units = 0;
    for (size_t si = 0; si < files.size(); si++) {
        Sample& S = samples[si];
        S.lh = files[si];
        S.t_begin = std::chrono::steady_clock::now();
        rc = ambi_graph_read_lh(S.lh.c_str(), &S.g);
        if (rc == AMBI_ERR_OPEN) return die("Cannot open file " + S.lh);            // Graph.cpp:111-114
        if (rc != 0) return die(std::string("input error: ") + ambi_error_string(rc));
        ambi_graph_t* g = S.g;
        size_t printed = 0;
        // PROP I1 / C1 (TRX-BFB, localhap.cpp:79-88): the graph has been rebuilt while loading; the reference leaves it in ./new.lh
        S.trx_before = ambi_graph_trx_before(g, nullptr, 0) > 0;
        if (S.trx_before) { (void)ambi_graph_write_lh(g, "./new.lh"); if ((rc = ambi_graph_trx_original(g, &S.g_file)) != 0) return die(ambi_error_string(rc)); }
        print_log(g, &printed);
        if (!juncs.empty()) { if ((rc = ambi_graph_read_juncs(g, juncs.c_str())) != 0) return die(ambi_error_string(rc)); print_log(g, &printed); }
        ambi_graph_sizes(g, &S.n_seg, &S.n_junc, &S.n_chr);
        ambi_graph_props(g, &S.ins_mode, &S.con_mode, S.main_chr, sizeof(S.main_chr));
        S.cn_all.resize(S.n_seg); S.seg_start.resize(S.n_seg); S.seg_end.resize(S.n_seg);
        ambi_graph_segments(g, nullptr, nullptr, S.seg_start.data(), S.seg_end.data(), nullptr, S.cn_all.data());
        S.head.assign(S.n_chr, ""); S.unit.assign(S.n_chr, -1);
        // Two batches, every chromosome a unit:
        //   probe batch   -- localhap.cpp:136-170 for all chromosomes of the sample at once: junction CNs, bias, getIndelBias, shortcut;
        //   solve         -- per chromosome, in order: ILP model -> <prefix>.lp/.mps -> `cbc` -> <prefix>.sol (the reference
        //                    reuses the same file names for every chromosome, so each .sol is taken in right after its solve);
        //   reconstruct batch -- localhap.cpp:222-262 for all chromosomes (of all samples) at once.
        // This program's own stdout lines come out in the reference's order (per chromosome: the three ILP progress lines, then
        // the path lines); they are held back until the reconstruct batch is done, so only the solver's own chatter, which
        // the reference interleaves between them, moves to the front.  Both batches take their streams, events, pinned words and
        // device blocks from the engine's per-device pool: one resident context serves them in turn.
        ambi_batch_t* probe; ambi_batch_create(&probe);
        for (int c = 0; c < S.n_chr; c++)
            if ((rc = ambi_batch_add_chromosome(probe, g, c, 0, nullptr, nullptr, 0)) < 0) return die(ambi_error_string(rc));
        if ((rc = ambi_batch_upload(probe)) != 0 || (rc = ambi_batch_run(probe, 0, nullptr)) != 0 || (rc = ambi_batch_download(probe)) != 0)
            return die(std::string("engine: ") + ambi_error_string(rc));
        for (int c = 0; c < S.n_chr; c++) {
            int32_t s, e;
            ambi_graph_chromosome(g, c, &s, &e);
            const int n = e - s + 1;
            ambi_unit_result_t pr; ambi_batch_unit_result(probe, c, &pr);
            std::vector<double> junc_cn(2 * (n + 1)), seg_cn(n + 1);
            std::vector<int32_t> inv(n + 1);
            ambi_batch_unit_prepare(probe, c, junc_cn.data(), seg_cn.data(), nullptr, inv.data());
            // getIndelBias of chromosome c has edited its segment CNs (localhap.cpp:147); the ILP of chromosome c sees the
            // edits of chromosomes <= c only (its loop bound is the CN sum over ALL segments, LGM.cpp:4708-4711)
            for (int i = 1; i <= n; i++) { S.cn_all[s - 1 + i - 1] = seg_cn[i]; if (inv[i] >= 0) S.num_inv++; }
            if (pr.status == AMBI_ST_SHORTCUT) {
                if ((rc = ambi_batch_add_chromosome(b, g, c, 0, nullptr, nullptr, 0)) < 0) return die(ambi_error_string(rc));
                S.unit[c] = rc; n_units++;
                continue;
            }
            double max_cn = 0;
            for (double v : S.cn_all) max_cn += v;
            ambi_ilp_t* ilp = nullptr;
            // BFB_ILP (LGM.cpp:4397-4752): the rows are listed on the host, the non-zeros (56.5 M at 256 segments) written by
            // ambi_ilp_fill_kernel on the device; small chromosomes are not worth the launch and take the host generator
            if (n >= 32) rc = ambi_ilp_build_device(g, c, seg_cn.data(), junc_cn.data(), pr.bias, max_cn, junc_info ? 1 : 0, nullptr, &ilp);
            else rc = ambi_ilp_build(g, c, seg_cn.data(), junc_cn.data(), pr.bias, max_cn, junc_info ? 1 : 0, &ilp);
            if (rc != 0) return die(ambi_error_string(rc));
            S.head[c] = "Declare done\nILP formula done\nVariable constrains done\n";
            ambi_ilp_write_mps(ilp, (prefix + ".mps").c_str());   // LGM.cpp:4749-4750: both side files
            ambi_ilp_write_lp(ilp, (prefix + ".lp").c_str());
            ambi_ilp_destroy(ilp);
            (void)remove(("./" + prefix + ".sol").c_str());        // a stale .sol of an earlier chromosome or run must not pass for this solve
            std::string cmd = "cbc " + prefix + ".lp solve solu " + prefix + ".sol";   // localhap.cpp:179-181
            if (solver_timeout > 0) { char t[64]; snprintf(t, sizeof(t), "timeout -k 5 %.0f ", solver_timeout); cmd = t + cmd; }
            std::cout.flush();
            const int solver_rc = system(cmd.c_str());
            // the reference ignores the exit status (a missing .sol is what it notices, localhap.cpp:187-190); say what happened
            if (solver_rc != 0) {
                const int code = WIFEXITED(solver_rc) ? WEXITSTATUS(solver_rc) : -1;
                if (solver_timeout > 0 && code == 124) std::cerr << "ILP error: cbc did not finish within " << solver_timeout << " s (chromosome " << c << ")" << std::endl;
                else std::cerr << "ILP warning: `" << cmd << "` ended with status " << code << std::endl;
            }
            rc = ambi_batch_add_chromosome_sol(b, g, c, ("./" + prefix + ".sol").c_str());
            if (rc == AMBI_ERR_SOL_OPEN) return die("ILP error: cannot open file ./" + prefix + ".sol");   // localhap.cpp:187-190
            if (rc < 0) return die(ambi_error_string(rc));
            S.unit[c] = rc; n_units++;
        }
        ambi_batch_destroy(probe);
    }
    const uint32_t flags = (reversed ? AMBI_FLAG_REVERSED : 0u) | (all ? AMBI_FLAG_ALL : 0u);
    if (n_devices != 0) rc = ambi_batch_run_sharded(b, flags, device_list.empty() ? nullptr : device_list.data(), n_devices);
    else if ((rc = ambi_batch_upload(b)) == 0 && (rc = ambi_batch_run(b, flags, nullptr)) == 0) rc = ambi_batch_download(b);
    if (rc != 0) return die(std::string("engine: ") + ambi_error_string(rc));
    int refused_total = 0;
    for (Sample& S : samples) {
        ambi_graph_t* g = S.g;
        std::vector<std::vector<int32_t>> paths(S.n_chr);
        std::vector<OutJ> out_acc;
        int refused = 0;
        for (int c = 0; c < S.n_chr; c++) {
            const int u = S.unit[c];
            std::cout << S.head[c];
            ambi_unit_result_t r; ambi_batch_unit_result(b, u, &r);
            if (r.status < 0 || r.status == AMBI_ST_NO_VALID_ORDER) {
                // where the reference would print this chromosome's path.  AMBI_ERR_REF_UB: the reference itself reads past the
                // end of its breakpoint vector on this input (LGM.cpp:3436-3442) -- whatever it prints there is not defined by
                // its source, so nothing is printed here; the other chromosomes follow, the exit status is 1.
                std::cout.flush();
                std::cerr << "bfb: chromosome " << c << ": " << ambi_error_string(r.status) << std::endl;
                refused++;
                continue;
            }
            std::vector<int32_t> p(r.path_len), q(r.path_indel_len);
            ambi_batch_unit_path(b, u, 0, p.data(), r.path_len);
            ambi_batch_unit_path(b, u, 1, q.data(), r.path_indel_len);
            if (all && r.status == AMBI_ST_OK) {
                // --all: one line per valid order; the flipped orientation only if the last order was invalid (LGM.cpp:3672-3695)
                const int64_t stride = 2ll * r.path_len + 64;
                for (int pass = 0; pass < 2; pass++) {
                    int64_t nv = 0;
                    ambi_batch_all_count(b, u, pass, &nv);
                    for (int64_t lo = 0; lo < nv; lo += 64) {
                        const int64_t cnt = nv - lo < 64 ? nv - lo : 64;
                        std::vector<int32_t> len((size_t)cnt), cells((size_t)(cnt * stride));
                        if ((rc = ambi_batch_all_paths(b, u, pass, lo, cnt, len.data(), cells.data(), stride)) != 0)
                            return die(std::string("bfb --all: ") + ambi_error_string(rc));
                        for (int64_t j = 0; j < cnt; j++) {
                            if (len[j] < 0) return die(std::string("bfb --all: ") + ambi_error_string(len[j]));
                            std::vector<int32_t> pj(cells.begin() + j * stride, cells.begin() + j * stride + len[j]);
                            std::cout << path_text(g, pj) << std::endl;
                        }
                    }
                }
            } else
                std::cout << path_text(g, p) << std::endl;                               // printBFB (LGM.cpp:3411-3429)
            if (r.status == AMBI_ST_INFEASIBLE) std::cout << "ILP is unsolvable.\n";    // localhap.cpp:217
            else if (r.indel_printed) std::cout << "BFB path with insertion, deletion, or duplication:\n" << path_text(g, q) << std::endl;
            if (S.trx_before && r.status == AMBI_ST_OK) {
                // virusBFB (localhap.cpp:263): the path goes back to the segments of the file; its junction steps are counted there
                std::vector<int32_t> back(q);
                back.resize(q.size() + 8);
                std::string text(64 + 16 * (q.size() + 8) * 2, '\0');
                int64_t tl = 0;
                const int nl = ambi_graph_trx_restore(g, back.data(), (int32_t)q.size(), (int32_t)back.size(), &text[0], (int64_t)text.size(), &tl);
                if (nl < 0) { std::cout.flush(); std::cerr << "bfb: chromosome " << c << ": " << ambi_error_string(nl) << std::endl; refused++; continue; }
                back.resize(nl);
                text.resize((size_t)std::min<int64_t>(tl, (int64_t)text.size() - 1));
                std::cout << text;
                paths[c] = back;
                for (int i = 0; i + 1 < nl; i++) {
                    const int uu = back[i], vv = back[i + 1];
                    if (!(std::abs(std::abs(uu) - std::abs(vv)) == 1 && (uu > 0) == (vv > 0))) merge_steps(out_acc, uu, vv, 1, true);   // localhap.cpp:267-289
                }
                continue;
            }
            paths[c] = q;
            std::vector<int32_t> ju(r.n_out_junc), jv(r.n_out_junc), jc(r.n_out_junc);
            ambi_batch_unit_out_juncs(b, u, ju.data(), jv.data(), jc.data(), r.n_out_junc);
            for (int k = 0; k < r.n_out_junc; k++) merge_steps(out_acc, ju[k], jv[k], jc[k], true);   // localhap.cpp:267-289
        }
        if (refused) { refused_total += refused; continue; }     // (this sample has no translocation stage and leaves no side-file rows)
        int path_len = 0, cn_sum = 0, max_cn_i = 0;
        for (auto& p : paths) path_len += (int)p.size();
        for (double v : S.cn_all) { cn_sum += v; max_cn_i = (max_cn_i > v) ? max_cn_i : v; }   // localhap.cpp:290-293 (int fed with doubles)
        if (S.ins_mode == 2 || S.con_mode == 2) {   // localhap.cpp:295-316
            if (!S.main_chr[0]) return die("BFB-TRX needs PROP M:<chr>");
            std::cout << "BFB with translocation:\n";
            std::vector<int64_t> offs(S.n_chr + 1, 0);
            for (int c = 0; c < S.n_chr; c++) offs[c + 1] = offs[c] + (int64_t)paths[c].size();
            std::vector<int32_t> flat((size_t)offs[S.n_chr] + 1), out((size_t)offs[S.n_chr] * 2 + 16);
            for (int c = 0; c < S.n_chr; c++) std::copy(paths[c].begin(), paths[c].end(), flat.begin() + offs[c]);
            int len = ambi_translocation_bfb(g, flat.data(), offs.data(), S.n_chr, out.data(), (int32_t)out.size());
            if (len < 0) return die(ambi_error_string(len));
            out.resize(len);
            std::cout << path_text(g, out) << std::endl;
            for (int i = 0; i + 1 < len; i++) {
                int u = out[i], v = out[i + 1];
                if (!(std::abs(std::abs(u) - std::abs(v)) == 1 && (u > 0) == (v > 0))) merge_steps(out_acc, u, v, 1, false);
            }
        }
        // side files (localhap.cpp:326-337, 382-388)
        {
            int32_t n_seg = S.n_seg, n_junc = S.n_junc, n_chr = S.n_chr;
            ambi_graph_sizes(g, &n_seg, &n_junc, &n_chr);
            std::vector<int32_t> js(n_junc), jt(n_junc); std::vector<int8_t> jsd(n_junc), jtd(n_junc); std::vector<double> jcn(n_junc);
            ambi_graph_junctions(g, js.data(), jsd.data(), jt.data(), jtd.data(), nullptr, jcn.data(), nullptr, nullptr);
            auto chrom = [&](int id) { char nm[256]; ambi_graph_chrom_name(g, id, nm, sizeof(nm)); return std::string(nm); };
            auto vend = [&](int id, int dir) { return dir > 0 ? S.seg_end[id - 1] : S.seg_start[id - 1]; };      // Vertex::getEnd
            auto vstart = [&](int id, int dir) { return dir > 0 ? S.seg_start[id - 1] : S.seg_end[id - 1]; };    // Vertex::getStart
            std::ofstream sv("simulation_sv.txt", std::ios_base::app);
            for (int j = 0; j < n_junc; j++)
                sv << S.lh << "\t" << juncs << "\t" << chrom(js[j]) << "\t" << vend(js[j], jsd[j]) << "\t" << (jsd[j] > 0 ? '+' : '-') << "\t"
                   << chrom(jt[j]) << "\t" << vstart(jt[j], jtd[j]) << "\t" << (jtd[j] > 0 ? '+' : '-') << "\t" << jcn[j] << "\tinput\n";
            // (PROP I1 / C1: the vertices of the restored paths are segments of the FILE)
            std::vector<int32_t> fs, fe;
            if (S.trx_before) {
                int32_t fn = 0, fj = 0, fc = 0;
                ambi_graph_sizes(S.g_file, &fn, &fj, &fc);
                fs.resize(fn); fe.resize(fn);
                ambi_graph_segments(S.g_file, nullptr, nullptr, fs.data(), fe.data(), nullptr, nullptr);
            }
            auto ochrom = [&](int id) { if (!S.trx_before) return chrom(id); char nm[256]; ambi_graph_chrom_name(S.g_file, id, nm, sizeof(nm)); return std::string(nm); };
            auto ovend = [&](int id, int dir) { return !S.trx_before ? vend(id, dir) : (dir > 0 ? fe[id - 1] : fs[id - 1]); };
            auto ovstart = [&](int id, int dir) { return !S.trx_before ? vstart(id, dir) : (dir > 0 ? fs[id - 1] : fe[id - 1]); };
            for (auto& j : out_acc) {
                int a = std::abs(j.u), bb = std::abs(j.v), ad = j.u > 0 ? 1 : -1, bd = j.v > 0 ? 1 : -1;
                sv << S.lh << "\t" << juncs << "\t" << ochrom(a) << "\t" << ovend(a, ad) << "\t" << (ad > 0 ? '+' : '-') << "\t"
                   << ochrom(bb) << "\t" << ovstart(bb, bd) << "\t" << (bd > 0 ? '+' : '-') << "\t" << j.cn << "\toutput\n";
            }
            auto t_end = std::chrono::steady_clock::now();
            std::ofstream tf("time.csv", std::ios_base::app);
            tf << S.lh.substr(0, S.lh.find(".")) << "," << n_seg << "," << S.num_inv << "," << n_junc - S.num_inv << "," << cn_sum << "," << path_len << ","
               << max_cn_i << "," << std::chrono::duration_cast<std::chrono::microseconds>(t_end - S.t_begin).count() / 1000000.0 << "\n";
        }
    }
    ambi_batch_destroy(b);
    for (Sample& S : samples) { ambi_graph_destroy(S.g); if (S.g_file) ambi_graph_destroy(S.g_file); }
    if (refused_total) return die("bfb: " + std::to_string(refused_total) + " chromosome(s) without a path");
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    setenv("GPU_MAX_HW_QUEUES", "8", 0);   // the engine's four streams on hardware queues of their own (before the HIP runtime starts)
    Args A = parse(argc, argv);
    if (A.help) {
        std::cout << "Local Haplotype constructer\nUsage:\n  Ambigram --op bfb|sc_bfb --in_lh <file[,file...]> --lp_prefix <name> [--juncdb <file> --junc_info true] "
                     "[--reversed true] [--all true] [--solver_timeout <seconds>] [--devices N|all|<ordinal,ordinal,...>]\n"
                     "  --op bfb: --in_lh may list several samples; their chromosomes are reconstructed as one batch, over the devices named by --devices\n";
        return 0;
    }
    const std::string op = A.kv.count("op") ? A.kv["op"] : "";
    std::cout << op << std::endl;   // localhap.cpp:47
    if (op == "sc_bfb") return run_sc_bfb(A);
    if (op != "bfb") return 0;   // the reference does nothing for any other op (localhap.cpp:49, :390)
    return run_bfb(A);
}
