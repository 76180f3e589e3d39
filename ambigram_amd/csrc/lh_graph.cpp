// lh_graph.cpp -- host-side readers/writers around the BFB path (see lh_graph.hpp).
#include "lh_graph.hpp"
#include <array>
#include <memory>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace ambi {

const char* lh_error_string(int code) {
    switch (code) {
        case LH_OK: return "ok";
        case LH_ERR_OPEN: return "Cannot open file";
        case LH_ERR_MALFORMED: return "malformed .lh line (missing column)";
        case LH_ERR_UNKNOWN_SEG: return "junction or SOURCE/SINK refers to an unknown segment";
        case LH_ERR_SOURCE_SINK: return "SOURCE and SINK lists differ in length";
        case LH_ERR_PLOIDY: return "input error: ploidy / purity information missing";
        case LH_ERR_SEG_IDS: return "segment ids must be 1..N in file order";
        case LH_ERR_SOL_OPEN: return "ILP error: cannot open file";
        case LH_ERR_LINE_TOO_LONG: return "line longer than 8191 bytes";
        case LH_ERR_UNSUPPORTED: return "TRX-BFB (PROP I1/C1): the reference reads what nothing has set on this input (no junction between the listed chromosomes, a one-vertex path, or a .juncs file with these modes)";
        default: return "unknown error";
    }
}

// ---- tokenizer with strtok's rules (skip leading delimiters, cut at the next one) but local state ----
namespace {
struct Cutter {
    char* p;
    explicit Cutter(char* s) : p(s) {}
    char* next(const char* delims) {
        if (!p) return nullptr;
        while (*p && strchr(delims, *p)) p++;
        if (!*p) { p = nullptr; return nullptr; }
        char* tok = p;
        while (*p && !strchr(delims, *p)) p++;
        if (*p) { *p = '\0'; p++; } else p = nullptr;
        return tok;
    }
};
std::string dtoa(double v) { std::ostringstream o; o << v; return o.str(); }   // what `cout << double` prints
bool same_edges(int s1, int sd1, int t1, int td1, int s2, int sd2, int t2, int td2) {
    // Graph.cpp:489-499 compares the printed forms of edge A and edge B
    int a1 = sd1 > 0 ? s1 : -s1, b1 = td1 > 0 ? t1 : -t1;
    int a2 = sd2 > 0 ? s2 : -s2, b2 = td2 > 0 ? t2 : -t2;
    return (a1 == a2 && b1 == b2) || (a1 == -b2 && b1 == -a2);
}
}  // namespace

// one key for an edge (a, b) and its complement (-b, -a): the smaller of the two pairs (ids > 0, so signs are faithful)
static uint64_t edge_key(int s, int sd, int t, int td) {
    int64_t a = sd > 0 ? s : -(int64_t)s, b = td > 0 ? t : -(int64_t)t;
    int64_t a2 = -b, b2 = -a;
    if (a2 < a || (a2 == a && b2 < b)) { a = a2; b = b2; }
    return ((uint64_t)(uint32_t)(int32_t)a << 32) | (uint64_t)(uint32_t)(int32_t)b;
}

int LhGraph::find_junction(int src, int sdir, int tgt, int tdir) const {
    if (src > 0 && tgt > 0 && junc_index.size() == (size_t)n_junc()) {   // every junction is indexed (ids are positive)
        auto it = junc_index.find(edge_key(src, sdir, tgt, tdir));
        return it == junc_index.end() ? -1 : it->second;
    }
    for (int i = 0; i < n_junc(); i++)
        if (same_edges(j_src[i], j_sdir[i], j_tgt[i], j_tdir[i], src, sdir, tgt, tdir)) return i;
    return -1;
}

bool LhGraph::add_junction(int src, int sdir, int tgt, int tdir, double cov, double cn, bool inferred, bool bounded) {
    for (; segs_indexed < seg_id.size(); segs_indexed++) seg_index.emplace(seg_id[segs_indexed], (int32_t)segs_indexed);
    if (!seg_index.count(src) || !seg_index.count(tgt)) return false;
    if (find_junction(src, sdir, tgt, tdir) >= 0) return true;   // duplicate: ignored (Graph.cpp:592-595)
    if (src > 0 && tgt > 0 && junc_index.size() == (size_t)n_junc()) junc_index.emplace(edge_key(src, sdir, tgt, tdir), (int32_t)n_junc());
    j_src.push_back(src); j_tgt.push_back(tgt); j_sdir.push_back((int8_t)sdir); j_tdir.push_back((int8_t)tdir);
    j_cov.push_back(cov); j_cn.push_back(cn); j_inferred.push_back(inferred); j_bounded.push_back(bounded);
    return true;
}

int hap_depth(LhGraph& g) {   // Graph.cpp:312-367
    if (g.avg_ploidy < 0) {
        if (g.avg_tumor_ploidy < 0 || g.purity < 0) return LH_ERR_PLOIDY;
        g.avg_ploidy = g.purity * g.avg_tumor_ploidy + (1 - g.purity) * 2;
    } else if (g.avg_tumor_ploidy >= 0) {
        if (g.purity < 0) g.log.push_back("WARN: no purity information provided, use the given AVG_PLOIDY");
        else {
            double pt = g.purity * g.avg_tumor_ploidy;
            g.ratio = 1 - pt / (pt + (1 - g.purity) * 2);
            g.ratio_set = true;
            double ap = g.purity * g.avg_tumor_ploidy + (1 - g.purity) * 2;
            if (!(std::abs(g.avg_ploidy - ap) <= 0.1)) g.avg_ploidy = ap;
        }
    } else {
        g.log.push_back("WARN: only AVG_PLOIDY is given, use that");
    }
    g.haploid_depth = g.avg_cov_raw * g.purity / g.avg_ploidy;
    g.avg_coverage = g.avg_ploidy * g.haploid_depth;
    g.avg_cov_junc = g.avg_ploidy * g.haploid_depth;
    return LH_OK;
}

void copy_num(LhGraph& g) {   // Graph.cpp:369-405: only entries with CN <= 0 are (re)computed
    for (int i = 0; i < g.n_seg(); i++) {
        if (g.seg_cn[i] > 0) continue;
        double c;
        if (g.seg_id[i] >= g.virus_seg_start) c = g.seg_cov[i] / g.avg_cov_raw * 2;
        else c = (g.seg_cov[i] - g.avg_cov_raw * g.ratio) / g.haploid_depth;
        g.seg_cn[i] = std::max(c, 0.0);
        g.log.push_back("SEG" + std::to_string(g.seg_id[i]) + " " + dtoa(g.seg_cov[i]) + " " + dtoa(g.seg_cn[i]));
    }
    for (int i = 0; i < g.n_junc(); i++) {
        if (g.j_cn[i] > 0) continue;
        if (g.j_inferred[i]) g.log.push_back(dtoa(g.haploid_depth));
        g.j_cn[i] = std::max((g.j_cov[i] - g.avg_cov_raw * g.ratio) / g.haploid_depth, 0.0);
    }
}

static void read_props(const std::string& path, LhGraph& g) {   // LGM.cpp:3941-3987
    std::ifstream f(path);
    std::string line;
    while (std::getline(f, line)) {
        std::istringstream ss(line);
        std::string w;
        if (!(ss >> w) || w != "PROP") continue;
        while (ss >> w) {
            auto fields = [&](size_t from, std::vector<std::string>& out) {
                size_t last = from;
                while (true) {
                    size_t pos = w.find(':', last);
                    out.push_back(w.substr(last, pos == std::string::npos ? std::string::npos : pos - last));
                    if (pos == std::string::npos) break;
                    last = pos + 1;
                }
            };
            if (w[0] == 'M') g.main_chr = w.size() > 2 ? w.substr(2) : "";
            else if (w[0] == 'I') {
                size_t from = 2;
                if (w.size() > 1 && w[1] != ':') { g.ins_mode = w[1] - '0'; from = 3; } else g.ins_mode = 2;
                fields(from, g.ins_chr);
            } else if (w[0] == 'C') {
                size_t from = 2;
                if (w.size() > 1 && w[1] != ':') { g.con_mode = w[1] - '0'; from = 3; } else g.con_mode = 2;
                fields(from, g.con_chr);
            } else if (w[0] == 'S') {
                std::vector<std::string> t;
                fields(2, t);
                for (auto& x : t) g.start_segs.push_back(atoi(x.c_str()));
            }
        }
    }
}

int read_lh(const std::string& path, LhGraph& g) {
    std::ifstream f(path);
    if (!f) return LH_ERR_OPEN;
    g.log.push_back("Reading graph...");
    std::string raw;
    bool more = true;
    while (more) {
        // the reference loops `while(!eof) getline(buf, 8192)`: a final line without '\n' is still processed,
        // and a trailing '\n' yields one last empty line
        if (!std::getline(f, raw)) { more = false; if (f.eof() && raw.empty()) { /* one empty line */ } }
        if (raw.size() > 8191) return LH_ERR_LINE_TOO_LONG;
        std::vector<char> buf(raw.begin(), raw.end());
        buf.push_back('\0');
        raw.clear();
        char* line = buf.data();
        const char* q = line;
        while (*q == ' ' || *q == '\t') q++;
        if (*q == '#') continue;
        Cutter c(line);
        char* key = c.next(" \t");
        if (!key) continue;
        auto need = [&](const char* d) -> char* { return c.next(d); };
        if (!strcmp(key, "SAMPLE_NAME")) { char* t = need(" "); if (!t) return LH_ERR_MALFORMED; g.sample_name = t; }
        else if (!strcmp(key, "AVG_CHR_SEG_DP")) {
            char* t = need(" ");
            if (t) { Cutter s(t); while (char* x = s.next(",")) g.avg_coverages.push_back(atof(x)); }
        }
        else if (!strcmp(key, "AVG_WHOLE_HOST_DP")) { char* t = need(" "); if (!t) return LH_ERR_MALFORMED; g.avg_cov_raw = atof(t); }
        else if (!strcmp(key, "AVG_VIRUS_SEG_DP")) { char* t = need(" "); if (!t) return LH_ERR_MALFORMED; g.avg_virus_dp = atof(t); }
        else if (!strcmp(key, "VIRUS_START")) { char* t = need(" "); if (!t) return LH_ERR_MALFORMED; g.virus_seg_start = atoi(t); g.virus_seg_start_set = true; }
        else if (!strcmp(key, "AVG_JUNC_DP")) { char* t = need(" "); if (!t) return LH_ERR_MALFORMED; g.avg_cov_junc = atof(t); }
        else if (!strcmp(key, "PURITY")) { char* t = need(" "); if (!t) return LH_ERR_MALFORMED; g.purity = atof(t); }
        else if (!strcmp(key, "AVG_TUMOR_PLOIDY")) { char* t = need(" "); if (!t) return LH_ERR_MALFORMED; g.avg_tumor_ploidy = atof(t); }
        else if (!strcmp(key, "AVG_PLOIDY")) { char* t = need(" "); if (!t) return LH_ERR_MALFORMED; g.avg_ploidy = atof(t); }
        else if (!strcmp(key, "PLOIDY")) {
            char* t = need(" "); if (!t) return LH_ERR_MALFORMED;
            g.ploidy = t;
            Cutter s(t); char* x = s.next("m"); if (!x) return LH_ERR_MALFORMED;
            g.expected_ploidy = atoi(x);
        }
        else if (!strcmp(key, "SOURCE") || !strcmp(key, "SINK")) {
            std::vector<int32_t>& dst = key[1] == 'O' ? g.source_ids : g.sink_ids;
            char* t = need(" ");
            if (t) { Cutter s(t); while (char* x = s.next(",")) dst.push_back(atoi(x)); }
        }
        else if (!strcmp(key, "SEG")) {
            char* node = need(" "); char* cov = need(" "); char* cn = need(" ");
            if (!node || !cov || !cn) return LH_ERR_MALFORMED;
            Cutter s(node);
            char* h = s.next(":"); char* id = s.next(":"); char* chrom = s.next(":"); char* st = s.next(":"); char* en = s.next(":");
            if (!h || !id || !chrom || !st || !en) return LH_ERR_MALFORMED;
            int sid = atoi(id);
            int chr = -1;   // uninitialised in the reference when no SOURCE/SINK range matches
            for (size_t i = 0; i < g.source_ids.size() && i < g.sink_ids.size(); i++)
                if (g.source_ids[i] <= sid && sid <= g.sink_ids[i]) chr = (int)i;
            g.seg_id.push_back(sid); g.seg_chr.push_back(chr); g.seg_chrom.push_back(chrom);
            g.seg_start.push_back(atoi(st)); g.seg_end.push_back(atoi(en));
            g.seg_cov.push_back(std::max(atof(cov), 0.0)); g.seg_cn.push_back(atof(cn));
            g.seg_partition.push_back(0);
        }
        else if (!strcmp(key, "JUNC")) {
            char* sn = need(" "); char* tn = need(" "); char* cov = need(" "); char* cn = need(" ");
            char* inf = need(" "); char* bnd = need(" ");
            if (!sn || !tn || !cov || !cn || !inf || !bnd) return LH_ERR_MALFORMED;
            double jc = atof(cov), jn = atof(cn);
            if (jc <= 0 && jn <= 0) continue;   // Graph.cpp:211
            Cutter a(sn); char* h1 = a.next(":"); char* sid = a.next(":"); char* sd = a.next(":");
            Cutter b(tn); char* h2 = b.next(":"); char* tid = b.next(":"); char* td = b.next(":");
            if (!h1 || !sid || !sd || !h2 || !tid || !td) return LH_ERR_MALFORMED;
            if (!g.add_junction(atoi(sid), sd[0] == '+' ? 1 : -1, atoi(tid), td[0] == '+' ? 1 : -1, jc, jn, inf[0] == 'I', bnd[0] == 'B'))
                return LH_ERR_UNKNOWN_SEG;
        }
    }
    if (g.source_ids.size() != g.sink_ids.size()) return LH_ERR_SOURCE_SINK;
    for (int i = 0; i < g.n_seg(); i++) if (g.seg_id[i] != i + 1) return LH_ERR_SEG_IDS;
    for (size_t i = 0; i < g.source_ids.size(); i++) {
        if (g.source_ids[i] < 1 || g.source_ids[i] > g.n_seg() || g.sink_ids[i] < 1 || g.sink_ids[i] > g.n_seg()) return LH_ERR_UNKNOWN_SEG;
    }
    int rc = hap_depth(g);
    if (rc != LH_OK) return rc;
    copy_num(g);
    read_props(path, g);
    if (g.ins_mode == 1 || g.con_mode == 1) { rc = trx_rebuild(g); if (rc != LH_OK) return rc; }   // localhap.cpp:79-88
    set_partitions(g);
    return LH_OK;
}

void set_partitions(LhGraph& g) {   // localhap.cpp:94-98
    for (int c = 0; c < g.n_chr(); c++)
        for (int j = g.source_ids[c]; j <= g.sink_ids[c]; j++)
            if (j >= 1 && j <= g.n_seg()) g.seg_partition[j - 1] = c;
}

int read_juncs(LhGraph& g, const std::string& path) {   // LGM.cpp:5096-5156
    if (path.empty()) return LH_OK;
    if (g.trx) return LH_ERR_UNSUPPORTED;   // PROP I1 / C1: readComponents reads the rebuilt graph's mean coverage, which nothing ever sets (Graph.cpp:25-34)
    std::ifstream f(path);
    std::string line;
    auto& res = g.components;
    while (std::getline(f, line)) {
        std::istringstream iss(line);
        std::vector<int> ids; std::vector<char> sg; std::string w;
        while (iss >> w) { ids.push_back(atoi(w.substr(0, w.size() - 1).c_str())); sg.push_back(w.back()); }
        for (int id : ids) if (id < 1 || id > g.n_seg()) return LH_ERR_UNKNOWN_SEG;
        size_t last = 0;
        for (size_t i = 1; i < ids.size(); i++) {
            if (g.seg_partition[ids[last] - 1] != g.seg_partition[ids[i] - 1] || sg[i - 1] != sg[i]) {
                if (i - last >= 2) {
                    std::vector<int32_t> sub(ids.begin() + last, ids.begin() + i);
                    std::sort(sub.begin(), sub.end());
                    res.push_back(sub);
                }
                int s = ids[i - 1], t = ids[i];
                g.log.push_back(std::to_string(s) + sg[i - 1] + " -> " + std::to_string(t) + sg[i]);
                int sd = sg[i - 1] == '+' ? 1 : -1, td = sg[i] == '+' ? 1 : -1;
                int j = g.find_junction(s, sd, t, td);
                if (j < 0) g.add_junction(s, sd, t, td, g.avg_coverage, 1, false, true);
                else if (g.j_cn[j] < 2) g.j_cn[j] = 2;
                last = i;
            }
        }
        if (ids.size() >= last + 2) {
            std::vector<int32_t> sub(ids.begin() + last, ids.end());
            std::sort(sub.begin(), sub.end());
            res.push_back(sub);
        }
    }
    std::sort(res.begin(), res.end());
    res.erase(std::unique(res.begin(), res.end()), res.end());
    return LH_OK;
}

// Graph::writeGraph (Graph.cpp:239-266): the graph as it stands (after the copy-number maths and any .juncs additions) back
// into .lh text.  The reference writes a fixed sample name ("TEST"), only these header keys, every segment with its
// lower-bound flag ('B': a freshly read segment always has it, Segment.cpp:16) and numbers through operator<< of a default
// ostream (= printf %g); it also prints "write seg" to stdout, which goes to the graph's log here.
int write_lh(LhGraph& g, const std::string& path) {
    FILE* f = fopen(path.c_str(), "w");
    if (!f) return LH_ERR_OPEN;
    auto num = [](double v) { char b[64]; snprintf(b, sizeof b, "%g", v); return std::string(b); };
    std::string src, snk;
    for (int32_t v : g.source_ids) { src += std::to_string(v); src += ','; }      // Graph.cpp:780-796: a comma behind every id
    for (int32_t v : g.sink_ids) { snk += std::to_string(v); snk += ','; }
    fprintf(f, "SAMPLE_NAME TEST\nAVG_SEG_DP %s\nAVG_JUNC_DP %s\nPURITY %s\nAVG_PLOIDY %s\nPLOIDY %s\nSOURCE %s\nSINK %s\n", num(g.avg_coverage).c_str(),
            num(g.avg_cov_junc).c_str(), num(g.purity).c_str(), num(g.avg_ploidy).c_str(), g.ploidy.c_str(), src.c_str(), snk.c_str());
    g.log.push_back("write seg");
    for (int i = 0; i < g.n_seg(); i++)
        fprintf(f, "SEG H:%d:%s:%d:%d %s %s B\n", g.seg_id[i], g.seg_chrom[i].c_str(), g.seg_start[i], g.seg_end[i], num(g.seg_cov[i]).c_str(), num(g.seg_cn[i]).c_str());
    for (int j = 0; j < g.n_junc(); j++)
        fprintf(f, "JUNC H:%d:%c H:%d:%c %s %s %c %c\n", g.j_src[j], g.j_sdir[j] > 0 ? '+' : '-', g.j_tgt[j], g.j_tdir[j] > 0 ? '+' : '-', num(g.j_cov[j]).c_str(),
                num(g.j_cn[j]).c_str(), g.j_inferred[j] ? 'I' : 'U', g.j_bounded[j] ? 'B' : 'U');
    const bool bad = ferror(f) != 0;
    if (fclose(f) != 0 || bad) return LH_ERR_OPEN;
    return LH_OK;
}


// ------------------------------------------------------------------------------------------------------------------
// TRX-BFB (PROP I1 / C1)
// ------------------------------------------------------------------------------------------------------------------
namespace {
// The rebuilt graph while it is being put together: segment k (1-based) is a copy of segment from[k-1] of the file on chromosome
// chr[k-1]; `conv` is the reference's own map (file id -> rebuilt id, 0 = no place) -- std::unordered_map<int,int> with the same
// sequence of inserts, because the reference PRINTS it in iteration order.
struct Rebuild {
    std::vector<int32_t> from, chr;
    std::unordered_map<int, int> conv;
    void place(int file_id, int chr_id) { conv.insert({file_id, (int)from.size() + 1}); from.push_back(file_id); chr.push_back(chr_id); }
    void drop(int file_id) { conv.insert({file_id, 0}); }
};
int a_src_of(const LhGraph& g, int j) { return g.j_sdir[j] > 0 ? g.j_src[j] : -g.j_src[j]; }
int a_tgt_of(const LhGraph& g, int j) { return g.j_tdir[j] > 0 ? g.j_tgt[j] : -g.j_tgt[j]; }
// the rebuilt graph from the placement + the junctions that found a place (LGM.cpp:4254-4293 / :4356-4393)
void finish_rebuild(LhGraph& g, Rebuild& R, std::vector<int32_t>& unused, const std::vector<std::array<int32_t, 5>>& kept /* j, id1, id2, dir1, dir2 */) {
    auto T = std::make_shared<TrxBefore>();
    T->original = g;
    T->original.trx.reset();
    T->unused_sv = unused;
    T->original_of.assign(R.from.size() + 1, 0);
    g.log.push_back("Seg conversion:");
    for (auto it = R.conv.begin(); it != R.conv.end(); ++it) {
        g.log.push_back(std::to_string(it->first) + "-" + std::to_string(it->second));
        if (it->second > 0) T->original_of[it->second] = it->first;
    }
    const LhGraph& O = T->original;
    LhGraph N;
    // Graph(vector...) (Graph.cpp:25-34) sets purity and the two ploidies to -1 and leaves the other header fields alone; the
    // depths of the file are carried over here so that new.lh has defined numbers where the reference prints whatever was there
    N.sample_name = O.sample_name; N.ploidy = O.ploidy; N.avg_coverages = O.avg_coverages;
    N.avg_cov_raw = O.avg_cov_raw; N.avg_virus_dp = O.avg_virus_dp; N.avg_cov_junc = O.avg_cov_junc; N.avg_coverage = O.avg_coverage;
    N.purity = -1; N.avg_ploidy = -1; N.avg_tumor_ploidy = -1;
    N.main_chr = O.main_chr; N.ins_mode = O.ins_mode; N.con_mode = O.con_mode; N.ins_chr = O.ins_chr; N.con_chr = O.con_chr; N.start_segs = O.start_segs;
    for (size_t k = 0; k < R.from.size(); k++) {
        const int f = R.from[k] - 1;
        N.seg_id.push_back((int32_t)k + 1); N.seg_chr.push_back(R.chr[k]); N.seg_start.push_back(O.seg_start[f]); N.seg_end.push_back(O.seg_end[f]);
        N.seg_partition.push_back(0); N.seg_chrom.push_back(O.seg_chrom[f]); N.seg_cov.push_back(O.seg_cov[f]); N.seg_cn.push_back(O.seg_cn[f]);
    }
    for (auto& k : kept) {   // plain copies: the reference pushes them without the duplicate test of Graph::addJunction
        const int j = k[0];
        N.j_src.push_back(k[1]); N.j_tgt.push_back(k[2]); N.j_sdir.push_back((int8_t)k[3]); N.j_tdir.push_back((int8_t)k[4]);
        N.j_cov.push_back(O.j_cov[j]); N.j_cn.push_back(O.j_cn[j]); N.j_inferred.push_back(O.j_inferred[j]); N.j_bounded.push_back(O.j_bounded[j]);
    }
    N.source_ids.push_back(N.seg_id.front());
    for (int k = 1; k < N.n_seg(); k++)
        if (N.seg_chr[k] != N.seg_chr[k - 1]) { N.sink_ids.push_back(N.seg_id[k - 1]); N.source_ids.push_back(N.seg_id[k]); }
    N.sink_ids.push_back(N.seg_id.back());
    N.log = g.log;
    N.trx = T;
    g = std::move(N);
}
}  // namespace

int trx_rebuild(LhGraph& g) {
    const int n = g.n_seg(), m = g.n_junc();
    for (int i = 0; i < n; i++) if (g.seg_id[i] != i + 1) return LH_ERR_SEG_IDS;
    Rebuild R;
    std::vector<int32_t> unused;
    std::vector<std::array<int32_t, 5>> kept;
    auto chrom = [&](int id) -> const std::string& { return g.seg_chrom[id - 1]; };
    if (g.ins_mode == 1) {
        // ---- insertBeforeBFB (LGM.cpp:4195-4295): the chain of junctions main -> inserted -> ... -> main gives the segment of the main
        // chromosome in front of the insertion (sID), the one behind it (eID) and the inserted segments in between
        const std::vector<std::string>& L = g.ins_chr;
        std::vector<int> ids;
        std::vector<char> seen(m, 0);
        for (size_t i = 1; i < L.size(); i++)
            for (int j = 0; j < m; j++) {
                if (seen[j]) continue;
                const std::string &c1 = chrom(g.j_src[j]), &c2 = chrom(g.j_tgt[j]);
                const bool fwd = L[i - 1] == c1 && L[i] == c2, back = L[i - 1] == c2 && L[i] == c1;
                if (!fwd && !back) continue;
                int id1 = g.j_src[j], id2 = g.j_tgt[j];
                if (back) std::swap(id1, id2);
                if (!ids.empty() && ids.back() != id1) {   // the segments between the end of the last junction and the start of this one (:4217-4222)
                    const int from = ids.back();
                    if (from < id1) { for (int k = from; k < id1; k++) ids.push_back(k); }
                    else { for (int k = from; k > id1; k--) ids.push_back(k); }
                }
                ids.push_back(id1); ids.push_back(id2);
                seen[j] = 1;
                break;
            }
        ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
        if (ids.size() < 2) return LH_ERR_UNSUPPORTED;      // (the reference reads front() / back() of an empty vector)
        if (ids.front() > ids.back()) std::reverse(ids.begin(), ids.end());
        const int sID = ids.front(), eID = ids.back();
        const std::vector<int> inserted(ids.begin() + 1, ids.end() - 1);
        for (int id : inserted) if (id < 1 || id > n) return LH_ERR_UNSUPPORTED;
        std::vector<int> gone_chr;                           // the chromosomes the inserted segments come from disappear as such (:4231-4238)
        for (int id : inserted) gone_chr.push_back(g.seg_chr[id - 1]);
        for (int i = 1; i <= n; i++) {
            if (i < sID || i > eID) {
                if (std::find(gone_chr.begin(), gone_chr.end(), g.seg_chr[i - 1]) != gone_chr.end()) continue;
                R.place(i, g.seg_chr[i - 1]);
            } else {
                R.place(sID, g.seg_chr[sID - 1]);
                for (int k = sID + 1; k < eID; k++) R.drop(k);
                for (int id : inserted) R.place(id, g.seg_chr[sID - 1]);
                R.place(eID, g.seg_chr[eID - 1]);
                i = eID;
            }
        }
        for (int j = 0; j < m; j++) {                        // junctions (:4263-4283)
            if (a_src_of(g, j) == a_tgt_of(g, j)) continue;
            const int s = g.j_src[j], t = g.j_tgt[j];
            int id1 = R.conv[s] - 1, id2 = R.conv[t] - 1;   // (operator[]: a segment without an entry gets one, with 0, as in the reference)
            if (id1 == -1 || id2 == -1) { unused.push_back(j); continue; }
            int d1 = g.j_sdir[j], d2 = g.j_tdir[j];
            const bool touches = std::find(inserted.begin(), inserted.end(), s) != inserted.end() || std::find(inserted.begin(), inserted.end(), t) != inserted.end();
            if (touches) { if (id1 > id2) std::swap(id1, id2); d1 = 1; d2 = 1; }   // the insertion's junctions become plain adjacencies
            g.log.push_back(std::to_string(s) + "-" + std::to_string(t) + " " + std::to_string(id1 + 1) + "-" + std::to_string(id2 + 1));
            kept.push_back({j, id1 + 1, id2 + 1, d1, d2});
        }
    } else {
        // ---- concatBeforeBFB (LGM.cpp:4297-4395): the first junction between the two chromosomes; the kept arm of the first one up
        // to that junction, then the kept arm of the second one from it, both on the first one's chromosome; the other arms have no place
        if (g.con_chr.size() < 2) return LH_ERR_UNSUPPORTED;
        int at = -1;
        for (int j = 0; j < m && at < 0; j++) {
            const std::string &c1 = chrom(g.j_src[j]), &c2 = chrom(g.j_tgt[j]);
            if ((c1 == g.con_chr[0] && c2 == g.con_chr[1]) || (c2 == g.con_chr[0] && c1 == g.con_chr[1])) at = j;
        }
        if (at < 0) return LH_ERR_UNSUPPORTED;              // (the reference goes on with unset variables)
        const int sID = g.j_src[at], eID = g.j_tgt[at];
        const bool s_plus = g.j_sdir[at] > 0, e_plus = g.j_tdir[at] > 0;
        g.log.push_back("Concat segs: " + std::to_string(sID) + (s_plus ? "+" : "-") + " " + std::to_string(eID) + (e_plus ? "+" : "-"));
        const int c1 = g.seg_chr[sID - 1], c2 = g.seg_chr[eID - 1], on = g.seg_chr[sID - 1];
        if (c1 < 0 || c1 >= g.n_chr() || c2 < 0 || c2 >= g.n_chr()) return LH_ERR_UNSUPPORTED;
        const int lo1 = g.source_ids[c1], hi1 = g.sink_ids[c1], lo2 = g.source_ids[c2], hi2 = g.sink_ids[c2];
        if (s_plus) { for (int i = lo1; i <= sID; i++) R.place(i, on); for (int i = sID + 1; i <= hi1; i++) R.drop(i); }
        else { for (int i = hi1; i >= sID; i--) R.place(i, on); for (int i = sID - 1; i >= lo1; i--) R.drop(i); }
        if (e_plus) { for (int i = eID; i <= hi2; i++) R.place(i, on); for (int i = lo2; i < eID; i++) R.drop(i); }
        else { for (int i = eID; i >= lo2; i--) R.place(i, on); for (int i = hi2; i > eID; i--) R.drop(i); }
        for (int i = 1; i <= n; i++) if (g.seg_chr[i - 1] != c1 && g.seg_chr[i - 1] != c2) R.place(i, g.seg_chr[i - 1]);
        for (int j = 0; j < m; j++) {                        // junctions (:4365-4383)
            const int s = g.j_src[j], t = g.j_tgt[j];
            int id1 = R.conv[s] - 1, id2 = R.conv[t] - 1;
            int d1 = g.j_sdir[j], d2 = g.j_tdir[j];
            g.log.push_back(std::to_string(s) + (d1 > 0 ? "+" : "-") + " - " + std::to_string(t) + (d2 > 0 ? "+" : "-") + " " + std::to_string(id1 + 1) + "-" + std::to_string(id2 + 1));
            if (id1 == -1 || id2 == -1) { unused.push_back(j); continue; }
            if ((s == sID && t == eID) || (s == eID && t == sID)) { if (id1 > id2) std::swap(id1, id2); d1 = 1; d2 = 1; }   // the joint becomes an adjacency
            kept.push_back({j, id1 + 1, id2 + 1, d1, d2});
        }
    }
    if (R.from.empty()) return LH_ERR_UNSUPPORTED;
    finish_rebuild(g, R, unused, kept);
    return LH_OK;
}

int trx_restore_path(const LhGraph& rebuilt, std::vector<int32_t>& path, std::vector<std::string>& lines) {
    if (!rebuilt.trx) return LH_ERR_UNSUPPORTED;
    const TrxBefore& T = *rebuilt.trx;
    const LhGraph& O = T.original;
    const int P = (int)path.size();
    if (P < 2) return LH_ERR_UNSUPPORTED;                   // (path->at(1) throws in the reference)
    for (int v : path) { const int id = v < 0 ? -v : v; if (id < 1 || id >= (int)T.original_of.size() || T.original_of[id] < 1) return LH_ERR_UNSUPPORTED; }
    // the vertex a junction of the file leads to from vertex `from` when it is followed to segment `seg`: the FIRST such edge in the
    // order the reference registered them at `from` (Junction.cpp:95-121: by junction; edge A at its source, edge B at its source,
    // no edge B for a fold-back of one segment onto itself); 0 if none
    auto edge_to = [&](int from, int seg) {
        for (int j = 0; j < O.n_junc(); j++) {
            const int as = a_src_of(O, j), at = a_tgt_of(O, j);
            if (as == from && (at < 0 ? -at : at) == seg) return at;
            const bool self_fold = O.j_sdir[j] != O.j_tdir[j] && O.j_src[j] == O.j_tgt[j];
            if (!self_fold && -at == from && (as < 0 ? -as : as) == seg) return -as;
        }
        return 0;
    };
    std::vector<int32_t> out(P);
    std::vector<char> turn(P, 0);                           // a strand change in front of cell k of the rebuilt path (:3841-3845)
    for (int k = 1; k < P; k++) turn[k] = (path[k - 1] > 0) != (path[k] > 0);
    auto file_id = [&](int v) { return T.original_of[v < 0 ? -v : v]; };
    {   // first vertex (:3847-3875)
        const int s1 = file_id(path[0]), s2 = file_id(path[1]);
        if (O.seg_chr[s1 - 1] != O.seg_chr[s2 - 1]) {
            if (edge_to(s1, s2)) out[0] = s1;
            else if (edge_to(-s1, s2)) out[0] = -s1;
            else return LH_ERR_UNSUPPORTED;                  // (the reference keeps the vertex of the rebuilt graph)
        } else out[0] = path[0] > 0 ? s1 : -s1;
    }
    for (int k = 1; k < P; k++) {                           // (:3877-3900)
        const int seg = file_id(path[k]), prev = out[k - 1];
        if (O.seg_chr[(prev < 0 ? -prev : prev) - 1] != O.seg_chr[seg - 1]) {   // across the translocation: the strand the junction of the file says
            const int to = edge_to(prev, seg);
            if (!to) return LH_ERR_UNSUPPORTED;
            out[k] = to;
        } else if (turn[k]) out[k] = prev > 0 ? -seg : seg;
        else out[k] = prev > 0 ? seg : -seg;
    }
    path = out;
    lines.push_back("TRX-BFB mode: BFB path in the first stage:");
    lines.push_back(format_path(O, path.data(), (int)path.size()));
    // second stage (:3904-3938): the first unused junction one of whose edges starts at a vertex of the path cuts it -- either the
    // head (up to the first occurrence of the other edge's end) or the tail (behind the last occurrence of that start)
    for (int j : T.unused_sv) {
        const int as = a_src_of(O, j), at = a_tgt_of(O, j), bs = -at, bt = -as;
        auto last_of = [&](int v) { for (int i = (int)path.size() - 1; i >= 0; i--) if (path[i] == v) return i; return -1; };
        auto first_of = [&](int v) { for (int i = 0; i < (int)path.size(); i++) if (path[i] == v) return i; return -1; };
        bool edge_a = true;
        int last = last_of(as);
        if (last < 0) { last = last_of(bs); edge_a = false; }
        if (last < 0) continue;
        const int head_end = edge_a ? bt : at, new_head = edge_a ? bs : as, new_tail = edge_a ? at : bt;
        const int first = first_of(head_end);
        const int from_back = (int)path.size() - 1 - last;      // pos1 - rbegin()
        if (first >= 0 && first < from_back) {
            path.erase(path.begin(), path.begin() + first);
            path.insert(path.begin(), new_head);
        } else {
            path.erase(path.begin() + last + 1, path.end());
            path.push_back(new_tail);
        }
        lines.push_back("TRX-BFB mode: BFB path in the second stage:");
        lines.push_back(format_path(O, path.data(), (int)path.size()));
        break;
    }
    return LH_OK;
}

int read_sol(const std::string& path, SolFile& s) {   // localhap.cpp:184-212
    std::ifstream f(path);
    if (!f) return LH_ERR_SOL_OPEN;
    std::string w, v;
    while (f >> w) {
        if (w == "Infeasible") { s.infeasible = true; break; }
        if (w == "value") { double t = 0; f >> t; s.objective += t; }
        if (w[0] == 'x') {
            int x = atoi(w.c_str() + 1);
            // the reference consumes the next token only for x < numComp; epsilon columns come after all element
            // columns in CBC's listing and their value token never starts with 'x', so reading it here is equivalent
            if (!(f >> v)) break;
            s.col.push_back(x); s.val.push_back(atoi(v.c_str()));
        }
    }
    return LH_OK;
}

std::string format_path(const LhGraph& g, const int32_t* path, int len) {   // LGM.cpp:3411-3429
    std::string s;
    auto tok = [](int v) { return std::to_string(v < 0 ? -v : v) + (v > 0 ? "+" : "-"); };
    auto chr = [&](int v) { int id = v < 0 ? -v : v; return (id >= 1 && id <= g.n_seg()) ? g.seg_chr[id - 1] : -2; };
    for (int i = 1; i < len; i++) {
        s += tok(path[i - 1]);
        if (chr(path[i - 1]) != chr(path[i])) s += "||";
        else if ((path[i - 1] > 0) != (path[i] > 0)) s += "|";
    }
    if (len > 0) s += tok(path[len - 1]);
    return s;
}

bool column_to_element(int col, int start_id, int end_id, int* is_loop, int* a, int* b) {
    const int n = end_id - start_id + 1;
    const int num_pat = n * (n + 1) / 2;
    if (col < 0 || col >= 2 * num_pat) return false;
    *is_loop = col >= num_pat;
    int r = col % num_pat;
    int da = 0;
    while (r >= n - da) { r -= n - da; da++; }   // rows of the (a,b) triangle have n, n-1, ... entries
    *a = start_id + da;
    *b = *a + r;
    return true;
}

void merge_out_junctions(std::vector<OutJunction>& out, const int32_t* p, int len, bool increase) {
    for (int i = 0; i + 1 < len; i++) {
        int u = p[i], v = p[i + 1];
        int au = u < 0 ? -u : u, av = v < 0 ? -v : v;
        int d = au - av; if (d < 0) d = -d;
        if (d == 1 && ((u > 0) == (v > 0))) continue;
        bool has = false;
        for (auto& j : out)
            if ((j.u == u && j.v == v) || (j.u == -v && j.v == -u)) { has = true; if (increase) j.count += 1; }
        if (!has) out.push_back({u, v, 1});
    }
}

// ------------------------------------------------------------------------------------------------------
// BFB-TRX (PROP I2 / C2): stitch the per-chromosome paths along the inter-chromosomal junctions.
// LGM.cpp:4052-4193.  Positions are plain indices; an index equal to the container size plays the role of end().
// ------------------------------------------------------------------------------------------------------
void translocation_bfb(const LhGraph& g, std::vector<std::vector<int32_t>>& paths, std::vector<int32_t>& res) {
    auto chr_of = [&](int v) { return g.seg_chr[(v < 0 ? -v : v) - 1]; };
    auto chrom_of = [&](int v) -> const std::string& { return g.seg_chrom[(v < 0 ? -v : v) - 1]; };
    auto vsrc = [&](int j) { return g.j_sdir[j] > 0 ? g.j_src[j] : -g.j_src[j]; };   // edge A source / target
    auto vtgt = [&](int j) { return g.j_tdir[j] > 0 ? g.j_tgt[j] : -g.j_tgt[j]; };
    auto revcomp = [](std::vector<int32_t>& v) { std::reverse(v.begin(), v.end()); for (auto& x : v) x = -x; };
    auto first_of = [](const std::vector<int32_t>& v, long from, int val) -> long {
        for (long i = from < 0 ? 0 : from; i < (long)v.size(); i++) if (v[i] == val) return i;
        return (long)v.size();
    };
    auto last_of = [](const std::vector<int32_t>& v, int val) -> long {
        for (long i = (long)v.size() - 1; i >= 0; i--) if (v[i] == val) return i;
        return -1;
    };
    std::vector<int> sv;
    for (int j = 0; j < g.n_junc(); j++) if (chr_of(g.j_src[j]) != chr_of(g.j_tgt[j])) sv.push_back(j);
    for (auto& p : paths) if (!p.empty() && chrom_of(p[0]) == g.main_chr) res.insert(res.end(), p.begin(), p.end());
    long start_pos = 0;
    while (!sv.empty()) {
        std::vector<int32_t> grp;
        for (size_t i = 0; i < sv.size(); i++) {
            int j = sv[i];
            if (chrom_of(g.j_src[j]) == g.main_chr) { grp = {vsrc(j), vtgt(j)}; sv.erase(sv.begin() + i); break; }
            if (chrom_of(g.j_tgt[j]) == g.main_chr) { grp = {-vtgt(j), -vsrc(j)}; sv.erase(sv.begin() + i); break; }
        }
        if (grp.empty()) break;
        for (long i = 0; i < (long)sv.size(); i++) {
            int j = sv[i];
            if (chr_of(grp.back()) == chr_of(vsrc(j))) { grp.push_back(vsrc(j)); grp.push_back(vtgt(j)); }
            else if (chr_of(grp.back()) == chr_of(-vtgt(j))) { grp.push_back(-vtgt(j)); grp.push_back(-vsrc(j)); }
            else continue;
            sv.erase(sv.begin() + i);
            i = -1;
            if (chrom_of(grp.back()) == g.main_chr) break;
        }
        if (grp.size() == 2) {   // concatenation
            long p1 = last_of(res, grp[0]);
            if (p1 < 0) { revcomp(grp); p1 = last_of(res, grp[0]); }
            if (p1 < 0) continue;
            res.resize(p1 + 1);
            int c = chr_of(grp[1]);
            if (c < 0 || c >= (int)paths.size()) continue;
            std::vector<int32_t>& pp = paths[c];
            long p2 = first_of(pp, 0, grp[1]);
            if (p2 == (long)pp.size()) { revcomp(pp); p2 = first_of(pp, 0, grp[1]); }
            if (p2 == (long)pp.size()) continue;
            res.insert(res.end(), pp.begin() + p2, pp.end());
            start_pos = 0;
        } else {   // insertion
            if (std::abs(grp.front()) > std::abs(grp.back())) revcomp(grp);
            struct Pos { int where; long idx; };   // where: -1 = res, else chromosome index
            std::vector<Pos> pos;
            auto locate = [&]() {
                pos.clear();
                long flag = first_of(res, start_pos, grp[0]);
                pos.push_back({-1, flag});
                if (flag != (long)res.size()) {
                    for (size_t i = 1; i + 1 < grp.size(); i += 2) {
                        int c = chr_of(grp[i]);
                        std::vector<int32_t>& pp = paths[c];
                        long a = first_of(pp, 0, grp[i]);
                        if (a == (long)pp.size()) { revcomp(pp); a = first_of(pp, 0, grp[i]); }
                        if (a == (long)pp.size()) break;
                        pos.push_back({c, a});
                        long b = last_of(pp, grp[i + 1]);
                        if (b < 0 || a > b + 1) { revcomp(pp); b = last_of(pp, grp[i + 1]); }
                        if (b < 0 || a > b + 1) break;
                        pos.push_back({c, b});
                    }
                }
                pos.push_back({-1, first_of(res, flag + 1, grp.back())});
            };
            locate();
            if (pos.size() < grp.size() || pos.back().idx == (long)res.size()) { revcomp(grp); locate(); }
            if (pos.size() < grp.size() || pos.back().idx == (long)res.size()) continue;
            std::vector<int32_t> ins;
            for (size_t i = 1; i + 1 < pos.size(); i += 2) {
                const std::vector<int32_t>& pp = paths[pos[i].where];
                if (pos[i].idx <= pos[i + 1].idx) ins.insert(ins.end(), pp.begin() + pos[i].idx, pp.begin() + pos[i + 1].idx + 1);
            }
            if (ins.empty()) continue;
            long a = pos.front().idx + 1, b = pos.back().idx;
            if (a <= b) res.erase(res.begin() + a, res.begin() + b);
            res.insert(res.begin() + a, ins.begin(), ins.end());
            start_pos = first_of(res, 0, ins.back());
        }
    }
}

}  // namespace ambi
