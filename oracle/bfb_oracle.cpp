// oracle/bfb_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see bfb_oracle.hpp header comment).
// CPU restatement of the reference's `--op bfb` path; every function cites the reference
// file:line it follows.  Never linked into the product library.
#include "bfb_oracle.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <set>
#include <sstream>

namespace oracle {

// ------------------------------------------------------------------------------------------
// Graph model  (Graph.cpp / Segment.cpp / Junction.cpp)
// ------------------------------------------------------------------------------------------
Seg* Graph::segById(int id) {   // Graph.cpp:513-520
    for (auto& s : segs)
        if (s.id == id) return &s;
    return nullptr;               // the reference throws SegmentDoesNotExistException (uncaught -> abort)
}
bool Graph::hasSeg(int id) const {
    for (auto& s : segs)
        if (s.id == id) return true;
    return false;
}

static bool sameJunction(const Junc& x, const Junc& y) {
    // Graph.cpp:489-499: compares the "id dir=>id dir" strings of edges A and B.
    return (x.a_src() == y.a_src() && x.a_tgt() == y.a_tgt() && x.b_src() == y.b_src() && x.b_tgt() == y.b_tgt()) ||
           (x.a_src() == y.b_src() && x.a_tgt() == y.b_tgt() && x.b_src() == y.a_src() && x.b_tgt() == y.a_tgt());
}

int findJunction(const Graph& g, int src, char sdir, int tgt, char tdir) {   // Graph.cpp:501-511
    Junc q; q.src = src; q.tgt = tgt; q.sdir = sdir; q.tdir = tdir;
    for (size_t i = 0; i < g.juncs.size(); i++)
        if (sameJunction(g.juncs[i], q)) return (int)i;
    return -1;
}

bool addJunction(Graph& g, int src, char sdir, int tgt, char tdir, double cov, double cn, bool inferred, bool bounded) {
    // Graph.cpp:579-610.  getSegmentById throws for unknown ids -> report failure.
    if (!g.hasSeg(src) || !g.hasSeg(tgt)) return false;
    if (findJunction(g, src, sdir, tgt, tdir) >= 0) return true;   // Graph.cpp:592-595: silently ignored
    Junc j; j.src = src; j.tgt = tgt; j.sdir = sdir; j.tdir = tdir; j.cov = cov; j.cn = cn;
    j.inferred = inferred; j.bounded = bounded;
    g.juncs.push_back(j);
    return true;
}

// strtok-driven line parser, Graph.cpp:109-237.  The real libc strtok/atof/atoi are used so the
// token semantics (space-only delimiters after the first token, nested re-tokenising) are inherited.
namespace {
struct Crash { bool hit = false; };
char* tok(char* s, const char* d, Crash& c) {
    char* t = strtok(s, d);
    if (!t) c.hit = true;
    return t;
}
}  // namespace

bool readGraph(const std::string& path, Graph& g, std::string& err) {
    std::ifstream f(path);
    if (!f) { err = "Cannot open file " + path; return false; }   // Graph.cpp:111-114 (exit 1)
    g.log.push_back("Reading graph...");
    std::vector<char> buf(8192);
    char* line = buf.data();
    Crash c;
    while (!f.eof()) {
        f.getline(line, 8192);
        if (f.fail() && !f.eof()) { err = "line longer than 8191 bytes (reference loops forever)"; return false; }
        char* p = line;
        while (*p != '\0') { if (*p != '\t' && *p != ' ') break; p++; }
        if (*p == '#') continue;
        char* token = strtok(line, " \t");
        if (token == NULL) continue;
        if (strcmp(token, "SAMPLE_NAME") == 0) {
            char* t = tok(NULL, " ", c); if (c.hit) break; g.sampleName = t;
        } else if (strcmp(token, "AVG_CHR_SEG_DP") == 0) {
            token = strtok(NULL, " "); token = strtok(token, ",");
            while (token != NULL) { g.avgCoverages.push_back(atof(token)); token = strtok(NULL, ","); }
        } else if (strcmp(token, "AVG_WHOLE_HOST_DP") == 0) {
            char* t = tok(NULL, " ", c); if (c.hit) break; g.avgCoverageRaw = atof(t);
        } else if (strcmp(token, "AVG_VIRUS_SEG_DP") == 0) {
            char* t = tok(NULL, " ", c); if (c.hit) break; g.avgVirusDP = atof(t);
        } else if (strcmp(token, "VIRUS_START") == 0) {
            char* t = tok(NULL, " ", c); if (c.hit) break; g.virusSegStart = atoi(t); g.virusSegStartSet = true;
        } else if (strcmp(token, "AVG_JUNC_DP") == 0) {
            char* t = tok(NULL, " ", c); if (c.hit) break; g.avgCoverageJunc = atof(t);
        } else if (strcmp(token, "PURITY") == 0) {
            char* t = tok(NULL, " ", c); if (c.hit) break; g.purity = atof(t);
        } else if (strcmp(token, "AVG_TUMOR_PLOIDY") == 0) {
            char* t = tok(NULL, " ", c); if (c.hit) break; g.avgTumorPloidy = atof(t);
        } else if (strcmp(token, "AVG_PLOIDY") == 0) {
            char* t = tok(NULL, " ", c); if (c.hit) break; g.avgPloidy = atof(t);
        } else if (strcmp(token, "PLOIDY") == 0) {
            char* t = tok(NULL, " ", c); if (c.hit) break;
            g.ploidy = t;
            char* m = tok(t, "m", c); if (c.hit) break; g.expectedPloidy = atoi(m);
        } else if (strcmp(token, "SOURCE") == 0) {
            token = strtok(NULL, " "); token = strtok(token, ",");
            while (token != NULL) { g.sourceIds.push_back(atoi(token)); token = strtok(NULL, ","); }
        } else if (strcmp(token, "SINK") == 0) {
            token = strtok(NULL, " "); token = strtok(token, ",");
            while (token != NULL) { g.sinkIds.push_back(atoi(token)); token = strtok(NULL, ","); }
        } else if (strcmp(token, "SEG") == 0) {
            char* node = tok(NULL, " ", c);
            char* t1 = tok(NULL, " ", c);
            char* t2 = tok(NULL, " ", c);
            if (c.hit) break;
            double segCoverage = std::max(atof(t1), 0.0);
            double segCopy = atof(t2);
            tok(node, ":", c);
            char* a = tok(NULL, ":", c); char* b = tok(NULL, ":", c); char* s = tok(NULL, ":", c); char* e = tok(NULL, ":", c);
            if (c.hit) break;
            Seg sg; sg.id = atoi(a); sg.chrom = b; sg.start = atoi(s); sg.end = atoi(e);
            sg.cov = segCoverage; sg.cn = segCopy;
            sg.chrId = -1;   // uninitialised in the reference when no SOURCE/SINK range matches (Graph.cpp:193-197)
            for (size_t i = 0; i < g.sourceIds.size(); i++)
                if (i < g.sinkIds.size() && g.sourceIds[i] <= sg.id && sg.id <= g.sinkIds[i]) sg.chrId = (int)i;
            g.segs.push_back(sg);
        } else if (strcmp(token, "JUNC") == 0) {
            char* sN = tok(NULL, " ", c); char* tN = tok(NULL, " ", c);
            char* t1 = tok(NULL, " ", c); char* t2 = tok(NULL, " ", c);
            char* t3 = tok(NULL, " ", c); char* t4 = tok(NULL, " ", c);
            if (c.hit) break;
            double junCoverage = atof(t1), junCopy = atof(t2);
            bool isInferred = (t3[0] == 'I'), isBounded = (t4[0] == 'B');
            if (junCoverage <= 0 && junCopy <= 0) continue;
            tok(sN, ":", c); char* a = tok(NULL, ":", c); char* ad = tok(NULL, ":", c);
            if (c.hit) break;
            int sourceId = atoi(a); char sourceDir = ad[0];
            tok(tN, ":", c); char* b = tok(NULL, ":", c); char* bd = tok(NULL, ":", c);
            if (c.hit) break;
            int targetId = atoi(b); char targetDir = bd[0];
            if (!addJunction(g, sourceId, sourceDir, targetId, targetDir, junCoverage, junCopy, isInferred, isBounded)) {
                err = "JUNC references unknown segment (reference aborts: SegmentDoesNotExistException)";
                return false;
            }
        }
    }
    if (c.hit) { err = "malformed line (reference dereferences a NULL strtok result)"; return false; }
    if (g.sourceIds.size() != g.sinkIds.size()) { err = "SOURCE/SINK size mismatch (Graph.cpp:227 assert)"; return false; }
    for (size_t i = 0; i < g.sourceIds.size(); i++)
        if (!g.hasSeg(g.sourceIds[i]) || !g.hasSeg(g.sinkIds[i])) { err = "SOURCE/SINK id without SEG"; return false; }
    return true;
}

bool calculateHapDepth(Graph& g, std::string& err) {   // Graph.cpp:312-367
    if (g.avgPloidy < 0) {
        if (g.avgTumorPloidy < 0) { err = "input error: there is no ploidy information provided."; return false; }
        if (g.purity < 0) { err = "input error: no purity information provided."; return false; }
        g.avgPloidy = g.purity * g.avgTumorPloidy + (1 - g.purity) * 2;
    } else {
        if (g.avgTumorPloidy >= 0) {
            if (g.purity < 0) {
                g.log.push_back("WARN: no purity information provided, use the given AVG_PLOIDY");
            } else {
                double ratio = 1 - (g.purity * g.avgTumorPloidy) / ((g.purity * g.avgTumorPloidy) + (1 - g.purity) * 2);
                double avgPloidy = g.purity * g.avgTumorPloidy + (1 - g.purity) * 2;
                g.ratio = ratio; g.ratioSet = true;
                if (std::abs(g.avgPloidy - avgPloidy) <= 0.1) { /* keep */ } else g.avgPloidy = avgPloidy;
            }
        } else {
            g.log.push_back("WARN: only AVG_PLOIDY is given, use that");
        }
    }
    g.haploidDepth = g.avgCoverageRaw * g.purity / g.avgPloidy;
    g.avgCoverage = g.avgPloidy * g.haploidDepth;
    g.avgCoverageJunc = g.avgPloidy * g.haploidDepth;
    return true;
}

static std::string fmtDouble(double v) { std::ostringstream os; os << v; return os.str(); }  // cout << double

void calculateCopyNum(Graph& g) {   // Graph.cpp:369-405
    double ratio = g.ratio, hDP = g.haploidDepth;
    for (auto& seg : g.segs) {
        if (seg.cn > 0) continue;
        double segCopy;
        if (seg.id >= g.virusSegStart) segCopy = seg.cov / g.avgCoverageRaw * 2;
        else { double depthT = seg.cov - g.avgCoverageRaw * ratio; segCopy = depthT / hDP; }
        seg.cn = std::max(segCopy, 0.0);
        g.log.push_back("SEG" + std::to_string(seg.id) + " " + fmtDouble(seg.cov) + " " + fmtDouble(seg.cn));
    }
    for (auto& j : g.juncs) {
        if (j.cn > 0) continue;
        if (j.inferred) g.log.push_back(fmtDouble(g.haploidDepth));
        double depthT = j.cov - g.avgCoverageRaw * ratio;
        j.cn = std::max(depthT / hDP, 0.0);
    }
}

void readBFBProps(const std::string& lhPath, Props& p) {   // LGM.cpp:3941-3987
    std::ifstream lhFile(lhPath);
    std::string line, prop;
    while (getline(lhFile, line)) {
        std::stringstream ss(line);
        prop.clear();
        ss >> prop;
        if (prop == "PROP") {
            while (ss >> prop) {
                size_t pos = 2, lastPos = 2;
                if (prop[0] == 'M') p.mainChr = prop.substr(2);
                else if (prop[0] == 'I') {
                    if (prop[1] != ':') { p.insMode = prop[1] - '0'; lastPos = 3; } else p.insMode = 2;
                    while (pos != std::string::npos) {
                        pos = prop.find(":", lastPos);
                        p.insChr.push_back(prop.substr(lastPos, pos - lastPos));
                        lastPos = pos + 1;
                    }
                } else if (prop[0] == 'C') {
                    if (prop[1] != ':') { p.conMode = prop[1] - '0'; lastPos = 3; } else p.conMode = 2;
                    while (pos != std::string::npos) {
                        pos = prop.find(":", lastPos);
                        p.conChr.push_back(prop.substr(lastPos, pos - lastPos));
                        lastPos = pos + 1;
                    }
                } else if (prop[0] == 'S') {
                    while (pos != std::string::npos) {
                        pos = prop.find(":", lastPos);
                        p.startSegs.push_back(stoi(prop.substr(lastPos, pos - lastPos)));
                        lastPos = pos + 1;
                    }
                }
            }
        }
    }
}

void readComponents(Graph& g, const std::string& juncsPath, std::vector<std::vector<int>>& res,
                    std::vector<std::string>& log) {   // LGM.cpp:5096-5156
    if (juncsPath.empty()) return;
    std::ifstream inFile(juncsPath);
    std::string line;
    while (getline(inFile, line)) {
        std::istringstream iss(line);
        std::vector<int> segs; std::vector<char> sign; std::string temp;
        while (iss >> temp) {
            segs.push_back(stoi(temp.substr(0, temp.length() - 1)));
            sign.push_back(temp.back());
        }
        size_t lastIdx = 0;
        for (size_t i = 1; i < segs.size(); i++) {
            Seg* a = g.segById(segs[lastIdx]); Seg* b = g.segById(segs[i]);
            if (!a || !b) return;   // reference aborts
            if (a->partition != b->partition || sign[i - 1] != sign[i]) {
                if (i - lastIdx >= 2) {
                    std::vector<int> subset(segs.begin() + lastIdx, segs.begin() + i);
                    std::sort(subset.begin(), subset.end());
                    res.push_back(subset);
                }
                int sourceId = segs[i - 1], targetId = segs[i];
                char sourceDir = sign[i - 1], targetDir = sign[i];
                log.push_back(std::to_string(sourceId) + sourceDir + " -> " + std::to_string(targetId) + targetDir);
                int j2 = findJunction(g, sourceId, sourceDir, targetId, targetDir);
                if (j2 < 0) addJunction(g, sourceId, sourceDir, targetId, targetDir, g.avgCoverage, 1, false, true);
                else if (g.juncs[j2].cn < 2) g.juncs[j2].cn = 2;
                lastIdx = i;
            }
        }
        if (segs.size() - lastIdx >= 2 && segs.size() >= lastIdx) {
            std::vector<int> subset(segs.begin() + lastIdx, segs.end());
            std::sort(subset.begin(), subset.end());
            res.push_back(subset);
        }
    }
    std::sort(res.begin(), res.end());
    res.erase(std::unique(res.begin(), res.end()), res.end());
}

// ------------------------------------------------------------------------------------------
// Per-chromosome pre-ILP scans
// ------------------------------------------------------------------------------------------
void getJuncCN(const Graph& g, int startSegID, int endSegID, Inversions& inversions, std::vector<double>& juncCN) {
    // LGM.cpp:3989-4050
    juncCN.assign((size_t)(endSegID + 1) * 2, 0.0);
    std::vector<int> inv;
    for (size_t ji = 0; ji < g.juncs.size(); ji++) {
        const Junc& junc = g.juncs[ji];
        int sourceID = junc.src, targetID = junc.tgt;
        if (sourceID < startSegID || sourceID > endSegID || targetID < startSegID || targetID > endSegID) continue;
        double copyNum = junc.cn;
        if (0.5 < copyNum && copyNum < 1) copyNum = 1;
        if (junc.sdir == junc.tdir) {
            if (sourceID + 1 == targetID) juncCN[sourceID * 2 + 0] += copyNum;
            else if (sourceID - 1 == targetID) juncCN[targetID * 2 + 0] += copyNum;
        } else {
            if (std::abs(sourceID - targetID) <= 2) {
                inv.push_back((int)ji);
                if (inversions.find(sourceID) == inversions.end()) {
                    inversions[sourceID] = (int)ji; juncCN[sourceID * 2 + 1] += copyNum;
                } else if (inversions.find(targetID) == inversions.end()) {
                    inversions[targetID] = (int)ji; juncCN[targetID * 2 + 1] += copyNum;
                }
            }
        }
    }
    for (int ji : inv) {
        int sourceID = g.juncs[ji].src, targetID = g.juncs[ji].tgt;
        if (inversions.find(sourceID) == inversions.end()) inversions[sourceID] = ji;
        if (inversions.find(targetID) == inversions.end()) inversions[targetID] = ji;
    }
}

int computeBias(const Graph& g, int startID, int endID, const Inversions& inv, const std::vector<double>& juncCN) {
    int bias = 1;   // localhap.cpp:141-146
    for (int i = startID; i <= endID; i++) {
        if (juncCN[i * 2 + 1] > 0) {
            auto it = inv.find(i);
            if (it != inv.end() && g.juncs[it->second].src != g.juncs[it->second].tgt) bias += int(juncCN[i * 2 + 1]) % 2;
        }
    }
    return bias;
}

void getIndelBias(Graph& g, int startSegID, int endSegID) {   // LGM.cpp:3699-3744
    std::vector<int> sv;
    for (size_t ji = 0; ji < g.juncs.size(); ji++) {
        const Junc& junc = g.juncs[ji];
        const Seg* s = g.segById(junc.src); const Seg* t = g.segById(junc.tgt);
        if (s->chrId != t->chrId) continue;
        int sourceID = junc.src, targetID = junc.tgt;
        char sourceDir = junc.sdir, targetDir = junc.tdir;
        if (sourceID < startSegID || sourceID > endSegID || targetID < startSegID || targetID > endSegID) continue;
        if (sourceDir != targetDir) continue;
        if ((sourceDir == '+' && targetID - sourceID == 1) || (sourceDir == '-' && sourceID - targetID == 1)) continue;
        sv.push_back((int)ji);
    }
    while (!sv.empty()) {
        std::vector<int> group;
        for (int i = 0; i < (int)sv.size(); i++) {
            const Junc& j = g.juncs[sv[i]];
            int sourceID = j.src, targetID = j.tgt;
            if (j.sdir == '-') sourceID = -sourceID;
            if (j.tdir == '-') targetID = -targetID;
            if (group.empty()) { group.push_back(sourceID); group.push_back(targetID); }
            else {
                if (targetID == group.front()) group.insert(group.begin(), sourceID);
                else if (sourceID == -group.front()) group.insert(group.begin(), -targetID);
                else if (group.back() == sourceID) group.push_back(targetID);
                else if (group.back() == -targetID) group.push_back(-sourceID);
                else continue;
            }
            sv.erase(sv.begin() + i);
            i--;
        }
        if (group.size() == 2) {
            if (group[0] < group[1]) { for (int j = group[0] + 1; j < group[1]; j++) g.segs[std::abs(j) - 1].cn += 1; }
            else { for (int j = group[1]; j <= group[0]; j++) g.segs[std::abs(j) - 1].cn -= 1; }
        } else {
            for (size_t j = 1; j + 1 < group.size(); j++) g.segs[std::abs(group[j]) - 1].cn -= 1;
        }
    }
}

std::map<std::string, int> makeVariableIdx(int startID, int endID, int* numPatOut) {
    // LGM.cpp:3254-3264 (combinations with len=2: all a<=b in lexicographic order) + localhap.cpp:122-133
    std::map<std::string, int> variableIdx;
    int idx = 0;
    std::vector<std::pair<int, int>> pats;
    for (int a = startID; a <= endID; a++)
        for (int b = a; b <= endID; b++) pats.push_back({a, b});
    int numPat = (int)pats.size();
    for (auto& p : pats) variableIdx["p:" + std::to_string(p.first) + "," + std::to_string(p.second)] = idx++;
    idx = 0;
    for (auto& p : pats) variableIdx["l:" + std::to_string(p.first) + "," + std::to_string(p.second)] = numPat + idx++;
    if (numPatOut) *numPatOut = numPat;
    return variableIdx;
}

// ------------------------------------------------------------------------------------------
// DAG + orders
// ------------------------------------------------------------------------------------------
static bool compareLoops(std::vector<int> a, std::vector<int> b) {   // LGM.cpp:3267-3274
    int diff1 = 0, diff2 = 0;
    if (a.size() > 0 && b.size() > 0) { diff1 = std::abs(a[0] - a[1]); diff2 = std::abs(b[0] - b[1]); }
    return (diff1 > diff2);
}

void constructDAG(const std::map<std::string, int>& variableIdx, const std::vector<int>& elementCN, Dag& dag) {
    // LGM.cpp:3276-3378
    auto& adj = dag.adj; auto& node2pat = dag.node2pat; auto& node2loop = dag.node2loop;
    std::vector<std::vector<int>> parents;
    for (auto iter = variableIdx.begin(); iter != variableIdx.end(); iter++) {
        if (elementCN[iter->second] > 0) {
            std::vector<int> temp;
            adj.push_back(temp); parents.push_back(temp);
            std::string key = iter->first;
            temp.push_back(stoi(key.substr(2, key.find(",") - 2)));
            temp.push_back(stoi(key.substr(key.find(",") + 1)));
            temp.push_back(elementCN[iter->second]);
            if (key[0] == 'p') { node2pat.push_back(temp); temp.clear(); node2loop.push_back(temp); }
            else { node2loop.push_back(temp); temp.clear(); node2pat.push_back(temp); }
        }
    }
    std::sort(node2loop.begin(), node2loop.end(), compareLoops);   // LGM.cpp:3303 (the real libstdc++ sort)
    for (size_t i = 0; i < node2pat.size(); i++) {
        if (node2pat[i].size() > 0) {
            for (size_t j = 0; j < node2pat.size(); j++) {
                if (node2pat[j].size() > 0 && (node2pat[i][0] == node2pat[j][0] || node2pat[i][1] == node2pat[j][1])) {
                    int diff1 = node2pat[i][0] - node2pat[i][1], diff2 = node2pat[j][0] - node2pat[j][1];
                    if (std::abs(diff1) > std::abs(diff2)) { adj[i].push_back((int)j); parents[j].push_back((int)i); }
                }
            }
            for (size_t j = 0; j < node2loop.size(); j++) {
                if (node2loop[j].size() > 0 && (node2pat[i][0] == node2loop[j][0] || node2pat[i][1] == node2loop[j][1])) {
                    int diff1 = node2pat[i][0] - node2pat[i][1], diff2 = node2loop[j][0] - node2loop[j][1];
                    if (std::abs(diff1) > std::abs(diff2)) { adj[i].push_back((int)j); parents[j].push_back((int)i); }
                }
            }
        }
    }
    for (size_t i = 0; i < node2loop.size(); i++) {
        if (node2loop[i].size() > 0) {
            for (size_t j = 0; j < node2pat.size(); j++) {
                if (std::find(parents[i].begin(), parents[i].end(), (int)j) != parents[i].end()) continue;
                if (node2pat[j].size() > 0 && (node2loop[i][0] == node2pat[j][0] || node2loop[i][1] == node2pat[j][1])) {
                    int diff1 = node2loop[i][0] - node2loop[i][1], diff2 = node2pat[j][0] - node2pat[j][1];
                    if (std::abs(diff1) > std::abs(diff2)) { adj[i].push_back((int)j); parents[j].push_back((int)i); }
                    else {
                        for (int parent : parents[i]) {
                            if (std::find(adj[parent].begin(), adj[parent].end(), (int)j) != adj[parent].end()) {
                                adj[i].push_back((int)j); parents[j].push_back((int)i);
                                break;
                            }
                        }
                    }
                }
            }
            for (size_t j = 0; j < node2loop.size(); j++) {
                if (node2loop[j].size() > 0 && (node2loop[i][0] == node2loop[j][0] || node2loop[i][1] == node2loop[j][1])) {
                    int diff1 = node2loop[i][0] - node2loop[i][1], diff2 = node2loop[j][0] - node2loop[j][1];
                    if (std::abs(diff1) > std::abs(diff2)) { adj[i].push_back((int)j); parents[j].push_back((int)i); }
                }
            }
        }
    }
}

namespace {
struct TopoCtx {
    const std::vector<std::vector<int>>* adj; std::vector<char> visited; std::vector<int> indeg, res;
    std::vector<std::vector<int>>* orders; size_t maxOrders; int num;
};
void topoRec(TopoCtx& c) {   // LGM.cpp:3380-3409
    if (c.orders->size() >= c.maxOrders) return;
    if ((int)c.res.size() == c.num) c.orders->push_back(c.res);
    for (int i = 0; i < (int)c.adj->size(); i++) {
        if (c.indeg[i] == 0 && !c.visited[i]) {
            for (int j : (*c.adj)[i]) c.indeg[j]--;
            c.res.push_back(i); c.visited[i] = 1;
            topoRec(c);
            c.visited[i] = 0; c.res.pop_back();
            for (int j : (*c.adj)[i]) c.indeg[j]++;
        }
    }
}
}  // namespace

void allTopologicalOrders(const Dag& dag, std::vector<std::vector<int>>& orders, size_t maxOrders) {
    TopoCtx c; c.adj = &dag.adj; c.num = (int)dag.adj.size();
    c.visited.assign(c.num, 0); c.indeg.assign(c.num, 0); c.orders = &orders; c.maxOrders = maxOrders;
    for (int i = 0; i < c.num; i++)   // localhap.cpp:246-250
        for (int nx : dag.adj[i]) c.indeg[nx]++;
    topoRec(c);
}

// ------------------------------------------------------------------------------------------
// getBFB / imperfectFBI
// ------------------------------------------------------------------------------------------
static inline int vid(int v) { return std::abs(v); }
static inline bool plus(int v) { return v > 0; }

void imperfectFBI(const Graph& g, std::vector<int>& bkp, const Inversions& inversions, bool* ub) {
    // LGM.cpp:3431-3512.  Indices replace iterators; find(pos+3,end,..) with pos+3 past end returns end in
    // libstdc++ (negative trip count, switch default) -- made explicit here.
    const long L = (long)bkp.size();
    long pos = 0;
    while (pos < L) {
        if (pos + 1 >= L) { if (ub) *ub = true; return; }   // reference reads *(pos+1) out of bounds
        long r = L;
        if (pos + 3 < L) {
            int want = -bkp[pos];
            for (long q = pos + 3; q < L; q++) if (bkp[q] == want) { r = q; break; }
        }
        long l = r - 1;
        if (r == L || bkp[l] != -bkp[pos + 1]) {
            int id = vid(bkp[pos + 1]);
            auto it = inversions.find(id);
            if (it != inversions.end()) {
                const Junc& J = g.juncs[it->second];
                if (plus(bkp[pos + 1])) bkp[pos + 1] = (J.src < J.tgt) ? J.src : J.tgt;
                else bkp[pos + 1] = (J.src < J.tgt) ? -J.tgt : -J.src;
            }
            if (pos > 0) {
                id = vid(bkp[pos]);
                it = inversions.find(id);
                if (it != inversions.end() && vid(bkp[pos - 1]) == id) {
                    const Junc& J = g.juncs[it->second];
                    int other = (J.src == id) ? J.tgt : J.src;
                    bkp[pos] = plus(bkp[pos]) ? other : -other;
                }
            }
            if (plus(bkp[pos]) && vid(bkp[pos]) > vid(bkp[pos + 1])) bkp[pos + 1] = bkp[pos];
            if (!plus(bkp[pos]) && vid(bkp[pos]) < vid(bkp[pos + 1])) bkp[pos + 1] = bkp[pos];
            pos += 2;
        } else {
            long p1 = pos + ((l - pos) / 2), p2 = p1 + 1;
            while (p1 >= pos - 1 && p1 > 0) {
                int id = vid(bkp[p1]);
                auto it = inversions.find(id);
                if (it != inversions.end()) {
                    const Junc& J = g.juncs[it->second];
                    if (p1 + 1 >= L) { if (ub) *ub = true; return; }
                    if (plus(bkp[p1])) {
                        if (J.src < J.tgt) { bkp[p1] = J.src; bkp[p1 + 1] = -J.tgt; }
                        else { bkp[p1] = J.tgt; bkp[p1 + 1] = -J.src; }
                    } else {
                        if (J.src < J.tgt) { bkp[p1] = -J.tgt; bkp[p1 + 1] = J.src; }
                        else { bkp[p1] = -J.src; bkp[p1 + 1] = J.tgt; }
                    }
                    if (p2 != p1 + 1) {
                        if (p1 > pos - 1) { if (p2 >= L) { if (ub) *ub = true; return; } bkp[p2] = -bkp[p1]; }
                        bkp[p2 - 1] = -bkp[p1 + 1];
                    }
                }
                p1 -= 2; p2 += 2;
            }
            pos = r + 1;
        }
    }
}

bool evalOrder(const Graph& g, const std::vector<int>& bfb, const Dag& dag, const Inversions& inversions,
               bool forwardDir, std::vector<int>& bkpPath, bool* ub, bool* crash) {
    // body of the per-order loop, LGM.cpp:3519-3658
    const auto& node2pat = dag.node2pat; const auto& node2loop = dag.node2loop;
    bkpPath.clear();
    if (bfb.empty()) { if (crash) *crash = true; return false; }   // bfb[0] on an empty order
    int start, end;
    if (node2pat[bfb[0]].size()) { start = node2pat[bfb[0]][0]; end = node2pat[bfb[0]][1]; }
    else if (node2loop[bfb[0]].size()) { start = node2loop[bfb[0]][0]; end = node2loop[bfb[0]][1]; }
    else { if (crash) *crash = true; return false; }   // node2loop[bfb[0]][0] on an EMPTY vector<int> (never allocated: a null dereference)
    if (forwardDir) {
        if (node2pat[bfb[0]].size()) { bkpPath.push_back(start); bkpPath.push_back(end); }
        else {
            int cn = node2loop[bfb[0]][2], num = 0;
            while (num < cn) { bkpPath.push_back(start); bkpPath.push_back(end); bkpPath.push_back(-end); bkpPath.push_back(-start); num++; }
        }
    } else {
        if (node2pat[bfb[0]].size()) { bkpPath.push_back(-end); bkpPath.push_back(-start); }
        else {
            int cn = node2loop[bfb[0]][2], num = 0;
            while (num < cn) { bkpPath.push_back(-end); bkpPath.push_back(-start); bkpPath.push_back(start); bkpPath.push_back(end); num++; }
        }
    }
    size_t i;
    for (i = 1; i < bfb.size(); i++) {
        if (node2pat[bfb[i]].size()) {
            start = node2pat[bfb[i]][0]; end = node2pat[bfb[i]][1];
            if (bkpPath.empty()) { if (crash) *crash = true; return false; }
            if (bkpPath.back() == -start) { bkpPath.push_back(start); bkpPath.push_back(end); }
            else if (bkpPath.back() == end) { bkpPath.push_back(-end); bkpPath.push_back(-start); }
            else break;
        } else if (node2loop[bfb[i]].size()) {
            start = node2loop[bfb[i]][0]; end = node2loop[bfb[i]][1];
            int v1 = -start, v2 = end;
            auto pos = std::find(bkpPath.rbegin(), bkpPath.rend(), v1);
            while (pos != bkpPath.rend() && ((bkpPath.rend() - pos) % 2 == 1 ||
                   (pos - bkpPath.rbegin() > 1 && vid(*(pos + 1)) < vid(*(pos - 2))))) {
                pos = std::find(pos + 1, bkpPath.rend(), v1);
            }
            if (pos == bkpPath.rend()) {
                pos = std::find(bkpPath.rbegin(), bkpPath.rend(), v2);
                while (pos != bkpPath.rend() && ((bkpPath.rend() - pos) % 2 == 1 ||
                       (pos - bkpPath.rbegin() > 1 && vid(*(pos + 1)) > vid(*(pos - 2))))) {
                    pos = std::find(pos + 1, bkpPath.rend(), v2);
                }
            }
            if (pos == bkpPath.rend()) break;
            int cn = node2loop[bfb[i]][2], num = 0;
            std::vector<int> loop;
            if (*pos == v1) {
                while (num < cn) { loop.push_back(start); loop.push_back(end); loop.push_back(-end); loop.push_back(-start); num++; }
                auto temp = pos.base() - 1;
                *temp = -start;
                if (temp + 1 != bkpPath.end()) *(temp + 1) = start;
            } else {
                while (num < cn) { loop.push_back(-end); loop.push_back(-start); loop.push_back(start); loop.push_back(end); num++; }
                auto temp = pos.base() - 1;
                *temp = end;
                if (temp + 1 != bkpPath.end()) *(temp + 1) = -end;
            }
            size_t at = pos.base() - bkpPath.begin();
            bkpPath.insert(bkpPath.begin() + at, loop.begin(), loop.end());
        }
    }
    imperfectFBI(g, bkpPath, inversions, ub);   // LGM.cpp:3656 (always, before the validity test)
    return i == bfb.size();
}

void expandBkp(const std::vector<int>& bkp, std::vector<int>& path) {   // LGM.cpp:3661-3670
    for (size_t j = 1; j < bkp.size(); j += 2) {
        if (plus(bkp[j - 1])) { for (int k = vid(bkp[j - 1]); k <= vid(bkp[j]); k++) path.push_back(k); }
        else { for (int k = vid(bkp[j - 1]); k >= vid(bkp[j]); k--) path.push_back(-k); }
    }
}

std::string formatPath(const Graph& g, const std::vector<int>& path) {   // LGM.cpp:3411-3429
    std::string s;
    auto info = [](int v) { return std::to_string(std::abs(v)) + (v > 0 ? "+" : "-"); };
    auto chr = [&](int v) {   // ids are 1..N in file order on every supported input; fall back to the scan otherwise
        int id = std::abs(v);
        if (id >= 1 && id <= (int)g.segs.size() && g.segs[id - 1].id == id) return g.segs[id - 1].chrId;
        for (auto& sg : g.segs) if (sg.id == id) return sg.chrId;
        return -2;
    };
    for (size_t i = 1; i < path.size(); i++) {
        s += info(path[i - 1]);
        if (chr(path[i - 1]) != chr(path[i])) s += "||";
        else if (plus(path[i - 1]) != plus(path[i])) s += "|";
    }
    if (!path.empty()) s += info(path.back());   // reference: back() on empty path is UB (segfault)
    return s;
}

void getBFB(const Graph& g, const std::vector<std::vector<int>>& orders, const Dag& dag, const Inversions& inv,
            bool isReversed, bool printAll, BfbResult& res, std::vector<std::string>& log) {
    // LGM.cpp:3514-3697
    bool forwardDir = !isReversed;
    std::vector<int> bkpPath;
    for (long n = 0; n < (long)orders.size(); n++) {
        bool ub = false, crash = false;
        bool valid = evalOrder(g, orders[n], dag, inv, forwardDir, bkpPath, &ub, &crash);
        res.evaluated++;
        if (ub || crash) res.undefinedBehaviour = true;
        // Two kinds of undefined behaviour in the reference.  (a) a null dereference (empty order, node with both slots
        // empty at bfb[0]): it crashes there, nothing after this order is ever printed.  (b) imperfectFBI touching the cell
        // behind the end of the breakpoint vector: on an order that is invalid anyway the outcome is "invalid" whatever
        // the stray cell holds (validity was fixed before imperfectFBI ran and the breakpoints are discarded), so the scan
        // goes on; on a VALID order the printed path would depend on the stray cell.  (a) and (b)-on-valid have no
        // defined result: the scan stops here and the unit counts as refused.
        if (crash || (ub && valid)) { res.undefinedOnValid = true; break; }
        if (valid) {
            if (res.path.empty()) {
                expandBkp(bkpPath, res.path);
                res.bkpFirst = bkpPath; res.firstValidOrder = n; res.firstValidOrientationForward = forwardDir ? 1 : 0;
            }
            if (printAll) {
                std::vector<int> temp; expandBkp(bkpPath, temp);
                log.push_back(formatPath(g, temp));
                res.allPaths.push_back(temp);
                res.allEvalIdx.push_back(res.evaluated - 1);
            } else {
                log.push_back(formatPath(g, res.path));
                break;
            }
        } else if (n == (long)orders.size() - 1 && forwardDir != isReversed) {
            n = -1; forwardDir = isReversed;
        }
    }
}

// ------------------------------------------------------------------------------------------
// indelBFB
// ------------------------------------------------------------------------------------------
bool indelBFB(const Graph& gc, std::vector<int>& path, int startSegID, int endSegID, std::vector<std::string>& log) {
    // LGM.cpp:3746-3837
    Graph& g = const_cast<Graph&>(gc);
    std::vector<int> sv;
    for (size_t ji = 0; ji < g.juncs.size(); ji++) {
        const Junc& junc = g.juncs[ji];
        if (g.segById(junc.src)->chrId != g.segById(junc.tgt)->chrId) continue;
        int sourceID = junc.src, targetID = junc.tgt;
        char sourceDir = junc.sdir, targetDir = junc.tdir;
        if (sourceID < startSegID || sourceID > endSegID || targetID < startSegID || targetID > endSegID) continue;
        if (sourceDir != targetDir && std::abs(sourceID - targetID) <= 2) continue;
        if (sourceDir == targetDir && ((sourceDir == '+' && targetID - sourceID == 1) || (sourceDir == '-' && sourceID - targetID == 1))) continue;
        sv.push_back((int)ji);
    }
    if (sv.empty()) return false;
    auto complementAll = [](std::vector<int>& grp) { std::reverse(grp.begin(), grp.end()); for (auto& v : grp) v = -v; };
    while (!sv.empty()) {
        std::vector<int> group;
        for (int i = 0; i < (int)sv.size(); i++) {
            const Junc& J = g.juncs[sv[i]];
            if (group.empty()) { group.push_back(J.a_src()); group.push_back(J.a_tgt()); }
            else {
                if (J.a_tgt() == group.front()) group.insert(group.begin(), J.a_src());
                else if (J.b_tgt() == group.front()) group.insert(group.begin(), J.b_src());
                else if (group.back() == J.a_src()) group.push_back(J.a_tgt());
                else if (group.back() == J.b_src()) group.push_back(J.b_tgt());
                else continue;
            }
            sv.erase(sv.begin() + i);
            i--;
        }
        if (group.size() == 2) {
            if (plus(group[0]) == plus(group[1])) {
                if ((plus(group[0]) && vid(group[0]) < vid(group[1])) || (!plus(group[0]) && vid(group[0]) > vid(group[1]))) {
                    // deletion
                    auto pos1 = std::find(path.begin(), path.end(), group[0]);
                    auto pos2 = (pos1 == path.end()) ? path.end() : std::find(pos1 + 1, path.end(), group[1]);
                    if (pos1 == path.end() || pos2 == path.end()) {
                        complementAll(group);
                        pos1 = std::find(path.begin(), path.end(), group[0]);
                        pos2 = (pos1 == path.end()) ? path.end() : std::find(pos1 + 1, path.end(), group[1]);
                    }
                    if (pos1 == path.end() || pos2 == path.end() || pos2 - pos1 > 3) continue;
                    path.erase(pos1 + 1, pos2);
                } else {
                    // duplication
                    auto pos1 = std::find(path.begin(), path.end(), group[0]);
                    auto pos2 = std::find(path.begin(), pos1, group[1]);
                    if (pos1 == path.end() || pos2 == pos1) {
                        complementAll(group);
                        pos1 = std::find(path.begin(), path.end(), group[0]);
                        pos2 = std::find(path.begin(), pos1, group[1]);
                    }
                    if (pos1 == path.end() || pos2 == pos1) continue;
                    std::vector<int> copy(pos2, pos1 + 1);   // self-range insert: libstdc++ copies [pos2,pos1+1) as it stood
                    size_t at = (pos1 + 1) - path.begin();
                    path.insert(path.begin() + at, copy.begin(), copy.end());
                }
            } else {
                // inversion
                auto pos1 = std::find(path.begin(), path.end(), group[0]);
                auto pos2 = (pos1 == path.end()) ? path.end() : std::find(pos1 + 1, path.end(), group[1]);
                if (pos1 == path.end() || pos2 == path.end()) {
                    complementAll(group);
                    pos1 = std::find(path.begin(), path.end(), group[0]);
                    pos2 = (pos1 == path.end()) ? path.end() : std::find(pos1 + 1, path.end(), group[1]);
                }
                if (pos1 == path.end() || pos2 == path.end() || pos2 - pos1 > 5) continue;
                path.erase(pos1 + 1, pos2);
            }
        } else {
            // insertion
            auto pos1 = std::find(path.begin(), path.end(), group.front());
            auto pos2 = (pos1 == path.end()) ? path.end() : std::find(pos1 + 1, path.end(), group.back());
            if (pos1 == path.end() || pos2 == path.end()) {
                complementAll(group);
                pos1 = std::find(path.begin(), path.end(), group.front());
                pos2 = (pos1 == path.end()) ? path.end() : std::find(pos1 + 1, path.end(), group.back());
            }
            if (pos1 == path.end() || pos2 == path.end()) continue;
            size_t at = (pos1 + 1) - path.begin();
            path.erase(pos1 + 1, pos2);
            path.insert(path.begin() + at, group.begin() + 1, group.end() - 1);
        }
    }
    log.push_back("BFB path with insertion, deletion, or duplication:");
    log.push_back(formatPath(g, path));
    return true;
}

// ------------------------------------------------------------------------------------------
// translocationBFB (BFB-TRX, I2/C2)   LGM.cpp:4052-4193
// Iterators become (container, index) pairs: container -1 = res, otherwise paths[chrId].
// ------------------------------------------------------------------------------------------
void translocationBFB(const Graph& g, std::vector<std::vector<int>>& paths, std::vector<int>& res,
                      const std::string& mainChr, std::vector<std::string>& log, std::vector<std::string>* trace) {
    log.push_back("BFB with translocation:");
    auto note = [&](const char* what) { if (trace) trace->push_back(what); };
    auto segOf = [&](int v) -> const Seg* { for (auto& s : g.segs) if (s.id == std::abs(v)) return &s; return nullptr; };
    auto chrIdOf = [&](int v) { return segOf(v)->chrId; };
    auto chromOf = [&](int v) { return segOf(v)->chrom; };
    std::vector<int> sv;
    for (size_t ji = 0; ji < g.juncs.size(); ji++)
        if (chrIdOf(g.juncs[ji].src) != chrIdOf(g.juncs[ji].tgt)) sv.push_back((int)ji);
    for (auto& p : paths)
        if (!p.empty() && chromOf(p[0]) == mainChr) res.insert(res.end(), p.begin(), p.end());
    auto complementAll = [](std::vector<int>& v) { std::reverse(v.begin(), v.end()); for (auto& x : v) x = -x; };
    auto findFrom = [](const std::vector<int>& v, long from, int val) -> long {
        // std::find(v.begin()+from, v.end(), val); from > size behaves like an empty range (returns end)
        for (long q = from; q < (long)v.size(); q++) if (v[q] == val) return q;
        return (long)v.size();
    };
    auto rfind = [](const std::vector<int>& v, int val) -> long {   // find(rbegin,rend): index of last match or -1
        for (long q = (long)v.size() - 1; q >= 0; q--) if (v[q] == val) return q;
        return -1;
    };
    long startPos = 0;
    while (!sv.empty()) {
        std::vector<int> group;
        for (int i = 0; i < (int)sv.size(); i++) {
            const Junc& J = g.juncs[sv[i]];
            if (chromOf(J.src) == mainChr) { group.push_back(J.a_src()); group.push_back(J.a_tgt()); sv.erase(sv.begin() + i); break; }
            else if (chromOf(J.tgt) == mainChr) { group.push_back(J.b_src()); group.push_back(J.b_tgt()); sv.erase(sv.begin() + i); break; }
        }
        if (group.empty()) break;
        for (int i = 0; i < (int)sv.size(); i++) {
            const Junc& J = g.juncs[sv[i]];
            if (chrIdOf(group.back()) == chrIdOf(J.a_src())) { group.push_back(J.a_src()); group.push_back(J.a_tgt()); }
            else if (chrIdOf(group.back()) == chrIdOf(J.b_src())) { group.push_back(J.b_src()); group.push_back(J.b_tgt()); }
            else continue;
            sv.erase(sv.begin() + i);
            i = -1;
            if (chromOf(group.back()) == mainChr) break;
        }
        if (group.size() == 2) {   // concatenation
            long pos1 = rfind(res, group[0]);
            if (pos1 < 0) { complementAll(group); pos1 = rfind(res, group[0]); }
            if (pos1 < 0) { note("concat-skip"); continue; }
            res.erase(res.begin() + pos1 + 1, res.end());
            int id = chrIdOf(group[1]);
            if (id < 0 || id >= (int)paths.size()) { note("concat-skip"); continue; }
            long pos2 = findFrom(paths[id], 0, group[1]);
            if (pos2 == (long)paths[id].size()) { complementAll(paths[id]); pos2 = findFrom(paths[id], 0, group[1]); }
            if (pos2 == (long)paths[id].size()) { note("concat-skip"); continue; }
            res.insert(res.end(), paths[id].begin() + pos2, paths[id].end());
            startPos = 0;
            note("concat");
        } else {   // insertion
            if (vid(group.front()) > vid(group.back())) complementAll(group);
            struct It { int c; long i; };
            std::vector<It> pos;
            auto attempt = [&](long& flagOut) {
                pos.clear();
                long flag = findFrom(res, startPos, group[0]);
                flagOut = flag;
                pos.push_back({-1, flag});
                if (flag != (long)res.size()) {
                    for (size_t i = 1; i + 1 < group.size(); i += 2) {
                        int id = chrIdOf(group[i]);
                        std::vector<int>& pp = paths[id];
                        long pos1 = findFrom(pp, 0, group[i]);
                        if (pos1 == (long)pp.size()) { complementAll(pp); pos1 = findFrom(pp, 0, group[i]); }
                        if (pos1 == (long)pp.size()) break;
                        pos.push_back({id, pos1});
                        long r2 = rfind(pp, group[i + 1]);   // pos2.base() == r2+1
                        if (r2 < 0 || pos1 > r2 + 1) { complementAll(pp); r2 = rfind(pp, group[i + 1]); }
                        if (r2 < 0 || pos1 > r2 + 1) break;
                        pos.push_back({id, r2});
                    }
                }
                pos.push_back({-1, findFrom(res, flag + 1, group.back())});
            };
            long flag;
            attempt(flag);
            bool retried = false;
            if (pos.size() < group.size() || pos.back().i == (long)res.size()) { complementAll(group); attempt(flag); retried = true; }
            if (pos.size() < group.size() || pos.back().i == (long)res.size()) { note("insert-skip"); continue; }
            std::vector<int> temp;
            for (size_t i = 1; i + 1 < pos.size(); i += 2) {
                const std::vector<int>& pp = paths[pos[i].c];
                if (pos[i].i <= pos[i + 1].i) temp.insert(temp.end(), pp.begin() + pos[i].i, pp.begin() + pos[i + 1].i + 1);
            }
            if (temp.empty()) { note("insert-skip"); continue; }
            note(retried ? "insert-retry" : "insert");
            long a = pos.front().i + 1, b = pos.back().i;
            if (a <= b) res.erase(res.begin() + a, res.begin() + b);
            res.insert(res.begin() + a, temp.begin(), temp.end());
            startPos = findFrom(res, 0, temp.back());
        }
    }
    log.push_back(formatPath(g, res));
}


// ------------------------------------------------------------------------------------------
// TRX-BFB: insertBeforeBFB / concatBeforeBFB / virusBFB   (LGM.cpp:4195-4395, 3839-3939)
// Segment* becomes a segment id of the respective graph, Vertex* a signed id; `new Graph(vectors)` copies the four vectors
// (see the header).  std::unordered_map<int,int> is the reference's own container: the order of the "Seg conversion" lines is
// whatever libstdc++ makes of the same sequence of inserts.
// ------------------------------------------------------------------------------------------
static Seg copySeg(int newId, int chrId, const Seg& from) {   // Segment::Segment(int, int, Segment*)  Segment.cpp:27-45
    Seg s;
    s.id = newId; s.chrId = chrId; s.chrom = from.chrom; s.start = from.start; s.end = from.end;
    s.cov = from.cov; s.cn = from.cn; s.partition = 0;
    return s;
}
static void rebuiltSourcesSinks(const std::vector<Seg>& mSegs, std::vector<int>& mSources, std::vector<int>& mSinks) {   // LGM.cpp:4254-4262 / :4356-4364
    mSources.push_back(mSegs[0].id);
    for (size_t i = 1; i < mSegs.size(); i++)
        if (mSegs[i].chrId != mSegs[i - 1].chrId) { mSinks.push_back(mSegs[i - 1].id); mSources.push_back(mSegs[i].id); }
    mSinks.push_back(mSegs.back().id);
}
static void finishRebuild(Graph& g, std::vector<Seg>& mSegs, std::vector<Junc>& mJuncs, std::unordered_map<int, int>& segConversion,
                          TrxMap& map, std::vector<std::string>& log) {
    std::vector<int> mSources, mSinks;
    rebuiltSourcesSinks(mSegs, mSources, mSinks);
    log.push_back("Seg conversion:");                             // LGM.cpp:4285-4291 / :4385-4391
    map.originalOf.assign(mSegs.size() + 1, 0);
    for (auto iter = segConversion.begin(); iter != segConversion.end(); iter++) {
        log.push_back(std::to_string(iter->first) + "-" + std::to_string(iter->second));
        if (iter->second > 0) map.originalOf[iter->second] = iter->first;
    }
    map.original = g;
    Graph ng;                                                     // Graph(vector...) Graph.cpp:25-34: purity / ploidies -1, the vectors copied
    ng.purity = -1; ng.avgPloidy = -1; ng.avgTumorPloidy = -1;
    ng.segs = mSegs; ng.juncs = mJuncs; ng.sourceIds = mSources; ng.sinkIds = mSinks;
    g = ng;
    log.push_back("write seg");                                   // g->writeGraph("./new.lh"), Graph.cpp:249
}

bool insertBeforeBFB(Graph& g, const std::vector<std::string>& insChr, TrxMap& map, std::vector<std::string>& log, std::string& err) {
    std::unordered_map<int, int> segConversion;
    const std::vector<Seg>& segs = g.segs;
    const std::vector<Junc>& juncs = g.juncs;
    std::vector<Seg> mSegs;
    std::vector<Junc> mJuncs;
    auto chromOf = [&](int id) { return segs[id - 1].chrom; };   // junc->getSource()->getChrom()
    // search for segments and junctions involved in insertion  (:4204-4227)
    std::vector<int> insertionIDs, deletedChrIDs;
    std::vector<size_t> visited;
    for (size_t i = 1; i < insChr.size(); i++) {
        for (size_t j = 0; j < juncs.size(); j++) {
            if (std::find(visited.begin(), visited.end(), j) != visited.end()) continue;
            std::string chr1 = chromOf(juncs[j].src), chr2 = chromOf(juncs[j].tgt);
            if ((insChr[i - 1] == chr1 && insChr[i] == chr2) || (insChr[i - 1] == chr2 && insChr[i] == chr1)) {
                int id1 = juncs[j].src, id2 = juncs[j].tgt;
                if (insChr[i - 1] == chr2 && insChr[i] == chr1) std::swap(id1, id2);
                if (!insertionIDs.empty() && insertionIDs.back() != id1) {
                    if (insertionIDs.back() < id1) { for (int k = insertionIDs.back(); k < id1; k++) insertionIDs.push_back(k); }
                    else { for (int k = insertionIDs.back(); k > id1; k--) insertionIDs.push_back(k); }
                }
                insertionIDs.push_back(id1); insertionIDs.push_back(id2);
                visited.push_back(j);
                break;
            }
        }
    }
    insertionIDs.erase(std::unique(insertionIDs.begin(), insertionIDs.end()), insertionIDs.end());
    if (insertionIDs.size() < 2) { err = "insertBeforeBFB: no junction between the chromosomes of the I1 list (the reference reads an empty vector)"; return false; }
    if (insertionIDs.front() > insertionIDs.back()) std::reverse(insertionIDs.begin(), insertionIDs.end());
    const int sID = insertionIDs.front(), eID = insertionIDs.back();
    insertionIDs.erase(insertionIDs.begin());
    insertionIDs.pop_back();
    for (int id : insertionIDs) deletedChrIDs.push_back(segs[id - 1].chrId);
    // set mSegs  (:4234-4253)
    for (int i = 1; i <= (int)segs.size(); i++) {
        if (i < sID || i > eID) {
            if (std::find(deletedChrIDs.begin(), deletedChrIDs.end(), segs[i - 1].chrId) != deletedChrIDs.end()) continue;
            segConversion.insert(std::pair<int, int>(i, (int)mSegs.size() + 1));
            mSegs.push_back(copySeg((int)mSegs.size() + 1, segs[i - 1].chrId, segs[i - 1]));
        } else {
            segConversion.insert(std::pair<int, int>(sID, (int)mSegs.size() + 1));
            mSegs.push_back(copySeg((int)mSegs.size() + 1, segs[sID - 1].chrId, segs[sID - 1]));
            for (int j = sID + 1; j < eID; j++) segConversion.insert(std::pair<int, int>(j, 0));
            for (int id : insertionIDs) {
                segConversion.insert(std::pair<int, int>(id, (int)mSegs.size() + 1));
                mSegs.push_back(copySeg((int)mSegs.size() + 1, segs[sID - 1].chrId, segs[id - 1]));
            }
            segConversion.insert(std::pair<int, int>(eID, (int)mSegs.size() + 1));
            mSegs.push_back(copySeg((int)mSegs.size() + 1, segs[eID - 1].chrId, segs[eID - 1]));
            i = eID;
        }
    }
    // set mJuncs  (:4263-4283)
    for (const Junc& junc : juncs) {
        if (junc.a_src() == junc.a_tgt()) continue;
        const int startSegID = junc.src, targetSegID = junc.tgt;
        int id1 = segConversion[startSegID] - 1, id2 = segConversion[targetSegID] - 1;
        if (id1 == -1 || id2 == -1) { map.unusedSV.push_back(junc); continue; }
        char dir1 = junc.sdir, dir2 = junc.tdir;
        if (std::find(insertionIDs.begin(), insertionIDs.end(), startSegID) != insertionIDs.end() ||
            std::find(insertionIDs.begin(), insertionIDs.end(), targetSegID) != insertionIDs.end()) {
            if (id1 > id2) std::swap(id1, id2);
            dir1 = '+'; dir2 = '+';
        }
        log.push_back(std::to_string(startSegID) + "-" + std::to_string(targetSegID) + " " + std::to_string(id1 + 1) + "-" + std::to_string(id2 + 1));
        Junc nj = junc;
        nj.src = mSegs[id1].id; nj.tgt = mSegs[id2].id; nj.sdir = dir1; nj.tdir = dir2;
        mJuncs.push_back(nj);
    }
    finishRebuild(g, mSegs, mJuncs, segConversion, map, log);
    return true;
}

bool concatBeforeBFB(Graph& g, const std::vector<std::string>& conChr, TrxMap& map, std::vector<std::string>& log, std::string& err) {
    std::unordered_map<int, int> segConversion;
    const std::vector<Seg>& segs = g.segs;
    const std::vector<Junc>& juncs = g.juncs;
    const std::vector<int>& sources = g.sourceIds; const std::vector<int>& sinks = g.sinkIds;
    std::vector<Seg> mSegs;
    std::vector<Junc> mJuncs;
    if (conChr.size() < 2) { err = "concatBeforeBFB: C1 needs two chromosomes"; return false; }
    // the junction of the concatenation  (:4305-4323)
    int sID = 0, eID = 0; char sDir = 0, eDir = 0; bool found = false;
    for (size_t i = 0; i < juncs.size(); i++) {
        const Junc& junc = juncs[i];
        const std::string& c1 = segs[junc.src - 1].chrom; const std::string& c2 = segs[junc.tgt - 1].chrom;
        if ((c1 == conChr[0] && c2 == conChr[1]) || (c2 == conChr[0] && c1 == conChr[1])) {
            sID = junc.src; eID = junc.tgt; sDir = junc.sdir; eDir = junc.tdir; found = true;
            break;
        }
    }
    if (!found) { err = "concatBeforeBFB: no junction between the two chromosomes of the C1 list (the reference reads unset variables)"; return false; }
    log.push_back("Concat segs: " + std::to_string(sID) + sDir + " " + std::to_string(eID) + eDir);
    // set mSegs  (:4325-4355)
    const int chrID1 = segs[sID - 1].chrId;
    if (sDir == '+') {
        for (int i = sources[chrID1]; i <= sID; i++) {
            segConversion.insert(std::pair<int, int>(i, (int)mSegs.size() + 1));
            mSegs.push_back(copySeg((int)mSegs.size() + 1, segs[sID - 1].chrId, segs[i - 1]));
        }
        for (int i = sID + 1; i <= sinks[chrID1]; i++) segConversion.insert(std::pair<int, int>(i, 0));
    } else {
        for (int i = sinks[chrID1]; i >= sID; i--) {
            segConversion.insert(std::pair<int, int>(i, (int)mSegs.size() + 1));
            mSegs.push_back(copySeg((int)mSegs.size() + 1, segs[sID - 1].chrId, segs[i - 1]));
        }
        for (int i = sID - 1; i >= sources[chrID1]; i--) segConversion.insert(std::pair<int, int>(i, 0));
    }
    const int chrID2 = segs[eID - 1].chrId;
    if (eDir == '+') {
        for (int i = eID; i <= sinks[chrID2]; i++) {
            segConversion.insert(std::pair<int, int>(i, (int)mSegs.size() + 1));
            mSegs.push_back(copySeg((int)mSegs.size() + 1, segs[sID - 1].chrId, segs[i - 1]));
        }
        for (int i = sources[chrID2]; i < eID; i++) segConversion.insert(std::pair<int, int>(i, 0));
    } else {
        for (int i = eID; i >= sources[chrID2]; i--) {
            segConversion.insert(std::pair<int, int>(i, (int)mSegs.size() + 1));
            mSegs.push_back(copySeg((int)mSegs.size() + 1, segs[sID - 1].chrId, segs[i - 1]));
        }
        for (int i = sinks[chrID2]; i > eID; i--) segConversion.insert(std::pair<int, int>(i, 0));
    }
    for (int i = 1; i <= (int)segs.size(); i++) {
        if (segs[i - 1].chrId != chrID1 && segs[i - 1].chrId != chrID2) {
            segConversion.insert(std::pair<int, int>(i, (int)mSegs.size() + 1));
            mSegs.push_back(copySeg((int)mSegs.size() + 1, segs[i - 1].chrId, segs[i - 1]));
        }
    }
    // set mJuncs  (:4365-4383)
    for (const Junc& junc : juncs) {
        const int startSegID = junc.src, targetSegID = junc.tgt;
        int id1 = segConversion[startSegID] - 1, id2 = segConversion[targetSegID] - 1;
        char dir1 = junc.sdir, dir2 = junc.tdir;
        log.push_back(std::to_string(startSegID) + dir1 + " - " + std::to_string(targetSegID) + dir2 + " " + std::to_string(id1 + 1) + "-" + std::to_string(id2 + 1));
        if (id1 == -1 || id2 == -1) { map.unusedSV.push_back(junc); continue; }
        if ((startSegID == sID && targetSegID == eID) || (startSegID == eID && targetSegID == sID)) {
            if (id1 > id2) std::swap(id1, id2);
            dir1 = '+'; dir2 = '+';
        }
        Junc nj = junc;
        nj.src = mSegs[id1].id; nj.tgt = mSegs[id2].id; nj.sdir = dir1; nj.tdir = dir2;
        mJuncs.push_back(nj);
    }
    finishRebuild(g, mSegs, mJuncs, segConversion, map, log);
    return true;
}

// edges that leave vertex v of graph g, in the order Junction::insertEdgesToVertices registered them (Junction.cpp:95-121: edge A at
// its source, edge B at its source -- except for a fold-back of one segment onto itself, whose edge B is never registered): targets
static std::vector<int> edgeTargetsAsSource(const Graph& g, int v) {
    std::vector<int> t;
    for (const Junc& j : g.juncs) {
        if (j.a_src() == v) t.push_back(j.a_tgt());
        const bool selfFold = j.sdir != j.tdir && j.src == j.tgt;
        if (!selfFold && j.b_src() == v) t.push_back(j.b_tgt());
    }
    return t;
}

std::string formatPathMixed(const Graph& original, const Graph& rebuilt, const std::vector<int>& path, const std::vector<char>& rebuiltVertex) {
    std::string s;
    auto info = [](int v) { return std::to_string(std::abs(v)) + (v > 0 ? "+" : "-"); };
    auto chr = [&](size_t i) { const Graph& gg = rebuiltVertex[i] ? rebuilt : original; return gg.segs[std::abs(path[i]) - 1].chrId; };
    for (size_t i = 1; i < path.size(); i++) {
        s += info(path[i - 1]);
        if (chr(i - 1) != chr(i)) s += "||";
        else if (plus(path[i - 1]) != plus(path[i])) s += "|";
    }
    if (!path.empty()) s += info(path.back());
    return s;
}

bool virusBFB(const TrxMap& map, const Graph& rebuilt, std::vector<int>& path, std::vector<char>& rebuiltVertex,
              std::vector<std::string>& log, std::string& err) {
    const Graph& og = map.original;
    rebuiltVertex.assign(path.size(), 1);
    if (path.size() < 2) { err = "virusBFB: a path of one vertex (path->at(1) throws in the reference)"; return false; }
    auto origSeg = [&](int v) -> const Seg& { return og.segs[map.originalOf[std::abs(v)] - 1]; };   // originalSegs[v->getSegment()]
    // restore path segment into original segments  (:3841-3845)
    std::vector<bool> isFBI{false};
    for (size_t k = 1; k < path.size(); k++) isFBI.push_back(plus(path[k - 1]) != plus(path[k]));
    // first vertex  (:3847-3875)
    const Seg& seg1 = origSeg(path[0]); const Seg& seg2 = origSeg(path[1]);
    if (seg1.chrId != seg2.chrId) {
        bool found = false;
        for (int t : edgeTargetsAsSource(og, seg1.id)) if (std::abs(t) == seg2.id) { path[0] = seg1.id; rebuiltVertex[0] = 0; found = true; break; }
        if (!found) for (int t : edgeTargetsAsSource(og, -seg1.id)) if (std::abs(t) == seg2.id) { path[0] = -seg1.id; rebuiltVertex[0] = 0; break; }
    } else {
        path[0] = plus(path[0]) ? seg1.id : -seg1.id; rebuiltVertex[0] = 0;
    }
    // remaining vertices  (:3877-3900)
    for (size_t k = 1; k < path.size(); k++) {
        const Seg& seg = origSeg(path[k]);
        const Graph& pg = rebuiltVertex[k - 1] ? rebuilt : og;          // (*(iter-1))->getSegment(): whatever graph that vertex belongs to
        const int prevChr = pg.segs[std::abs(path[k - 1]) - 1].chrId;
        if (prevChr != seg.chrId) {
            // edges of the previous vertex: a vertex of the rebuilt graph has none (its junctions were never registered, LGM.cpp:4281 / :4382)
            if (!rebuiltVertex[k - 1])
                for (int t : edgeTargetsAsSource(og, path[k - 1])) if (std::abs(t) == seg.id) { path[k] = t; rebuiltVertex[k] = 0; break; }
        } else if (isFBI[k]) {
            path[k] = plus(path[k - 1]) ? -seg.id : seg.id; rebuiltVertex[k] = 0;
        } else {
            path[k] = plus(path[k - 1]) ? seg.id : -seg.id; rebuiltVertex[k] = 0;
        }
    }
    log.push_back("TRX-BFB mode: BFB path in the first stage:");
    log.push_back(formatPathMixed(og, rebuilt, path, rebuiltVertex));
    // deal with extra SV in the second stage  (:3904-3938); vertex identity = (graph, signed id): the SVs are junctions of the original graph
    auto findRev = [&](int v) { for (size_t r = 0; r < path.size(); r++) { size_t i = path.size() - 1 - r; if (!rebuiltVertex[i] && path[i] == v) return r; } return path.size(); };   // distance from rbegin()
    auto findFwd = [&](int v) { for (size_t i = 0; i < path.size(); i++) if (!rebuiltVertex[i] && path[i] == v) return i; return path.size(); };
    for (size_t i = 0; i < map.unusedSV.size(); i++) {
        const Junc& sv = map.unusedSV[i];
        bool isEdgeA = true;
        size_t pos1 = findRev(sv.a_src());
        if (pos1 == path.size()) { pos1 = findRev(sv.b_src()); isEdgeA = false; }
        if (pos1 == path.size()) continue;
        const int headVertex = isEdgeA ? sv.b_tgt() : sv.a_tgt();         // looked for from the front
        const int newHead = isEdgeA ? sv.b_src() : sv.a_src();
        const int newTail = isEdgeA ? sv.a_tgt() : sv.b_tgt();
        const size_t pos2 = findFwd(headVertex);
        if (pos2 != path.size() && pos2 < pos1) {
            path.erase(path.begin(), path.begin() + pos2); rebuiltVertex.erase(rebuiltVertex.begin(), rebuiltVertex.begin() + pos2);
            path.insert(path.begin(), newHead); rebuiltVertex.insert(rebuiltVertex.begin(), 0);
        } else {
            const size_t base = path.size() - pos1;                      // pos1.base(): the element behind the one found
            path.erase(path.begin() + base, path.end()); rebuiltVertex.erase(rebuiltVertex.begin() + base, rebuiltVertex.end());
            path.push_back(newTail); rebuiltVertex.push_back(0);
        }
        log.push_back("TRX-BFB mode: BFB path in the second stage:");
        log.push_back(formatPathMixed(og, rebuilt, path, rebuiltVertex));
        break;
    }
    return true;
}

void synthesizeOutputJuncs(const std::vector<int>& p, std::vector<OutJunc>& out, bool increase) {
    // localhap.cpp:269-289 (increase=true) and :298-315 (BFB-TRX result, increase=false)
    if (p.empty()) return;   // reference: size()-1 underflows on an empty path
    for (size_t i = 0; i + 1 < p.size(); i++) {
        int u = p[i], v = p[i + 1];
        if (!(std::abs(vid(u) - vid(v)) == 1 && plus(u) == plus(v))) {
            bool hasJunc = false;
            for (auto& j : out) {
                if ((j.u == u && j.v == v) || (-j.v == u && -j.u == v)) { hasJunc = true; if (increase) j.count += 1; }
            }
            if (!hasJunc) out.push_back({u, v, 1});
        }
    }
}

bool readSol(const std::string& path, Sol& sol) {   // localhap.cpp:184-212 (token scan)
    std::ifstream f(path);
    if (!f) return false;
    std::string element, cn;
    while (f >> element) {
        if (element == "Infeasible") { sol.infeasible = true; break; }
        if (element == "value") { double t = 0; f >> t; sol.objective += t; }
        if (element[0] == 'x') {
            int x = stoi(element.substr(1));
            f >> cn;
            sol.cols.push_back({x, stoi(cn)});
        }
    }
    return true;
}

// ------------------------------------------------------------------------------------------
// whole run  (localhap.cpp:49-388)
// ------------------------------------------------------------------------------------------
RunResult runBfb(const RunOptions& opt) {
    RunResult R;
    R.log.push_back("bfb");
    Graph g;
    if (!readGraph(opt.lh, g, R.err)) return R;
    if (!calculateHapDepth(g, R.err)) return R;
    calculateCopyNum(g);
    for (auto& l : g.log) R.log.push_back(l);
    g.log.clear();
    Props props;
    readBFBProps(opt.lh, props);
    TrxMap trx;
    Graph rebuilt;
    const bool trxBefore = props.insMode == 1 || props.conMode == 1;     // localhap.cpp:79-88
    if (trxBefore) {
        if (!opt.juncs.empty()) { R.err = "TRX-BFB (I1/C1) with a .juncs file: readComponents reads a member of the rebuilt graph that nothing sets"; return R; }
        if (props.insMode == 1) { if (!insertBeforeBFB(g, props.insChr, trx, R.log, R.err)) return R; }
        else if (!concatBeforeBFB(g, props.conChr, trx, R.log, R.err)) return R;
        rebuilt = g;
        R.trxBefore = true; R.originalOf = trx.originalOf;
    }
    for (size_t i = 0; i < g.sourceIds.size(); i++)   // localhap.cpp:94-98
        for (int j = g.sourceIds[i]; j <= g.sinkIds[i]; j++) {
            if (j - 1 < 0 || j - 1 >= (int)g.segs.size()) { R.err = "segment ids must be 1..N"; return R; }
            g.segs[j - 1].partition = (int)i;
        }
    std::vector<std::vector<int>> components;
    readComponents(g, opt.juncs, components, R.log);
    R.targetCN.assign(g.segs.size(), 0);
    size_t solCursor = 0;
    for (size_t n = 0; n < g.sinkIds.size(); n++) {
        ChrStage st;
        int startID = g.sourceIds[n], endID = g.sinkIds[n];
        st.startID = startID; st.endID = endID;
        int numPat = 0;
        std::map<std::string, int> variableIdx = makeVariableIdx(startID, endID, &numPat);
        int numComp = (int)variableIdx.size();
        auto tA = std::chrono::steady_clock::now();
        Inversions inversions;
        getJuncCN(g, startID, endID, inversions, st.juncCN);
        R.numInv += (int)inversions.size();
        st.bias = computeBias(g, startID, endID, inversions, st.juncCN);
        getIndelBias(g, startID, endID);
        R.reconSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - tA).count();
        for (auto& s : g.segs) st.segCNAfterIndelBias.push_back(s.cn);
        {
            std::vector<std::pair<int, int>> iv(inversions.begin(), inversions.end());
            std::sort(iv.begin(), iv.end());
            for (auto& kv : iv) { st.invSeg.push_back(kv.first); st.invJunc.push_back(kv.second); }
        }
        double inversionCNSum = 0;
        for (int i = 0; i <= endID; i++) inversionCNSum += st.juncCN[i * 2 + 1];
        std::vector<std::vector<int>> validComponents;
        for (auto& comp : components)
            if (g.segById(comp[0]) && g.segById(comp[0])->partition == (int)n) validComponents.push_back(comp);
        if (std::abs(inversionCNSum) < 0.000001 && validComponents.size() == 0) {   // localhap.cpp:164-170
            st.shortcut = true;
            std::vector<int> temp;
            for (int i = startID; i <= endID; i++) temp.push_back(i);
            R.log.push_back(formatPath(g, temp));
            R.paths.push_back(temp);
            R.chr.push_back(st);
            continue;
        }
        R.log.push_back("Declare done");            // BFB_ILP progress lines (LGM.cpp:4418,4705,4731)
        R.log.push_back("ILP formula done");
        R.log.push_back("Variable constrains done");
        if (solCursor >= opt.solPerChr.size()) { R.err = "missing .sol for chromosome " + std::to_string(n); return R; }
        Sol sol;
        if (!readSol(opt.solPerChr[solCursor++], sol)) { R.err = "ILP error: cannot open file"; return R; }
        std::vector<int> elementCN(numComp, 0);
        R.ilpError += sol.objective;
        for (auto& c : sol.cols) if (c.first >= 0 && c.first < numComp) elementCN[c.first] = c.second;
        if (sol.infeasible) {
            st.infeasible = true;
            std::vector<int> temp;
            for (int i = startID; i <= endID; i++) temp.push_back(i);
            R.log.push_back(formatPath(g, temp));
            R.log.push_back("ILP is unsolvable.");
            R.paths.push_back(temp);
            R.chr.push_back(st);
            continue;
        }
        auto tB = std::chrono::steady_clock::now();
        for (auto iter = variableIdx.begin(); iter != variableIdx.end(); iter++) {   // localhap.cpp:222-232
            if (elementCN[iter->second] > 0) {
                const std::string& key = iter->first;
                int idx1 = stoi(key.substr(2, key.find(",") - 2)), idx2 = stoi(key.substr(key.find(",") + 1));
                for (int i = idx1 - 1; i < idx2; i++) {
                    if (key[0] == 'p') R.targetCN[i] += elementCN[iter->second];
                    else R.targetCN[i] += elementCN[iter->second] * 2;
                }
            }
        }
        constructDAG(variableIdx, elementCN, st.dag);
        std::vector<std::vector<int>> orders;
        allTopologicalOrders(st.dag, orders, opt.maxOrders);
        st.numOrders = (long)orders.size();
        getBFB(g, orders, st.dag, inversions, opt.reversed, opt.all, st.bfb, R.log);
        if (opt.keepOrders) st.orders = orders;
        std::vector<int> path = st.bfb.path;
        st.indelPrinted = indelBFB(g, path, startID, endID, R.log);
        R.reconSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - tB).count();
        st.pathAfterIndel = path;
        if (trxBefore) {                                                 // localhap.cpp:263
            std::vector<char> rv;
            if (!virusBFB(trx, rebuilt, path, rv, R.log, R.err)) return R;
            for (char c : rv) if (c) { R.err = "virusBFB left a vertex of the rebuilt graph in the path (no junction of the original graph leads to its segment)"; return R; }
        }
        R.paths.push_back(path);
        R.chr.push_back(st);
    }
    {
        auto tC = std::chrono::steady_clock::now();
        for (auto& p : R.paths) { R.pathLen += (int)p.size(); synthesizeOutputJuncs(p, R.outJuncs, true); }
        R.reconSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - tC).count();
    }
    for (auto& s : g.segs) {   // localhap.cpp:290-293 (int accumulators fed with doubles)
        R.cnSum += s.cn;
        R.maxCN = (R.maxCN > s.cn) ? R.maxCN : s.cn;
    }
    if (props.insMode == 2 || props.conMode == 2) {
        if (props.mainChr.empty()) { R.err = "BFB-TRX without M:<chr> (reference segfaults)"; return R; }
        R.trxRun = true;
        translocationBFB(g, R.paths, R.trxPath, props.mainChr, R.log, &R.trxTrace);
        synthesizeOutputJuncs(R.trxPath, R.outJuncs, false);
    }
    R.ok = true;
    return R;
}

// ------------------------------------------------------------------------------------------
// whole `--op sc_bfb` run  (localhap.cpp:390-679)
// ------------------------------------------------------------------------------------------
ScResult runScBfb(const std::vector<std::string>& lhs, const RunOptions& opt) {
    ScResult R;
    R.log.push_back("sc_bfb");
    std::vector<Graph> graphs(lhs.size());
    for (size_t k = 0; k < lhs.size(); k++) {   // :402-414
        if (!readGraph(lhs[k], graphs[k], R.err)) return R;
        if (!calculateHapDepth(graphs[k], R.err)) return R;
        calculateCopyNum(graphs[k]);
        for (auto& l : graphs[k].log) R.log.push_back(l);
        graphs[k].log.clear();
    }
    const int numGraphs = (int)graphs.size();
    if (numGraphs == 0) { R.err = "no .lh"; return R; }
    std::vector<std::vector<int>> evolution(numGraphs);   // :430-434 (the `edges` option is hard-wired to "")
    for (int i = 0; i < numGraphs; i++) for (int j = i + 1; j < numGraphs; j++) evolution[i].push_back(j);
    Graph& g = graphs[0];
    if (!calculateHapDepth(g, R.err)) return R;   // :438-439: a second time on the first graph
    calculateCopyNum(g);
    for (auto& l : g.log) R.log.push_back(l);
    g.log.clear();
    Props props;
    readBFBProps(lhs[0], props);                  // lhRawFn was cut at the first comma by strtok_r (:405-407)
    if (props.insMode == 1 || props.conMode == 1) { R.err = "TRX-BFB (I1/C1) not supported"; return R; }
    for (size_t i = 0; i < g.sourceIds.size(); i++)
        for (int j = g.sourceIds[i]; j <= g.sinkIds[i]; j++) {
            if (j - 1 < 0 || j - 1 >= (int)g.segs.size()) { R.err = "segment ids must be 1..N"; return R; }
            g.segs[j - 1].partition = (int)i;
        }
    R.paths.assign(numGraphs, {});
    std::vector<int> targetCN(g.segs.size(), 0);
    size_t solCursor = 0;
    for (size_t n = 0; n < g.sourceIds.size(); n++) {
        int startID = g.sourceIds[n], endID = g.sinkIds[n];
        int numPat = 0;
        std::map<std::string, int> variableIdx = makeVariableIdx(startID, endID, &numPat);
        int numComp = (int)variableIdx.size();
        Inversions inversions; std::vector<double> juncCN;
        getJuncCN(g, startID, endID, inversions, juncCN);
        (void)computeBias(g, startID, endID, inversions, juncCN);
        getIndelBias(g, startID, endID);          // :497, on the FIRST graph only
        double inversionCNSum = 0;
        for (int i = 0; i <= endID; i++) inversionCNSum += juncCN[i * 2 + 1];
        std::vector<ChrStage> stages(numGraphs);
        if (std::abs(inversionCNSum) < 0.000001) {   // :505-512: decided by the first graph; nothing is printed
            for (int k = 0; k < numGraphs; k++) {
                std::vector<int> temp;
                for (int i = startID; i <= endID; i++) temp.push_back(i);
                R.paths[k].push_back(temp);
                stages[k].shortcut = true;
            }
            R.chr.push_back(stages);
            continue;
        }
        R.log.push_back("Declare done");             // BFB_ILP_SC: LGM.cpp:4773, :5010 (once per graph), :5084
        for (int k = 0; k < numGraphs; k++) R.log.push_back("ILP formula done");
        R.log.push_back("Variable constrains done");
        if (solCursor >= opt.solPerChr.size()) { R.err = "missing .sol for chromosome " + std::to_string(n); return R; }
        Sol sol;
        if (!readSol(opt.solPerChr[solCursor++], sol)) { R.err = "ILP error: cannot open file"; return R; }
        std::vector<int> elementCN((size_t)numComp * numGraphs, 0);
        for (auto& c : sol.cols) if (c.first >= 0 && c.first < numComp * numGraphs) elementCN[c.first] = c.second;   // :536
        if (sol.infeasible) {                         // :543-551
            R.log.push_back("ILP is unsolvable.");
            for (int k = 0; k < numGraphs; k++) {
                std::vector<int> temp;
                for (int i = startID; i <= endID; i++) temp.push_back(i);
                R.paths[k].push_back(temp);
                stages[k].infeasible = true;
            }
            R.chr.push_back(stages);
            continue;
        }
        for (int k = 0; k < numGraphs; k++) {         // :554-621
            ChrStage& st = stages[k];
            st.startID = startID; st.endID = endID;
            for (auto& kv : variableIdx) kv.second = kv.second % numComp + k * numComp;   // :557-562 (p: i + k*numComp, l: i + numComp/2 + k*numComp)
            for (auto iter = variableIdx.begin(); iter != variableIdx.end(); iter++)       // :565-576 (accumulated, never read)
                if (elementCN[iter->second] > 0) {
                    const std::string& key = iter->first;
                    int idx1 = stoi(key.substr(2, key.find(",") - 2)), idx2 = stoi(key.substr(key.find(",") + 1));
                    for (int i = idx1 - 1; i < idx2; i++) targetCN[i] += (key[0] == 'p') ? elementCN[iter->second] : elementCN[iter->second] * 2;
                }
            constructDAG(variableIdx, elementCN, st.dag);
            std::vector<std::vector<int>> orders;
            allTopologicalOrders(st.dag, orders, opt.maxOrders);
            st.numOrders = (long)orders.size();
            Inversions graphInversions;               // :603-604: of graph k
            getJuncCN(graphs[k], startID, endID, graphInversions, st.juncCN);
            getBFB(graphs[k], orders, st.dag, graphInversions, opt.reversed, opt.all, st.bfb, R.log);
            std::vector<int> path = st.bfb.path;
            st.indelPrinted = indelBFB(graphs[k], path, startID, endID, R.log);
            st.pathAfterIndel = path;
            R.paths[k].push_back(path);
        }
        R.chr.push_back(stages);
    }
    for (int k = 0; k < numGraphs; k++) {             // :654-660
        for (auto& p : R.paths[k]) R.pathLen += (int)p.size();
        for (auto& sg : graphs[k].segs) { R.cnSum += sg.cn; R.maxCN = (R.maxCN > sg.cn) ? R.maxCN : sg.cn; }
    }
    R.trxPaths.assign(numGraphs, {});
    if (props.insMode == 2 || props.conMode == 2) {   // :661-664, every graph
        if (props.mainChr.empty()) { R.err = "BFB-TRX without M:<chr> (reference segfaults)"; return R; }
        for (int k = 0; k < numGraphs; k++) translocationBFB(graphs[k], R.paths[k], R.trxPaths[k], props.mainChr, R.log);
    }
    R.nSeg = (int)g.segs.size(); R.nJunc = (int)g.juncs.size();
    R.ok = true;
    return R;
}

}  // namespace oracle
