// oracle/bfb_ilp_oracle.cpp -- TEST INFRASTRUCTURE ONLY.
// Semantic restatement of LocalGenomicMap::BFB_ILP (LGM.cpp:4397-4752): the sparse row-major ILP the
// reference hands to CoinUtils before shelling out to `cbc`.  Rows are emitted in the reference's order with the
// reference's in-row entry order; "infinity" is DBL_MAX (OsiClp's getInfinity()).
// The LP/MPS *text* that CoinLpIO/CoinMpsIO write is third-party (coin-or-utils 2.11.6, environment.yml:15-20) and
// unpinned by any reference test -- parity for this stage is on (rowPtr, colIdx, val, bounds, objective, integrality).
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <map>
#include <set>
#include <string>

#include "bfb_oracle.hpp"

namespace oracle {

void buildBfbIlp(const Graph& g, int startSegID, int endSegID, const std::vector<double>& juncCN,
                 const std::vector<std::vector<int>>& components, bool juncsInfo, int bias, IlpModel& m,
                 bool literalHotLoop) {
    const double INF = DBL_MAX;
    // patterns == loops == all (a,b), a<=b, lexicographic (LGM.cpp:3254-3264); index(p:a,b)=rank, index(l:a,b)=numPat+rank
    std::vector<std::pair<int, int>> pats;
    for (int a = startSegID; a <= endSegID; a++)
        for (int b = a; b <= endSegID; b++) pats.push_back({a, b});
    const int numPat = (int)pats.size(), numLoop = numPat;
    const int n = endSegID - startSegID + 1;
    auto rank = [&](int a, int b) { int da = a - startSegID; return da * n - da * (da - 1) / 2 + (b - a); };
    auto P = [&](int a, int b) { return rank(a, b); };
    auto Lp = [&](int a, int b) { return numPat + rank(a, b); };
    std::vector<const Seg*> segs;   // LGM.cpp:4402-4406
    for (auto& s : g.segs) if (startSegID <= s.id && s.id <= endSegID) segs.push_back(&s);
    const int numSegments = (int)segs.size();
    const int numElements = 2 * numPat, numEpsilons = numSegments * 2;
    const int numVariables = numElements + numEpsilons + 1;
    m = IlpModel();
    m.numCols = numVariables;
    m.rowPtr.push_back(0);
    auto endRow = [&](double lo, double up) { m.rowPtr.push_back((int64_t)m.colIdx.size()); m.rowLo.push_back(lo); m.rowUp.push_back(up); };
    auto put = [&](int c, double v) { m.colIdx.push_back(c); m.val.push_back(v); };
    int idx = 0;
    for (int i = startSegID; i <= endSegID; i++) {   // LGM.cpp:4423-4496
        size_t rowStart = m.colIdx.size();
        for (int j = 0; j < numPat; j++) if (pats[j].first <= i && i <= pats[j].second) put(P(pats[j].first, pats[j].second), 1);
        for (int j = 0; j < numLoop; j++) if (pats[j].first <= i && i <= pats[j].second) put(Lp(pats[j].first, pats[j].second), 2);
        size_t rowLen = m.colIdx.size() - rowStart;
        put(numElements + idx / 2, 1);
        endRow(segs[i - startSegID]->cn, INF); idx++;
        for (size_t k = 0; k < rowLen; k++) put(m.colIdx[rowStart + k], m.val[rowStart + k]);
        put(numElements + idx / 2, -1);
        endRow(-INF, segs[i - startSegID]->cn); idx++;

        std::vector<double> coef(numElements, 0.0);
        for (int j = 0; j < numLoop; j++)
            if (pats[j].first == i || pats[j].second == i) coef[Lp(pats[j].first, pats[j].second)] += 1;
        if (literalHotLoop) {   // LGM.cpp:4464-4477, the O(numPat^2) double loop as written
            for (int j = 0; j < numPat; j++)
                for (int k = 0; k < numPat; k++)
                    if ((pats[j].first == i && pats[k].first == i) || (pats[j].second == i && pats[k].second == i)) {
                        int diff1 = pats[j].first - pats[j].second, diff2 = pats[k].first - pats[k].second;
                        if (std::abs(diff1) > std::abs(diff2)) { coef[P(pats[j].first, pats[j].second)] = 0.5; coef[P(pats[k].first, pats[k].second)] = 0.5; }
                    }
        } else {                // closed form of the same loop: any two patterns sharing start i (or end i) differ in length
            if (i < endSegID) for (int b = i; b <= endSegID; b++) coef[P(i, b)] = 0.5;
            if (i > startSegID) for (int a = startSegID; a <= i; a++) coef[P(a, i)] = 0.5;
        }
        rowStart = m.colIdx.size();
        for (int c = 0; c < numElements; c++) if (coef[c] > 0.1) put(c, coef[c]);
        rowLen = m.colIdx.size() - rowStart;
        put(numElements + idx / 2, 1);
        endRow(juncCN[i * 2 + 1], INF); idx++;
        for (size_t k = 0; k < rowLen; k++) put(m.colIdx[rowStart + k], m.val[rowStart + k]);
        put(numElements + idx / 2, -1);
        endRow(-INF, juncCN[i * 2 + 1]); idx++;
    }
    put(numVariables - 1, 1); endRow(bias, bias); idx++;   // LGM.cpp:4498-4503

    for (int i = 0; i < numPat; i++) {   // LGM.cpp:4544-4583
        int a = pats[i].first, b = pats[i].second;
        bool flag1 = (a > startSegID) || (b < endSegID), flag2 = a < b;
        if (flag1) {
            for (int j = startSegID; j < a; j++) put(P(j, b), 1);
            for (int j = b + 1; j <= endSegID; j++) put(P(a, j), 1);
            put(P(a, b), -1); endRow(0, INF); idx++;
        }
        if (flag2) {
            for (int j = a; j < b; j++) put(P(a, j), 1);
            for (int j = a + 1; j <= b; j++) put(P(j, b), 1);
            put(P(a, b), 1); endRow(0, 2); idx++;
        }
    }
    for (int i = 0; i < numLoop; i++) {   // LGM.cpp:4587-4612
        int a = pats[i].first, b = pats[i].second;
        bool flag = (a > startSegID) || (b < endSegID);
        if (flag) {
            for (int j = startSegID; j < a; j++) { put(P(j, b), 1); put(Lp(j, b), 1); }
            for (int j = b + 1; j <= endSegID; j++) { put(P(a, j), 1); put(Lp(a, j), 1); }
            put(Lp(a, b), -1); endRow(0, INF); idx++;
        }
    }
    for (int i = 0; i < numLoop; i++) {   // LGM.cpp:4615-4646
        int a = pats[i].first, b = pats[i].second;
        if (a < b) {
            for (int rep = 0; rep < 2; rep++) {
                for (int j = a; j < b; j++) put(Lp(a, j), 1);
                for (int j = a + 1; j <= b; j++) put(Lp(j, b), 1);
                put(rep == 0 ? Lp(a, b) : P(a, b), 1); endRow(0, 2); idx++;
            }
        }
    }
    for (int i = 0; i < numPat; i++) {   // LGM.cpp:4649-4681
        int a = pats[i].first, b = pats[i].second;
        if (a < b) {
            for (int j = a; j < b; j++) put(Lp(a, j), 1);
            for (int j = a + 1; j <= b; j++) put(P(j, b), 1);
            put(P(a, b), 1); endRow(0, 2); idx++;
            for (int j = a; j < b; j++) put(P(a, j), 1);
            for (int j = a + 1; j <= b; j++) put(Lp(j, b), 1);
            put(P(a, b), 1); endRow(0, 2); idx++;
        }
    }
    if (components.size() > 0 && juncsInfo) {   // LGM.cpp:4684-4703
        std::set<std::pair<int, int>> seen;
        for (auto& comp : components) {
            int s = std::min(comp.front(), comp.back()), e = std::max(comp.front(), comp.back());
            if (s == startSegID && e == endSegID) continue;
            if (seen.count({s, e})) continue;
            seen.insert({s, e});
            bool inRange = (s >= startSegID && e <= endSegID && s <= e);
            put(inRange ? Lp(s, e) : 0, 1);   // variableIdx[] on a missing key default-inserts 0
            put(inRange ? P(s, e) : 0, 1);
        }
        endRow(0, 5); idx++;
    }
    double maxCN = 0;   // LGM.cpp:4708-4711: over ALL segments of the graph
    for (auto& s : g.segs) maxCN += s.cn;
    m.colLo.assign(numVariables, 0); m.colUp.assign(numVariables, 0); m.obj.assign(numVariables, 0);
    for (int c = 0; c < numPat; c++) { m.colLo[c] = 0; m.colUp[c] = 1; }
    for (int c = numPat; c < numElements; c++) { m.colLo[c] = 0; m.colUp[c] = maxCN; }
    for (int c = 0; c < numEpsilons; c++) { m.colLo[numElements + c] = 0; m.colUp[numElements + c] = INF; }
    m.colLo[numVariables - 1] = bias; m.colUp[numVariables - 1] = bias;
    for (int c = 0; c < numVariables; c++) m.obj[c] = (c < numElements) ? 0 : (c < numVariables - 1 ? 1 : -1);
    m.numInt = numElements;
    (void)idx;
}

// LocalGenomicMap::BFB_ILP_SC (LGM.cpp:4754-5093), restated literally: std::map<std::string,int> variableIdx with the
// per-graph "+= numComp" shift (:4784-4791), the running row counter `idx` in the epsilon column of the segment rows
// (:4815, :4821, :4858, :4864), `cnt` for the linking epsilons (:5031-5072).  juncCNs[g] = getJuncCN of graph g.
void buildBfbIlpSc(const std::vector<const Graph*>& graphs, int startSegID, int endSegID, const std::vector<std::vector<double>>& juncCNs,
                   const std::vector<std::vector<int>>& evolution, IlpModel& m) {
    const double INF = DBL_MAX;
    const int numGraphs = (int)graphs.size();
    std::vector<std::vector<int>> patterns;
    for (int a = startSegID; a <= endSegID; a++)
        for (int b = a; b <= endSegID; b++) patterns.push_back({a, b});
    const std::vector<std::vector<int>>& loops = patterns;
    int numPatTmp = 0;
    std::map<std::string, int> variableIdx = makeVariableIdx(startSegID, endSegID, &numPatTmp);
    auto ps = [](int a, int b) { return "p:" + std::to_string(a) + "," + std::to_string(b); };
    auto ls = [](int a, int b) { return "l:" + std::to_string(a) + "," + std::to_string(b); };
    const int numSegments = endSegID - startSegID + 1;
    const int numElements = (int)variableIdx.size() * numGraphs;
    const int numEpsilons = numSegments * 2 * numGraphs + (numGraphs * (numGraphs - 1)) * (int)variableIdx.size();
    const int numVariables = numElements + numEpsilons, numPat = (int)patterns.size(), numLoop = (int)loops.size();
    const int numComp = numPat + numLoop;
    m = IlpModel();
    m.numCols = numVariables;
    m.rowPtr.push_back(0);
    m.colLo.assign(numVariables, 0); m.colUp.assign(numVariables, 0); m.obj.assign(numVariables, 0);
    int idx = 0;
    auto put = [&](int c, double v) { m.colIdx.push_back(c); m.val.push_back(v); };
    auto endRow = [&](double lo, double up) { m.rowPtr.push_back((int64_t)m.colIdx.size()); m.rowLo.push_back(lo); m.rowUp.push_back(up); idx++; };
    for (int n = 0; n < numGraphs; n++) {
        std::vector<const Seg*> segs;
        for (auto& sg : graphs[n]->segs) if (startSegID <= sg.id && sg.id <= endSegID) segs.push_back(&sg);
        if (n > 0)
            for (int i = 0; i < numPat; i++) { variableIdx[ps(patterns[i][0], patterns[i][1])] += numComp; variableIdx[ls(patterns[i][0], patterns[i][1])] += numComp; }
        const std::vector<double>& juncCN = juncCNs[n];
        for (int i = startSegID; i <= endSegID; i++) {
            std::vector<std::pair<int, double>> c1;
            for (int j = 0; j < numPat; j++) if (patterns[j][0] <= i && i <= patterns[j][1]) c1.push_back({variableIdx[ps(patterns[j][0], patterns[j][1])], 1});
            for (int j = 0; j < numLoop; j++) if (loops[j][0] <= i && i <= loops[j][1]) c1.push_back({variableIdx[ls(loops[j][0], loops[j][1])], 2});
            for (auto& e : c1) put(e.first, e.second);
            put(numElements + idx / 2, 1); endRow(segs[i - startSegID]->cn, INF);
            for (auto& e : c1) put(e.first, e.second);
            put(numElements + idx / 2, -1); endRow(-INF, segs[i - startSegID]->cn);
            std::vector<double> coef(numElements, 0.0);
            for (int j = 0; j < numLoop; j++) if (loops[j][0] == i || loops[j][1] == i) coef[variableIdx[ls(loops[j][0], loops[j][1])]] += 1;
            for (int j = 0; j < numPat; j++)
                for (int k = 0; k < numPat; k++)
                    if ((patterns[j][0] == i && patterns[k][0] == i) || (patterns[j][1] == i && patterns[k][1] == i)) {
                        int diff1 = patterns[j][0] - patterns[j][1], diff2 = patterns[k][0] - patterns[k][1];
                        if (std::abs(diff1) > std::abs(diff2)) {
                            coef[variableIdx[ps(patterns[j][0], patterns[j][1])]] = 0.5;
                            coef[variableIdx[ps(patterns[k][0], patterns[k][1])]] = 0.5;
                        }
                    }
            std::vector<std::pair<int, double>> c5;
            for (int q = 0; q < numElements; q++) if (coef[q] > 0.1) c5.push_back({q, coef[q]});
            for (auto& e : c5) put(e.first, e.second);
            put(numElements + idx / 2, 1); endRow(juncCN[i * 2 + 1], INF);
            for (auto& e : c5) put(e.first, e.second);
            put(numElements + idx / 2, -1); endRow(-INF, juncCN[i * 2 + 1]);
        }
        for (int i = 0; i < numPat; i++) {   // LGM.cpp:4867-4911
            std::vector<std::pair<int, double>> c8, c9;
            bool flag1 = false, flag2 = false;
            for (int j = startSegID; j < patterns[i][0]; j++) { flag1 = true; c8.push_back({variableIdx[ps(j, patterns[i][1])], 1}); }
            for (int j = patterns[i][1] + 1; j <= endSegID; j++) { flag1 = true; c8.push_back({variableIdx[ps(patterns[i][0], j)], 1}); }
            for (int j = patterns[i][0]; j < patterns[i][1]; j++) { flag2 = true; c9.push_back({variableIdx[ps(patterns[i][0], j)], 1}); }
            for (int j = patterns[i][0] + 1; j <= patterns[i][1]; j++) { flag2 = true; c9.push_back({variableIdx[ps(j, patterns[i][1])], 1}); }
            if (flag1) { for (auto& e : c8) put(e.first, e.second); put(variableIdx[ps(patterns[i][0], patterns[i][1])], -1); endRow(0, INF); }
            if (flag2) { for (auto& e : c9) put(e.first, e.second); put(variableIdx[ps(patterns[i][0], patterns[i][1])], 1); endRow(0, 2); }
        }
        for (int i = 0; i < numLoop; i++) {   // :4914-4940
            std::vector<std::pair<int, double>> c9;
            bool flag = false;
            for (int j = startSegID; j < loops[i][0]; j++) { flag = true; c9.push_back({variableIdx[ps(j, loops[i][1])], 1}); c9.push_back({variableIdx[ls(j, loops[i][1])], 1}); }
            for (int j = loops[i][1] + 1; j <= endSegID; j++) { flag = true; c9.push_back({variableIdx[ps(loops[i][0], j)], 1}); c9.push_back({variableIdx[ls(loops[i][0], j)], 1}); }
            if (flag) { for (auto& e : c9) put(e.first, e.second); put(variableIdx[ls(loops[i][0], loops[i][1])], -1); endRow(0, INF); }
        }
        for (int i = 0; i < numLoop; i++) {   // :4943-4974
            std::vector<std::pair<int, double>> c10;
            bool flag = false;
            for (int j = loops[i][0]; j < loops[i][1]; j++) { flag = true; c10.push_back({variableIdx[ls(loops[i][0], j)], 1}); }
            for (int j = loops[i][0] + 1; j <= loops[i][1]; j++) { flag = true; c10.push_back({variableIdx[ls(j, loops[i][1])], 1}); }
            if (flag) {
                for (auto& e : c10) put(e.first, e.second); put(variableIdx[ls(patterns[i][0], patterns[i][1])], 1); endRow(0, 2);
                for (auto& e : c10) put(e.first, e.second); put(variableIdx[ps(patterns[i][0], patterns[i][1])], 1); endRow(0, 2);
            }
        }
        for (int i = 0; i < numPat; i++) {   // :4977-5008
            std::vector<std::pair<int, double>> c10, c11;
            bool flag = false;
            for (int j = patterns[i][0]; j < patterns[i][1]; j++) { flag = true; c10.push_back({variableIdx[ls(patterns[i][0], j)], 1}); c11.push_back({variableIdx[ps(patterns[i][0], j)], 1}); }
            for (int j = patterns[i][0] + 1; j <= patterns[i][1]; j++) { flag = true; c10.push_back({variableIdx[ps(j, patterns[i][1])], 1}); c11.push_back({variableIdx[ls(j, patterns[i][1])], 1}); }
            if (flag) {
                const int key = variableIdx[ps(patterns[i][0], patterns[i][1])];
                for (auto& e : c10) put(e.first, e.second); put(key, 1); endRow(0, 2);
                for (auto& e : c11) put(e.first, e.second); put(key, 1); endRow(0, 2);
            }
        }
        double maxCN = 0;
        for (auto* sg : segs) maxCN += sg->cn;
        for (int i = 0; i < numPat; i++) { int c = variableIdx[ps(patterns[i][0], patterns[i][1])]; m.colLo[c] = 0; m.colUp[c] = 1; }
        for (int i = 0; i < numLoop; i++) { int c = variableIdx[ls(loops[i][0], loops[i][1])]; m.colLo[c] = 0; m.colUp[c] = maxCN; }
    }
    int cnt = (numElements + numSegments * 2 * numGraphs) * 2;
    for (auto& kv : variableIdx) kv.second = kv.second % numComp;
    for (size_t i = 0; i < evolution.size(); i++)
        for (int j : evolution[i]) {
            for (int k = 0; k < numPat; k++) {
                const int c = variableIdx[ps(patterns[k][0], patterns[k][1])];
                put(c + numComp * (int)i, 1); put(c + numComp * j, -1); put(cnt / 2, 1); endRow(0, INF); cnt++;
                put(c + numComp * (int)i, 1); put(c + numComp * j, -1); put(cnt / 2, -1); endRow(-INF, 0); cnt++;
            }
            for (int k = 0; k < numLoop; k++) {
                const int c = variableIdx[ls(loops[k][0], loops[k][1])];
                put(c + numComp * (int)i, 1); put(c + numComp * j, -1); put(cnt / 2, 1); endRow(0, INF); cnt++;
                put(c + numComp * (int)i, 1); put(c + numComp * j, -1); put(cnt / 2, -1); endRow(-INF, 0); cnt++;
            }
        }
    for (int i = 0; i < numEpsilons; i++) { m.colLo[numElements + i] = 0; m.colUp[numElements + i] = INF; }
    for (int i = 0; i < numVariables; i++) m.obj[i] = i < numElements ? 0 : 1;
    m.numInt = numElements;
}

}  // namespace oracle
