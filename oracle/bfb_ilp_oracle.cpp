// oracle/bfb_ilp_oracle.cpp -- TEST INFRASTRUCTURE ONLY.
// Semantic restatement of LocalGenomicMap::BFB_ILP (LGM.cpp:4397-4752): the sparse row-major ILP the
// reference hands to CoinUtils before shelling out to `cbc`.  Rows are emitted in the reference's order with the
// reference's in-row entry order; "infinity" is DBL_MAX (OsiClp's getInfinity()).
// The LP/MPS *text* that CoinLpIO/CoinMpsIO write is third-party (coin-or-utils 2.11.6, environment.yml:15-20) and
// unpinned by any reference test -- parity for this stage is on (rowPtr, colIdx, val, bounds, objective, integrality).
#include <cfloat>
#include <cmath>
#include <cstdlib>
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

}  // namespace oracle
