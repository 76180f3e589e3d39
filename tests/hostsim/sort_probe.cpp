// tests/hostsim/sort_probe.cpp -- TEST INFRASTRUCTURE ONLY.
// Exposes (a) the engine's restatement of libstdc++ std::sort (ambi_sort.hpp) and (b) the REAL std::sort with the
// reference's comparator (LGM.cpp:3267-3274) on the same records, so tests can compare the two permutations.
#include <algorithm>
#include <cstdlib>
#include <vector>
#include "../../ambigram_amd/csrc/ambi_sort.hpp"
#include "../../ambigram_amd/csrc/ambi_prepare.hpp"

static bool compareLoopsRef(std::vector<int> a, std::vector<int> b) {
    int diff1 = 0, diff2 = 0;
    if (a.size() > 0 && b.size() > 0) { diff1 = abs(a[0] - a[1]); diff2 = abs(b[0] - b[1]); }
    return (diff1 > diff2);
}

extern "C" {
// recs: n x 3 ints (a==0 -> empty slot). out_engine / out_std: n x 3 ints.
int hostsim_sort_both(const int* recs, int n, int* out_engine, int* out_std) {
    std::vector<ambi::Rec3> a(n);
    std::vector<std::vector<int>> v(n);
    for (int i = 0; i < n; i++) {
        for (int c = 0; c < 3; c++) a[i].v[c] = recs[3 * i + c];
        if (recs[3 * i] != 0) v[i] = {recs[3 * i], recs[3 * i + 1], recs[3 * i + 2]};
    }
    bool ub = false;
    uint32_t stack[ambi::kSortStack + 256];
    ambi::libstdcxx_sort_loops(a.data(), n, &ub, stack);
    std::sort(v.begin(), v.end(), compareLoopsRef);
    for (int i = 0; i < n; i++) {
        for (int c = 0; c < 3; c++) {
            out_engine[3 * i + c] = a[i].v[c];
            out_std[3 * i + c] = v[i].empty() ? 0 : v[i][c];
        }
    }
    return ub ? 1 : 0;
}
// std::map<std::string,int> key order vs the engine's key_less
int hostsim_key_less(int l1, int a1, int b1, int l2, int a2, int b2) { return ambi::key_less(l1, a1, b1, l2, a2, b2) ? 1 : 0; }
}
