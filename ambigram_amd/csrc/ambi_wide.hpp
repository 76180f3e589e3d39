// ambi_wide.hpp -- units with 64 .. 255 DAG nodes ("wide units"; up to 127 in round 3).
//
// The reference has no bound on the number of selected patterns / loops of a chromosome (constructDAG takes every solution
// column with a positive value, LGM.cpp:3276-3301).  The engine's fast path keeps a node set in one 64-bit word (lane j =
// node j in the DAG construction, 64-bit ideal masks in the lattice and the automaton), which is what rounds 1-2 refused
// above 63 nodes.  Wide units take this path instead: the same algorithms in their plain form over four-word node sets,
// everything in HBM, sized for correctness and not for speed (such units are rare and small in their order count):
//   construct_dag_wide   constructDAG (LGM.cpp:3276-3378) incl. the replay of the library sort in its memory form
//   lattice_wide         the order ideals in breadth-first order (every ideal's children in ascending node order), the
//                        completion counts, R
//   unrank_wide          order r of allTopologicalOrders (LGM.cpp:3380-3409: lexicographic) from the counts
// Behind that a wide unit is an ordinary unit: its order table is planned and written (a byte per node: 128-byte rows up to 127 nodes,
// 256-byte rows above), the scan for the
// first valid order is the parallel search that already serves units whose scan budget runs out (it reads the table),
// indelBFB / output junctions are the ordinary finish stages; the evaluation of an order reads the node records through
// the same eval_order, instantiated for the 256-entry record arrays.
// Limits (explicit statuses): 255 nodes (a row holds a node in a byte and 0xFF behind the last one), kWideIdealCap (4096) order ideals, kWideMaxOrders (2^20) orders per wide unit.
#pragma once
#include "ambi_group.hpp"
#include "ambi_orders.hpp"
#include "ambi_prepare.hpp"
#include "ambi_sort.hpp"

namespace ambi {

constexpr int kMaxNodesWide = 255;
constexpr int kWideNodeCap = 256;              // entries of the per-node arrays
constexpr int kWideWords = 4;                  // 64-bit words of a node set
constexpr int kWideIdealCap = 4096;            // order ideals of a wide unit
constexpr int kWideLinkCap = 4 * kWideIdealCap;
constexpr int64_t kWideMaxOrders = 1 << 20;    // orders of a wide unit (128- or 256-byte rows: a table of at most 256 MB, written by one thread per row)

struct WideSet { uint64_t w[kWideWords]; };
AMBI_HD WideSet wset_empty() { WideSet m; for (int k = 0; k < kWideWords; k++) m.w[k] = 0; return m; }
AMBI_HD bool wset_has(const WideSet& m, int v) { return ((m.w[v >> 6] >> (v & 63)) & 1ull) != 0; }
AMBI_HD WideSet wset_with(WideSet m, int v) { m.w[v >> 6] |= 1ull << (v & 63); return m; }
AMBI_HD bool wset_subset(const WideSet& a, const WideSet& of) { bool ok = true; for (int k = 0; k < kWideWords; k++) ok = ok && (a.w[k] & ~of.w[k]) == 0; return ok; }
AMBI_HD bool wset_eq(const WideSet& a, const WideSet& b) { bool ok = true; for (int k = 0; k < kWideWords; k++) ok = ok && a.w[k] == b.w[k]; return ok; }
AMBI_HD bool wset_meets(const WideSet& a, const WideSet& b) { bool any = false; for (int k = 0; k < kWideWords; k++) any = any || (a.w[k] & b.w[k]) != 0; return any; }
AMBI_HD int wset_count(const WideSet& a) { int c = 0; for (int k = 0; k < kWideWords; k++) c += popc64(a.w[k]); return c; }

// node records of a wide unit as the evaluation reads them (the same member names as Dag)
struct WideDag {
    int32_t K;
    int32_t pat[kWideNodeCap][3];
    int32_t loop[kWideNodeCap][3];
};

// device working set of one wide unit (HBM)
struct WideUnit {
    WideDag dag;
    WideSet succ[kWideNodeCap], pred[kWideNodeCap];
    int32_t nI, nC;
    WideSet ikey[kWideIdealCap];                  // ideals in breadth-first (discovery) order, 0 = the empty ideal
    uint64_t cnt[kWideIdealCap];               // completions of every ideal (saturated at kCountSat)
    int32_t cbase[kWideIdealCap + 1];          // first child link of every ideal
    int32_t child[kWideLinkCap];               // child ideal of a link; the links of an ideal are in ascending node order
    uint8_t cnode[kWideLinkCap];               // the node the link appends
    int32_t hslot[2 * kWideIdealCap];          // open-addressing hash: ideal index or -1
    // scratch of construct_dag_wide
    uint64_t akey[kWideNodeCap], bkey[kWideNodeCap];
    int32_t idx[kWideNodeCap];
    Rec3 loops[kWideNodeCap];
    uint32_t sort_keys[kWideNodeCap], sort_stack[kSortStack + 8];
};

AMBI_HD void atomic_or_wset(WideSet* p, int v) { atomic_or_u64(&p->w[v >> 6], 1ull << (v & 63)); }

// LGM.cpp:3276-3378, the plain form of construct_dag_g (ambi_prepare.hpp) over four-word node sets.  Every thread returns the
// same status.
template <class G>
AMBI_HD int construct_dag_wide(const G& g, const Element* el, int K, int seg_base, WideUnit& X) {
    if (K > kMaxNodesWide) return ST_ERR_TOO_MANY_NODES;
    WideDag& D = X.dag;
    for (int i = g.tid(); i < K; i += g.size()) {
        X.akey[i] = dec_key((uint32_t)(el[i].a + seg_base));
        X.bkey[i] = dec_key((uint32_t)(el[i].b + seg_base));
    }
    g.sync();
    // node numbering = rank of the key in std::map order (LGM.cpp:3279)
    for (int i = g.tid(); i < K; i += g.size()) {
        const int li = el[i].is_loop;
        const uint64_t ka = X.akey[i], kb = X.bkey[i];
        int r = 0;
        for (int j = 0; j < K; j++) {
            const int lj = el[j].is_loop;
            const uint64_t ja = X.akey[j], jb = X.bkey[j];
            const bool less = (lj != li) ? (lj > li) : ((ja != ka) ? (ja < ka) : (jb < kb));   // key_less(j, i)
            r += (j != i && less) ? 1 : 0;
        }
        X.idx[r] = i;
    }
    g.sync();
    if (g.tid() == 0) D.K = K;
    for (int i = g.tid(); i < kWideNodeCap; i += g.size()) {
        X.succ[i] = wset_empty(); X.pred[i] = wset_empty();
        for (int c = 0; c < 3; c++) { D.pat[i][c] = 0; D.loop[i][c] = 0; X.loops[i].v[c] = 0; }
    }
    g.sync();
    for (int i = g.tid(); i < K; i += g.size()) {
        const Element& e = el[X.idx[i]];
        if (!e.is_loop) { D.pat[i][0] = e.a; D.pat[i][1] = e.b; D.pat[i][2] = e.cn; }
        else { X.loops[i].v[0] = e.a; X.loops[i].v[1] = e.b; X.loops[i].v[2] = e.cn; }
    }
    g.sync();
    // std::sort(node2loop, compareLoops) (LGM.cpp:3303): the replay of the library's algorithm (ambi_sort.hpp, memory form)
    for (int i = g.tid(); i < K; i += g.size()) X.sort_keys[i] = loop_sort_key(X.loops[i].v[0], X.loops[i].v[1], i);
    g.sync();
    int ub = 0;
    if (g.tid() == 0) {
        MemWords keys{X.sort_keys}, stk{X.sort_stack};
        bool u = false;
        libstdcxx_sort_keys(keys, K, &u, stk);
        ub = u ? 1 : 0;
    }
    ub = g.bcast_i32(ub, 0);
    g.sync();
    if (ub) return ST_ERR_REF_UB;
    for (int i = g.tid(); i < K; i += g.size()) { const Rec3 r = X.loops[X.sort_keys[i] & 255u]; D.loop[i][0] = r.v[0]; D.loop[i][1] = r.v[1]; D.loop[i][2] = r.v[2]; }
    g.sync();
    // p -> p and p -> l edges (LGM.cpp:3311-3338)
    for (int i = g.tid(); i < K; i += g.size()) {
        if (D.pat[i][0] == 0) continue;
        const int d1 = iabs(D.pat[i][0] - D.pat[i][1]);
        for (int j = 0; j < K; j++) {
            bool e = false;
            if (D.pat[j][0] != 0 && (D.pat[i][0] == D.pat[j][0] || D.pat[i][1] == D.pat[j][1]) && d1 > iabs(D.pat[j][0] - D.pat[j][1])) e = true;
            if (D.loop[j][0] != 0 && (D.pat[i][0] == D.loop[j][0] || D.pat[i][1] == D.loop[j][1]) && d1 > iabs(D.loop[j][0] - D.loop[j][1])) e = true;
            if (e) { X.succ[i] = wset_with(X.succ[i], j); atomic_or_wset(&X.pred[j], i); }
        }
    }
    g.sync();
    // loops in index order (LGM.cpp:3339-3377): the "inherited from a parent" rule reads pred[i] and succ[parent] as they
    // stand when loop i is processed, so the loop over i is sequential; "succ[parent] has j" is "pred[j] has parent"
    for (int i = 0; i < K; i++) {
        const int la = D.loop[i][0], lb = D.loop[i][1];
        if (la == 0) continue;
        const int d1 = iabs(la - lb);
        const WideSet pred_i = X.pred[i];      // (complete: nothing writes pred[i] in step i -- a node has no edge to itself)
        g.sync();
        for (int j = g.tid(); j < K; j += g.size()) {
            const WideSet mine = X.pred[j];
            const int pa = D.pat[j][0], pb = D.pat[j][1], qa = D.loop[j][0], qb = D.loop[j][1];
            bool edge = false;
            if (!wset_has(pred_i, j) && pa != 0 && (la == pa || lb == pb))                       // l -> p
                edge = d1 > iabs(pa - pb) || wset_meets(pred_i, mine);   //   own size, or inherited from a parent
            if (qa != 0 && (la == qa || lb == qb) && d1 > iabs(qa - qb)) edge = true;            // l1 -> l2
            if (edge) { X.pred[j] = wset_with(mine, i); atomic_or_wset(&X.succ[i], j); }
        }
        g.sync();
    }
    return ST_OK;
}

AMBI_HD uint32_t hash_wset(const WideSet& k) {
    uint32_t h = hash_mask(k.w[0]) * 31u + hash_mask(k.w[1] ^ 0x9E3779B97F4A7C15ull);
    for (int x = 2; x < kWideWords; x++) h = h * 31u + hash_mask(k.w[x] ^ (0x9E3779B97F4A7C15ull * (uint64_t)x));
    return h;
}

// Order ideals in breadth-first order, the child links of every ideal in ascending node order, completion counts.
// Returns ST_OK / ST_ERR_IDEALS_CAPACITY; *R_out = number of orders (saturated at kCountSat).
template <class G>
AMBI_HD int lattice_wide(const G& g, WideUnit& X, uint64_t* R_out) {
    const int K = X.dag.K;
    for (int i = g.tid(); i < 2 * kWideIdealCap; i += g.size()) X.hslot[i] = -1;
    g.sync();
    if (g.tid() == 0) { X.ikey[0] = wset_empty(); X.hslot[hash_wset(wset_empty()) & (2 * kWideIdealCap - 1)] = 0; X.nI = 1; X.nC = 0; }
    g.sync();
    int status = ST_OK;
    for (int i = 0; ; i++) {
        const int nI = g.bcast_i32(X.nI, 0);
        if (i >= nI || status != ST_OK) break;
        const WideSet I = X.ikey[i];
        if (g.tid() == 0) {
            // nodes that may be appended: not in I, every predecessor in I -- in ascending order, each a child link
            X.cbase[i] = X.nC;
            for (int v = 0; v < K && status == ST_OK; v++) {
                if (wset_has(I, v) || !wset_subset(X.pred[v], I)) continue;
                const WideSet key = wset_with(I, v);
                uint32_t h = hash_wset(key) & (2 * kWideIdealCap - 1);
                int found = -1;
                while (true) {
                    const int s = X.hslot[h];
                    if (s < 0) break;
                    if (wset_eq(X.ikey[s], key)) { found = s; break; }
                    h = (h + 1) & (2 * kWideIdealCap - 1);
                }
                if (found < 0) {
                    if (X.nI >= kWideIdealCap) { status = ST_ERR_IDEALS_CAPACITY; break; }
                    found = X.nI++;
                    X.ikey[found] = key;
                    X.hslot[h] = found;
                }
                if (X.nC >= kWideLinkCap) { status = ST_ERR_IDEALS_CAPACITY; break; }
                X.child[X.nC] = found; X.cnode[X.nC] = (uint8_t)v; X.nC++;
            }
        }
        status = g.bcast_i32(status, 0);
        g.sync();
    }
    if (status != ST_OK) return status;
    if (g.tid() == 0) {
        const int nI = X.nI;
        X.cbase[nI] = X.nC;
        for (int i = nI - 1; i >= 0; i--) {       // children come later in breadth-first order
            uint64_t c = 0;
            if (X.cbase[i] == X.cbase[i + 1]) c = (wset_count(X.ikey[i]) == K) ? 1 : 0;   // complete, or stuck (a cyclic relation)
            else for (int k = X.cbase[i]; k < X.cbase[i + 1]; k++) { c += X.cnt[X.child[k]]; if (c > kCountSat) c = kCountSat; }
            X.cnt[i] = c;
        }
    }
    g.sync();
    *R_out = X.cnt[0];
    return ST_OK;
}

// order r (0-based) of the unit in the reference's (lexicographic) order; row: K node bytes
AMBI_HD void unrank_wide(const WideUnit& X, uint64_t r, uint8_t* row) {
    const int K = X.dag.K;
    int i = 0;
    for (int d = 0; d < K; d++) {
        int nxt = -1;
        for (int k = X.cbase[i]; k < X.cbase[i + 1]; k++) {
            const int c = X.child[k];
            const uint64_t cc = X.cnt[c];
            if (r < cc) { row[d] = X.cnode[k]; nxt = c; break; }
            r -= cc;
        }
        if (nxt < 0) { row[d] = 0xFF; return; }   // (not reached for r < R)
        i = nxt;
    }
}

}  // namespace ambi
