// ambi_orders.hpp -- all topological orders of the BFB DAG, in the reference's order.
//
// LocalGenomicMap::allTopologicalOrders (LGM.cpp:3380-3409, driven by localhap.cpp:237-254) is a recursive
// DFS that always tries the lowest-numbered unvisited in-degree-0 node first, so it emits the linear extensions
// of the DAG in LEXICOGRAPHIC order of their node sequences and stores all R of them (R x K ints).
//
// MI355X design: instead of one serial DFS the engine
//   1. builds the lattice of order ideals (down-sets) of the DAG by level-synchronous frontier expansion and
//      counts, for every ideal I, the number cnt[I] of ways to complete it  (ideal_build / ideal_count),
//   2. gives every lane a contiguous block of ranks: the lane UNRANKS its first order from the counts
//      (order_unrank) and then steps through its block with the lexicographic successor (order_next),
//   3. stages rows through LDS and writes the R x K uint8 table with coalesced stores (kernel side).
// The table is byte-identical to the reference's `orders` vector (row r = r-th order the reference pushes).
#pragma once
#include "ambi_common.hpp"
#include "ambi_group.hpp"

namespace ambi {

constexpr uint64_t kEmptyKey = ~0ull;     // K <= 63, so no ideal mask equals this
constexpr uint64_t kCountSat = 1ull << 62;

// ---- atomics usable from both builds ----
AMBI_HD uint64_t atomic_cas_u64(uint64_t* p, uint64_t expected, uint64_t desired) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint64_t)atomicCAS((unsigned long long*)p, (unsigned long long)expected, (unsigned long long)desired);
#else
    uint64_t old = *p;
    if (old == expected) *p = desired;
    return old;
#endif
}
AMBI_HD int atomic_add_i32(int* p, int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return atomicAdd(p, v);
#else
    int old = *p; *p = old + v; return old;
#endif
}

// nodes that may be appended to the ideal I: not in I, all predecessors in I
AMBI_HD uint64_t avail_mask(const uint64_t* pred, int K, uint64_t I) {
    uint64_t rem = ~I & (K >= 64 ? ~0ull : ((1ull << K) - 1));
    uint64_t out = 0;
    while (rem) {
        int v = __builtin_ctzll(rem);
        rem &= rem - 1;
        if ((pred[v] & ~I) == 0) out |= (1ull << v);
    }
    return out;
}

AMBI_HD uint32_t hash_mask(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (uint32_t)k;
}

// Ideal table of one unit: open addressing, capacity `cap` (power of two).
struct IdealTable {
    uint64_t* keys;    // [cap]   ideal masks, kEmptyKey = free
    uint64_t* cnt;     // [cap]   completions of the ideal
    int32_t* lvl;      // [cap/2] slots in discovery order, level by level
    int32_t* lvl_off;  // [kMaxNodes + 2]
    int32_t* counter;  // [1] number of discovered ideals
    int cap;
};

AMBI_HD int ideal_lookup(const IdealTable& T, uint64_t key) {
    uint32_t h = hash_mask(key) & (uint32_t)(T.cap - 1);
    for (int probe = 0; probe < T.cap; probe++) {
        uint64_t k = T.keys[h];
        if (k == key) return (int)h;
        if (k == kEmptyKey) return -1;
        h = (h + 1) & (uint32_t)(T.cap - 1);
    }
    return -1;
}

// returns slot; *fresh = true when this call inserted the key
AMBI_HD int ideal_insert(const IdealTable& T, uint64_t key, bool* fresh) {
    uint32_t h = hash_mask(key) & (uint32_t)(T.cap - 1);
    *fresh = false;
    for (int probe = 0; probe < T.cap; probe++) {
        uint64_t k = T.keys[h];
        if (k == key) return (int)h;
        if (k == kEmptyKey) {
            uint64_t old = atomic_cas_u64(&T.keys[h], kEmptyKey, key);
            if (old == kEmptyKey) { *fresh = true; return (int)h; }
            if (old == key) return (int)h;
        }
        h = (h + 1) & (uint32_t)(T.cap - 1);
    }
    return -1;
}

// Level-synchronous frontier expansion over the ideal lattice + backward count.  SPMD over group g.
// Returns status; *R_out = number of topological orders (saturated at 2^62).
template <class G>
AMBI_HD int ideal_build_and_count(const G& g, const uint64_t* pred, int K, const IdealTable& T, uint64_t* R_out) {
    const int maxIdeals = T.cap / 2;
    for (int i = g.tid(); i < T.cap; i += g.size()) { T.keys[i] = kEmptyKey; T.cnt[i] = 0; }
    g.sync();
    if (g.tid() == 0) {
        bool fresh;
        int s = ideal_insert(T, 0ull, &fresh);
        T.lvl[0] = s; T.lvl_off[0] = 0; T.lvl_off[1] = 1; *T.counter = 1;
    }
    g.sync();
    int overflow = 0;
    int last_level = 0;
    for (int d = 0; d < K; d++) {
        int lo = T.lvl_off[d], hi = T.lvl_off[d + 1];
        if (hi == lo) break;
        for (int idx = lo + g.tid(); idx < hi; idx += g.size()) {
            uint64_t I = T.keys[T.lvl[idx]];
            uint64_t av = avail_mask(pred, K, I);
            while (av) {
                int v = __builtin_ctzll(av);
                av &= av - 1;
                bool fresh;
                int s = ideal_insert(T, I | (1ull << v), &fresh);
                if (s < 0) { overflow = 1; break; }
                if (fresh) {
                    int pos = atomic_add_i32(T.counter, 1);
                    if (pos >= maxIdeals) { overflow = 1; break; }
                    T.lvl[pos] = s;
                }
            }
        }
        g.sync();
        if (g.any(overflow != 0)) return ST_ERR_IDEALS_CAPACITY;
        if (g.tid() == 0) T.lvl_off[d + 2] = *T.counter;
        g.sync();
        last_level = d + 1;
    }
    // backward count.  Level K holds the single full ideal when the relation is acyclic.
    const uint64_t full = (K >= 64) ? ~0ull : ((1ull << K) - 1);
    if (last_level == K) {
        int sfull = ideal_lookup(T, full);
        if (g.tid() == 0 && sfull >= 0) T.cnt[sfull] = 1;
    }
    g.sync();
    for (int d = last_level - 1; d >= 0; d--) {
        int lo = T.lvl_off[d], hi = T.lvl_off[d + 1];
        for (int idx = lo + g.tid(); idx < hi; idx += g.size()) {
            int slot = T.lvl[idx];
            uint64_t I = T.keys[slot];
            if (I == full) continue;
            uint64_t av = avail_mask(pred, K, I);
            uint64_t c = 0;
            while (av) {
                int v = __builtin_ctzll(av);
                av &= av - 1;
                int s = ideal_lookup(T, I | (1ull << v));
                if (s >= 0) { c += T.cnt[s]; if (c > kCountSat) c = kCountSat; }
            }
            T.cnt[slot] = c;
        }
        g.sync();
    }
    int s0 = ideal_lookup(T, 0ull);
    *R_out = (s0 >= 0) ? T.cnt[s0] : 0;
    return ST_OK;
}

// r-th (0-based) topological order in lexicographic order -> ord[0..K).  One thread.
AMBI_HD bool order_unrank(const uint64_t* pred, int K, const IdealTable& T, uint64_t r, uint8_t* ord, int stride = 1) {
    uint64_t I = 0;
    for (int d = 0; d < K; d++) {
        uint64_t av = avail_mask(pred, K, I);
        bool found = false;
        while (av) {
            int v = __builtin_ctzll(av);
            av &= av - 1;
            int s = ideal_lookup(T, I | (1ull << v));
            uint64_t c = (s >= 0) ? T.cnt[s] : 0;
            if (r < c) { ord[d * stride] = (uint8_t)v; I |= (1ull << v); found = true; break; }
            r -= c;
        }
        if (!found) return false;
    }
    return true;
}

// lexicographic successor of ord (in place); false when ord was the last order.  One thread.
AMBI_HD bool order_next(const uint64_t* pred, int K, uint8_t* ord, int stride = 1) {
    uint64_t I = (K >= 64) ? ~0ull : ((1ull << K) - 1);
    for (int d = K - 1; d >= 0; d--) {
        int v = ord[d * stride];
        I &= ~(1ull << v);
        uint64_t higher = (v >= 63) ? 0ull : (~0ull << (v + 1));
        uint64_t cand = avail_mask(pred, K, I) & higher;
        if (cand) {
            int w = __builtin_ctzll(cand);
            ord[d * stride] = (uint8_t)w;
            I |= (1ull << w);
            for (int e = d + 1; e < K; e++) {
                uint64_t av = avail_mask(pred, K, I);
                if (!av) return false;   // cannot happen in a DAG
                int x = __builtin_ctzll(av);
                ord[e * stride] = (uint8_t)x;
                I |= (1ull << x);
            }
            return true;
        }
    }
    return false;
}

// the lexicographically LAST order (greedy highest available node): getBFB's orientation flip looks at
// whether the last order is valid (LGM.cpp:3691-3695)
AMBI_HD bool order_last(const uint64_t* pred, int K, uint8_t* ord) {
    uint64_t I = 0;
    for (int d = 0; d < K; d++) {
        uint64_t av = avail_mask(pred, K, I);
        if (!av) return false;
        int v = 63 - __builtin_clzll(av);
        ord[d] = (uint8_t)v;
        I |= (1ull << v);
    }
    return true;
}

}  // namespace ambi
