// ambi_orders.hpp -- all topological orders of the BFB DAG, in the reference's order.
//
// LocalGenomicMap::allTopologicalOrders (LGM.cpp:3380-3409, driven by localhap.cpp:237-254) is a recursive
// DFS that always tries the lowest-numbered unvisited in-degree-0 node first, so it emits the linear extensions
// of the DAG in LEXICOGRAPHIC order of their node sequences and stores all R of them (R x K ints).
//
// MI355X design: instead of one serial DFS the engine
//   1. builds the lattice of order ideals (down-sets) of the DAG by level-synchronous frontier expansion (hash set
//      with atomicCAS, one frontier per level) and counts, for every ideal I, the number cnt[I] of ways to complete
//      it; the lattice is then frozen into a small AUTOMATON: per ideal index its available-node mask, its
//      completion count and the indices of its children  (ideal_build_and_count),
//   2. gives every lane a contiguous block of ranks: the lane UNRANKS its first order by walking the automaton
//      (order_unrank) and then steps through its block with the lexicographic successor, keeping the current row
//      packed in registers and per-depth (ideal, avail, node) stacks in LDS  (enumerate_rows),
//   3. writes the R x Kpad uint8 table with 16-byte stores straight from registers (4 rows per store group).
// Row r of the table = r-th order the reference pushes; columns K..Kpad-1 are 0xFF padding (Kpad = K rounded up
// to a multiple of 4 so that rows are dword aligned).
#pragma once
#include "ambi_common.hpp"
#include "ambi_group.hpp"

namespace ambi {

constexpr uint64_t kEmptyKey = ~0ull;     // K <= 63, so no ideal mask equals this
constexpr uint64_t kCountSat = 1ull << 62;
constexpr int kFirstRowStride = 64;       // bytes between the pre-unranked first orders of a unit (K <= 63)

// Row of the order table.  row_bits(K) bits per node -- 5 up to 32 nodes, 6 up to 63 -- node d in bits [d * bits, (d + 1) * bits) of
// the row's dwords (little end first: a field may straddle two dwords), the bits behind the K-th field all ones, rows a whole
// number of dwords: 12 bytes for K = 19 where rounds 1-3 wrote 20, 32 for K = 41 instead of 48.  The table is the reference's
// `orders` (LGM.cpp:3380-3409) in the engine's own layout: nothing but the engine's kernels and ambi_batch_unit_orders (which
// unpacks) ever reads it, and the block emission (ambi_enum_blocks.hpp) ORs prefix and suffix DWORDS, whatever the fields inside
// them are.  64..255 nodes (wide units, ambi_wide.hpp): one byte per node, 128 bytes up to 127 nodes, 256 above.
AMBI_HD int row_bits(int K) { return K <= 32 ? 5 : (K <= 63 ? 6 : 8); }
AMBI_HD bool row_packed(int K) { return K <= 63; }
AMBI_HD int row_stride(int K) { return K <= 63 ? 4 * ((K * row_bits(K) + 31) >> 5) : (K <= 127 ? 128 : 256); }
// dwords of the register form of a row in the general enumerate path (one byte per node there; packed when it is stored)
AMBI_HD int row_byte_words(int K) { return K <= 32 ? ((K + 3) >> 2) : (K <= 48 ? 12 : (K <= 63 ? 16 : 32)); }
// node d of a row (`row` = its first dword)
AMBI_HD int row_node(const uint32_t* row, int K, int d) {
    if (!row_packed(K)) return (int)reinterpret_cast<const uint8_t*>(row)[d];
    const int fb = row_bits(K), bit = d * fb, wi = bit >> 5, sh = bit & 31;
    uint32_t v = row[wi] >> sh;
    if (sh > 32 - fb) v |= row[wi + 1] << (32 - sh);
    return (int)(v & ((1u << fb) - 1u));
}
AMBI_HD int row_node(const uint8_t* row, int K, int d) { return row_node(reinterpret_cast<const uint32_t*>(row), K, d); }
// appends fields to a row being assembled dword by dword
struct RowBits {
    uint64_t acc = 0;     // bits not yet flushed (low end first)
    int n = 0;            // their number (< 32 after every put)
    int wi = 0;           // dword the low end of acc belongs to
    AMBI_HD void start(int bit) { wi = bit >> 5; n = bit & 31; acc = 0; }          // the row's bits below `bit` are zero
    template <class F> AMBI_HD void put(uint32_t v, int bits, F&& flush) {         // flush(wi, dword) for every completed dword
        acc |= (uint64_t)v << n;
        n += bits;
        if (n >= 32) { flush(wi, (uint32_t)acc); acc >>= 32; n -= 32; wi++; }
    }
    template <class F> AMBI_HD void finish(int total_words, F&& flush) {           // ones up to the end of the row
        while (wi < total_words) {
            const uint32_t ones = n >= 32 ? 0u : (0xFFFFFFFFu << n);
            flush(wi, (uint32_t)acc | ones);
            acc = 0; n = 0; wi++;
        }
    }
};

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

AMBI_HD int ctz64(uint64_t x) { return __builtin_ctzll(x); }
AMBI_HD int popc64(uint64_t x) { return __builtin_popcountll(x); }

// nodes that may be appended to the ideal I: not in I, all predecessors in I.  `pred` must have 64 readable entries
// (entries >= K are read and masked off).  Serial code on a lone wavefront is paid in taken branches (~25 cycles) and
// dependent memory round trips (~60 cycles from group memory), not in ALU operations (~4 cycles), so the loop is laid
// out as unrolled chunks of 8 nodes behind uniform guards: the loads of a chunk are independent and pipeline, a node
// costs three ALU operations ("has a predecessor outside I" -> its bit of the blocked mask).
AMBI_HD uint64_t avail_mask(const uint64_t* pred, int K, uint64_t I) {
    if (K <= 32) {
        const uint32_t ni = ~(uint32_t)I;
        uint32_t blocked = 0;
#define AMBI_AV8(B)                                                                                    \
        if (K > (B)) {                                                                                 \
            _Pragma("unroll") for (int v = (B); v < (B) + 8; v++) {                                    \
                const uint32_t x = (uint32_t)pred[v] & ni;                                             \
                blocked |= (x < 1u ? x : 1u) << v;                                                     \
            }                                                                                          \
        }
        AMBI_AV8(0) AMBI_AV8(8) AMBI_AV8(16) AMBI_AV8(24)
#undef AMBI_AV8
        const uint32_t kmask = K >= 32 ? ~0u : ((1u << K) - 1u);
        return (uint64_t)(~blocked & ni & kmask);
    }
    const uint64_t ni = ~I;
    uint64_t blocked = 0;
    for (int base = 0; base < K; base += 8) {
#pragma unroll
        for (int v = 0; v < 8; v++) {
            const uint64_t x = pred[(base + v) & 63] & ni;
            blocked |= (uint64_t)(x != 0) << ((base + v) & 63);
        }
    }
    const uint64_t kmask = K >= 64 ? ~0ull : ((1ull << K) - 1ull);
    return ~blocked & ni & kmask;
}

// two 32-bit multiplies (a 64-bit multiply is four quarter-rate instructions on the GPU, and the insert is on the
// critical path of every lattice level); the high half of the product is folded into the low bits the table uses
AMBI_HD uint32_t hash_mask(uint64_t k) {
    uint32_t h = (uint32_t)k * 0x9E3779B1u;
    h ^= (uint32_t)(k >> 32) * 0x85EBCA6Bu;
    return h ^ (h >> 15);
}

// The frozen lattice of one unit (HBM): ideals numbered 0..nI-1 in discovery (level) order, 0 = empty ideal.
struct IdealTable {
    int32_t* lvl_off;    // [kMaxNodes + 3]  first ideal index of every level (level = |ideal|)
    int32_t* counter;    // [2]      number of ideals, number of child links
    uint64_t* a_avail;   // [cap/2]  available-node mask per ideal index
    uint64_t* a_cnt;     // [cap/2]  completion count per ideal index
    int32_t* a_cbase;    // [cap/2+1] first child link per ideal index
    uint16_t* a_child;   // [child_cap] child ideal indices, ascending node order
    uint32_t* a_nblk;    // [cap/2]  emission blocks below the ideal (ambi_enum_blocks.hpp), for the run's block_max
    uint8_t* a_depth;    // [cap/2]  |ideal| = its level
    int cap, child_cap;
};

// Working memory of the lattice search (group-local memory for small lattices, HBM pools otherwise).
struct LatticeWork {
    uint64_t* keys;      // [cap]    open-addressing hash set of ideal masks, kEmptyKey = free
    int32_t* pos;        // [cap]    hash slot -> ideal index
    uint64_t* ikey;      // [cap/2]  ideal index -> mask
    uint64_t* cnt;       // [cap/2]  completions by ideal index
    int32_t* cbase;      // [cap/2+1]
    uint32_t* link;      // [link_cap] child links: hash slots while searching, ideal indices afterwards
    int32_t* lvl_off;    // [kMaxNodes + 3]
    int32_t* counter;    // [2]
    int cap, link_cap;
};

// read-only view of the automaton
struct AutoView {
    const uint64_t* avail;
    const uint64_t* cnt;
    const int32_t* cbase;
    const uint16_t* child;
    int nI;
};
AMBI_HD AutoView auto_view(const IdealTable& T) { return AutoView{T.a_avail, T.a_cnt, T.a_cbase, T.a_child, T.counter[0]}; }

// returns slot; *fresh = true when this call inserted the key
AMBI_HD int ideal_insert(const LatticeWork& W, uint64_t key, bool* fresh) {
    uint32_t h = hash_mask(key) & (uint32_t)(W.cap - 1);
    *fresh = false;
    for (int probe = 0; probe < W.cap; probe++) {
        // one round trip per probe: the compare-and-swap itself tells "free (now mine)", "already there" or "occupied"
        const uint64_t old = atomic_cas_u64(&W.keys[h], kEmptyKey, key);
        if (old == kEmptyKey) { *fresh = true; return (int)h; }
        if (old == key) return (int)h;
        h = (h + 1) & (uint32_t)(W.cap - 1);
    }
    return -1;
}

// Level-synchronous search over the ideal lattice, backward count, frozen automaton.  SPMD over group g.
//   search   : one thread per ideal of the current level: available nodes, one hash insert per child; the child's
//              hash slot is parked in the link array (child links of the level are laid out by a group prefix sum,
//              so link order = ideal order = the automaton's final order);
//   resolve  : every link slot -> ideal index (one parallel pass once all indices are assigned);
//   count    : cnt[I] = sum of cnt over I's links, deepest level first (no hashing);
//   freeze   : counts, child bases and 16-bit child indices to the table in HBM.
// Returns status; *R_out = number of topological orders (saturated at 2^62).
template <class G>
AMBI_HD int ideal_build_and_count(const G& g, const uint64_t* pred, int K, const LatticeWork& W, const IdealTable& T, uint64_t* R_out,
                                  uint8_t* first_rows = nullptr, int first_count = 0, int64_t* clk = nullptr, int block_max = 0) {
    int maxIdeals = W.cap / 2;
    if (maxIdeals > 65535) maxIdeals = 65535;
    for (int i = g.tid(); i < W.cap; i += g.size()) W.keys[i] = kEmptyKey;
    g.sync();
    if (g.tid() == 0) {
        bool fresh;
        int s = ideal_insert(W, 0ull, &fresh);
        W.pos[s] = 0; W.ikey[0] = 0ull; W.lvl_off[0] = 0; W.lvl_off[1] = 1; W.counter[0] = 1; W.counter[1] = 0;
    }
    g.sync();
    // The level bounds and the ideal counter are carried in registers (uniform over the group); the indices of the new
    // ideals of a step come from a flag rank over the group (no atomic, no round trip through the counter in memory).
    int overflow = 0, last_level = 0, links = 0;
    int lo = 0, hi = 1, count = 1;
    for (int d = 0; d <= K; d++) {
        if (hi == lo) break;
        for (int base = lo; base < hi; base += g.size()) {
            const int idx = base + g.tid();
            // lanes behind the end of the level run the same straight-line code on a valid entry and mask the result
            const bool act = idx < hi;
            const uint64_t I = W.ikey[act ? idx : lo];
            uint64_t av = avail_mask(pred, K, I);
            if (!act) av = 0;
            if (act) W.cnt[idx] = av;   // parked here until the search is over (no store to HBM inside the level loop)
            const int nch = popc64(av);
            int tot;
            const int ex = g.exscan_i32(nch, &tot);
            int k = links + ex;
            if (act) W.cbase[idx] = k;
            if (k + nch > W.link_cap) { overflow = 1; av = 0; }
            // step r inserts the r-th child of every ideal of the chunk
            while (g.any(av != 0)) {
                const bool on = av != 0;
                const uint64_t child = I | (av & (0ull - av));   // lowest available node appended
                av &= av - 1;
                // first probe in straight-line code: the compare-and-swap itself tells "free (now mine)", "already
                // there" or "occupied by another ideal"; only the last case (rare at load <= 1/2) enters the probe loop
                uint32_t h = hash_mask(child) & (uint32_t)(W.cap - 1);
                uint64_t old = child;
                if (on) old = atomic_cas_u64(&W.keys[h], kEmptyKey, child);
                bool fresh = on && old == kEmptyKey;
                bool miss = on && old != kEmptyKey && old != child;
                if (g.any(miss)) {
                    for (int probe = 1; probe < W.cap && miss; probe++) {
                        h = (h + 1) & (uint32_t)(W.cap - 1);
                        old = atomic_cas_u64(&W.keys[h], kEmptyKey, child);
                        if (old == kEmptyKey) { fresh = true; miss = false; }
                        else if (old == child) miss = false;
                    }
                    if (miss) { overflow = 1; av = 0; }   // table full
                }
                if (on && !miss) W.link[k++] = h;
                int nfresh;
                const int p = count + g.flag_exscan(fresh, &nfresh);
                if (fresh) {
                    if (p >= maxIdeals) { overflow = 1; av = 0; }
                    else { W.ikey[p] = child; W.pos[h] = p; }
                }
                count += nfresh;
            }
            links += tot;
        }
        if (g.any(overflow != 0)) return ST_ERR_IDEALS_CAPACITY;
        if (g.tid() == 0) W.lvl_off[d + 2] = count;
        g.sync();
        lo = hi; hi = count;
        last_level = d + 1;
    }
    if (g.tid() == 0) W.counter[0] = count;
    g.sync();
    clk_mark(g, clk, 26);
    const int nI = W.counter[0];
    if (links > T.child_cap) return ST_ERR_IDEALS_CAPACITY;
    if (g.tid() == 0) { W.cbase[nI] = links; W.counter[1] = links; }
    for (int k = g.tid(); k < links; k += g.size()) W.link[k] = (uint32_t)W.pos[W.link[k]];
    for (int p = g.tid(); p < nI; p += g.size()) T.a_avail[p] = W.cnt[p];
    g.sync();
    // backward count.  The full ideal (level K, present when the relation is acyclic) has one completion: itself.
    const uint64_t full = (K >= 64) ? ~0ull : ((1ull << K) - 1);
    for (int d = last_level - 1; d >= 0; d--) {
        const int lo = W.lvl_off[d], hi = W.lvl_off[d + 1];
        for (int idx = lo + g.tid(); idx < hi; idx += g.size()) {
            // completions, and in the same pass the emission blocks below the ideal (ambi_enum_blocks.hpp: an ideal with
            // at most block_max completions is one block, a larger one the sum over its children) -- the image build
            // reads them instead of repeating the level-by-level pass; W.pos is free once the links are resolved
            uint64_t c = 0;
            uint32_t nb = 0;
            if (W.ikey[idx] == full) c = 1;
            else {
                const int k1 = W.cbase[idx + 1];
                for (int k = W.cbase[idx]; k < k1; k++) {
                    const uint32_t l = W.link[k];
                    c += W.cnt[l]; if (c > kCountSat) c = kCountSat;
                    nb += (uint32_t)W.pos[l]; if (nb > (1u << 30)) nb = 1u << 30;
                }
            }
            W.cnt[idx] = c;
            W.pos[idx] = (int32_t)((c > (uint64_t)block_max) ? nb : 1u);
            T.a_depth[idx] = (uint8_t)d;
        }
        g.sync();
    }
    clk_mark(g, clk, 27);
    // freeze
    for (int p = g.tid(); p < nI; p += g.size()) { T.a_cnt[p] = W.cnt[p]; T.a_nblk[p] = (uint32_t)W.pos[p]; }
    if (T.a_cbase != W.cbase) for (int p = g.tid(); p <= nI; p += g.size()) T.a_cbase[p] = W.cbase[p];
    for (int k = g.tid(); k < links; k += g.size()) T.a_child[k] = (uint16_t)W.link[k];
    for (int d = g.tid(); d < kMaxNodes + 3; d += g.size()) T.lvl_off[d] = (d <= last_level + 1) ? W.lvl_off[d] : nI;
    for (int t = g.tid(); t < 2; t += g.size()) T.counter[t] = t == 0 ? nI : links;
    *R_out = W.cnt[0];
    clk_mark(g, clk, 28);
    // The first `first_count` orders (rows 0.. of the table, kFirstRowStride bytes apart), unranked here while the lattice
    // is still in group memory: the scan for the first valid order reads them instead of the order table, which takes
    // the enumerate kernel off its critical path.  Descent by counts over the links; the node of a step is the one bit
    // by which the child's mask exceeds the parent's.
    if (first_rows && first_count > 0 && W.cnt[0] > 0) {
        const uint64_t nfirst = W.cnt[0] < (uint64_t)first_count ? W.cnt[0] : (uint64_t)first_count;
        for (uint64_t n = g.tid(); n < nfirst; n += g.size()) {
            uint8_t* row = first_rows + n * kFirstRowStride;
            int i = 0;
            uint64_t r = n;
            for (int d = 0; d < K; d++) {
                int c = i;
                const int k1 = W.cbase[i + 1];
                for (int k = W.cbase[i]; k < k1; k++) {
                    c = (int)W.link[k];
                    const uint64_t cc = W.cnt[c];
                    if (r < cc) break;
                    r -= cc;
                }
                row[d] = (uint8_t)ctz64(W.ikey[c] & ~W.ikey[i]);
                i = c;
            }
        }
    }
    g.sync();
    return ST_OK;
}

// child of ideal i along node v (v must be available in i)
AMBI_HD int auto_child(const AutoView& A, int i, uint64_t av, int v) {
    return A.child[A.cbase[i] + popc64(av & ((1ull << v) - 1))];
}

// r-th (0-based) topological order in lexicographic order -> ord[d*stride], d in [0,K).  One thread.
AMBI_HD bool order_unrank(const AutoView& A, int K, uint64_t r, uint8_t* ord, int stride = 1) {
    int i = 0;
    for (int d = 0; d < K; d++) {
        uint64_t av = A.avail[i];
        int k = A.cbase[i];
        bool found = false;
        while (av) {
            int v = ctz64(av);
            av &= av - 1;
            int j = A.child[k++];
            uint64_t c = A.cnt[j];
            if (r < c) { ord[d * stride] = (uint8_t)v; i = j; found = true; break; }
            r -= c;
        }
        if (!found) return false;
    }
    return true;
}

// the lexicographically LAST order (greedy highest available node): getBFB's orientation flip looks at
// whether the last order is valid (LGM.cpp:3691-3695)
AMBI_HD bool order_last(const AutoView& A, int K, uint8_t* ord) {
    int i = 0;
    for (int d = 0; d < K; d++) {
        uint64_t av = A.avail[i];
        if (!av) return false;
        int v = 63 - __builtin_clzll(av);
        ord[d] = (uint8_t)v;
        i = auto_child(A, i, av, v);
    }
    return true;
}

// ------------------------------------------------------------------------------------------------
// Row enumeration: `nrows` consecutive orders starting at rank `first`, written as packed rows.
// ------------------------------------------------------------------------------------------------
// The current order packed 4 nodes per dword; doubles as the image of the output row.
template <int NW>
struct PackedRow {
    uint32_t w[NW];
    AMBI_HD void fill(uint32_t v) {
#pragma unroll
        for (int i = 0; i < NW; i++) w[i] = v;
    }
    AMBI_HD uint32_t word(int wi) const {
        uint32_t r = 0;
#pragma unroll
        for (int i = 0; i < NW; i++) r = (i == wi) ? w[i] : r;
        return r;
    }
    AMBI_HD void set_word(int wi, uint32_t v) {
#pragma unroll
        for (int i = 0; i < NW; i++) w[i] = (i == wi) ? v : w[i];
    }
    AMBI_HD uint32_t get(int d) const { return (word(d >> 2) >> ((d & 3) * 8)) & 0xFFu; }
    AMBI_HD void set(int d, uint32_t val) {
        const int sh = (d & 3) * 8;
        set_word(d >> 2, (word(d >> 2) & ~(0xFFu << sh)) | (val << sh));
    }
};

// Greedy-descent record of an ideal: [15:0] index of its first child (lowest available node appended),
// [23:16] that node, [24] the ideal has more than one available node (= a branch point of the DFS).
AMBI_HD uint32_t make_rec(uint64_t av, int first_child) {
    return (uint32_t)first_child | ((uint32_t)ctz64(av) << 16) | ((popc64(av) > 1 ? 1u : 0u) << 24);
}

// Automaton accessors.  GlobalAuto reads the 64-bit tables in HBM/L2; LdsAuto reads a compact copy (mask type M,
// greedy-descent records, 16-bit child bases and links) staged in the group's memory.
// mask arithmetic in the width of the mask type (32-bit ALU ops while K <= 32)
AMBI_HD int mctz(uint32_t x) { return __builtin_ctz(x); }
AMBI_HD int mctz(uint64_t x) { return __builtin_ctzll(x); }
AMBI_HD int mpopc(uint32_t x) { return __builtin_popcount(x); }
AMBI_HD int mpopc(uint64_t x) { return __builtin_popcountll(x); }
template <class M> AMBI_HD M mbelow(int v) { return (M)(((M)1 << v) - 1); }          // bits < v
template <class M> AMBI_HD M mabove(int v) { return (M)~((((M)2) << v) - 1); }       // bits > v  (v <= width-2)

struct GlobalAuto {
    typedef uint64_t mask_t;
    AutoView A;
    AMBI_HD uint64_t avail(int i) const { return A.avail[i]; }
    AMBI_HD int child(int i, uint64_t av, int v) const { return A.child[A.cbase[i] + mpopc(av & mbelow<uint64_t>(v))]; }
    AMBI_HD uint32_t rec(int i) const { return make_rec(A.avail[i], A.child[A.cbase[i]]); }
};
template <class M>
struct LdsAuto {
    typedef M mask_t;
    const M* av_;           // [nI]
    const uint32_t* rec_;   // [nI]
    const uint16_t* cb_;    // [nI]
    const uint16_t* ch_;    // [nC]
    AMBI_HD M avail(int i) const { return av_[i]; }
    AMBI_HD int child(int i, M av, int v) const { return ch_[cb_[i] + mpopc((M)(av & mbelow<M>(v)))]; }
    AMBI_HD uint32_t rec(int i) const { return rec_[i]; }
};
template <class M>
AMBI_HD int64_t lds_auto_bytes(int nI, int nC) { return (int64_t)nI * (int64_t)(sizeof(M) + 6) + 2ll * nC + 16; }

// Per-lane DFS stacks, only meaningful at BRANCH depths (positions that still have a larger available node):
// idx = ideal before that position, prev = next shallower branch depth (0xFF: none).  Depth-major layout so that
// the 64 lanes of one depth are contiguous: element d at [d*stride].
struct LaneStacks {
    uint16_t* idx;
    uint8_t* prev;
    int stride;
};

// store helpers: 16-byte groups where the destination allows, plain dwords otherwise
AMBI_HD void store4(uint32_t* dst, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
#if defined(__HIP_DEVICE_COMPILE__)
    *reinterpret_cast<uint4*>(dst) = make_uint4(a, b, c, d);
#else
    dst[0] = a; dst[1] = b; dst[2] = c; dst[3] = d;
#endif
}

// ... and a 16-byte read (group memory: one ds_read_b128)
AMBI_HD void load4(const uint32_t* src, uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint4 v = *reinterpret_cast<const uint4*>(src);
    a = v.x; b = v.y; c = v.z; d = v.w;
#else
    a = src[0]; b = src[1]; c = src[2]; d = src[3];
#endif
}

// Writes rows [first, first+nrows) of the order table.  `out` points at row `first` (16-byte aligned; rows are NW
// dwords).  Requires first + nrows <= R.  One thread; all lanes of a wave run it in lockstep on their own ranges.
//
// DFS without a backtracking loop: `top` is the deepest branch depth, every branch depth remembers the next
// shallower one, so the successor jumps straight to `top`, takes the next larger available node there and then
// descends greedily -- one greedy-descent record (one LDS read) per level.
template <int NW, class AUTO>
AMBI_HD void enumerate_rows(const AUTO& au, const AutoView& cntView, int K, uint64_t first, int nrows,
                            const LaneStacks& S, uint32_t* out) {
    // (NW = dwords of the register form, one byte per node; `out` = this lane's first row in the table, whose rows are
    // row_stride(K) bytes: the same for more than 32 nodes, 5 bits per node below -- packed when a row is stored)
    PackedRow<NW> row;
    row.fill(0xFFFFFFFFu);
    int top = -1;
    // ---- unrank the first order, recording the branch depths ----
    {
        int i = 0;
        uint64_t r = first;
        for (int d = 0; d < K; d++) {
            const uint64_t av = cntView.avail[i];
            int k = cntView.cbase[i];
            uint64_t a2 = av;
            int chosen = -1, j = 0;
            while (a2) {
                int v = ctz64(a2);
                a2 &= a2 - 1;
                j = cntView.child[k++];
                uint64_t c = cntView.cnt[j];
                if (r < c) { chosen = v; break; }
                r -= c;
            }
            if (chosen < 0) return;   // rank out of range (caller guarantees this cannot happen)
            if (a2) {                 // a larger available node remains at this depth
                S.idx[d * S.stride] = (uint16_t)i;
                S.prev[d * S.stride] = (uint8_t)(top < 0 ? 0xFF : top);
                top = d;
            }
            row.set(d, (uint32_t)chosen);
            i = j;
        }
    }
    // ---- lexicographic successor ----
    auto successor = [&]() {
        const int d = top;
        if (d < 0) return;         // was the last order
        typedef typename AUTO::mask_t M;
        const int i = S.idx[d * S.stride];
        const M av = au.avail(i);
        const int v = (int)row.get(d);
        const M cand = (M)(av & mabove<M>(v));
        const int w = mctz(cand);
        int nt = d;
        if ((M)(cand & (M)(cand - 1)) == 0) { int p = S.prev[d * S.stride]; nt = (p == 0xFF) ? -1 : p; }
        int j = au.child(i, av, w);
        int wi = d >> 2;
        uint32_t cur = row.word(wi);
        { const int sh = (d & 3) * 8; cur = (cur & ~(0xFFu << sh)) | ((uint32_t)w << sh); }
        for (int e = d + 1; e < K; e++) {
            const uint32_t rc = au.rec(j);
            if ((e & 3) == 0) { row.set_word(wi, cur); wi++; cur = 0xFFFFFFFFu; }
            { const int sh = (e & 3) * 8; cur = (cur & ~(0xFFu << sh)) | (((rc >> 16) & 0xFFu) << sh); }
            if (rc & (1u << 24)) {
                S.idx[e * S.stride] = (uint16_t)j;
                S.prev[e * S.stride] = (uint8_t)(nt < 0 ? 0xFF : nt);
                nt = e;
            }
            j = (int)(rc & 0xFFFFu);
        }
        row.set_word(wi, cur);
        top = nt;
    };
    if (row_packed(K)) {
        // packed fields: the general path is the rare one (units whose block image does not fit group memory), so a row is
        // simply re-packed from its byte form when it leaves, dword by dword
        const int onw = row_stride(K) / 4;
        constexpr int fb = NW <= 8 ? 5 : 6;         // (NW <= 8 <=> up to 32 nodes: row_bits(K), as a constant of the instantiation)
        constexpr uint32_t fmask = (1u << fb) - 1u;
        constexpr int kPW = NW <= 8 ? 6 : 13;      // packed dwords (+1 for the straddle of the last field): 5 up to 32 nodes, 12 up to 63
        for (int r0 = 0; r0 < nrows; r0++) {
            uint32_t p[kPW];
#pragma unroll
            for (int i = 0; i < kPW; i++) p[i] = 0xFFFFFFFFu;
#pragma unroll
            for (int e = 0; e < NW * 4 && e < 64; e++) {
                if (e < K) {
                    const uint32_t v = (row.w[e >> 2] >> ((e & 3) * 8)) & fmask;
                    const int bit = e * fb, wi = bit >> 5, sh = bit & 31;      // (constants once the loop is unrolled)
                    p[wi] = (p[wi] & ~(fmask << sh)) | (v << sh);
                    if (sh > 32 - fb) p[wi + 1] = (p[wi + 1] & ~(fmask >> (32 - sh))) | (v >> (32 - sh));
                }
            }
            uint32_t* dst = out + (size_t)r0 * onw;
#pragma unroll
            for (int k = 0; k < kPW - 1; k++) if (k < onw) dst[k] = p[k];
            if (r0 + 1 < nrows) successor();
        }
        return;
    }
    // rows leave in groups of four (4*NW dwords = NW 16-byte stores); every register index below is static
    for (int r0 = 0; r0 < nrows; r0 += 4) {
        PackedRow<NW> b0 = row, b1 = row, b2 = row, b3 = row;
        const int have = nrows - r0 < 4 ? nrows - r0 : 4;
        if (have > 1) { successor(); b1 = row; }
        if (have > 2) { successor(); b2 = row; }
        if (have > 3) { successor(); b3 = row; }
        uint32_t* dst = out + (size_t)r0 * NW;
        if (have == 4) {
            uint32_t c[4 * NW];
#pragma unroll
            for (int k = 0; k < NW; k++) { c[k] = b0.w[k]; c[NW + k] = b1.w[k]; c[2 * NW + k] = b2.w[k]; c[3 * NW + k] = b3.w[k]; }
#pragma unroll
            for (int k = 0; k < NW; k++) store4(dst + 4 * k, c[4 * k], c[4 * k + 1], c[4 * k + 2], c[4 * k + 3]);
            if (r0 + 4 < nrows) successor();
        } else {
#pragma unroll
            for (int k = 0; k < NW; k++) {
                dst[k] = b0.w[k];
                if (have > 1) dst[NW + k] = b1.w[k];
                if (have > 2) dst[2 * NW + k] = b2.w[k];
            }
        }
    }
}

}  // namespace ambi
