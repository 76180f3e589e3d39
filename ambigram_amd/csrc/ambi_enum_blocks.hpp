// ambi_enum_blocks.hpp -- order-table enumeration by SUFFIX BLOCKS (the fast path of ambi_enumerate_kernel).
//
// Observation: all completions of an order ideal J form a block of cnt[J] consecutive rows of the table whose last
// s = K - |J| columns depend on J only, and whose first |J| columns are one common prefix.  So the table is written
// block by block:
//   * an ideal is a BLOCK ROOT on a path when it is the first ideal with cnt <= kBlockMax on it; the suffix rows of
//     every possible root (cnt <= kBlockMax and a parent with cnt > kBlockMax, or the empty ideal) are unranked ONCE
//     per workgroup into an LDS table, stored in their final in-word byte positions (the depth of an ideal is fixed,
//     so the alignment of its suffix inside a row is fixed too);
//   * one WAVE-UNIFORM depth-first search walks the upper part of the order tree in lexicographic order (branch-depth
//     stack, greedy-descent records -- no per-lane state, no divergence) and stops at block roots;
//   * the rows of a block are produced data-parallel: lane j owns the j-th 16-byte group of the block's contiguous
//     byte range, ORs prefix words and suffix words (LDS reads) and issues one fully coalesced 16-byte store.
// Every byte of the table is written exactly once, in ascending address order per wave.
//
// The per-lane DFS of ambi_orders.hpp (enumerate_rows) stays as the general path for automata that do not fit the
// LDS budget.  Both produce the reference's `orders` (LGM.cpp:3380-3409) byte for byte.
#pragma once
#include "ambi_orders.hpp"

namespace ambi {

constexpr int kBlockMax = 256;   // largest block (rows) whose suffix table is kept in LDS

// wave-uniform value: broadcast lane 0's copy so that the compiler keeps it in scalar registers
AMBI_HD int uni(int x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_readfirstlane(x);
#else
    return x;
#endif
}

// LDS image of a unit for block emission (built cooperatively by the workgroup)
struct BlockTables {
    uint64_t* avail;    // [nI]  available-node masks
    uint32_t* rec;      // [nI]  greedy-descent records
    uint16_t* cbase;    // [nI]
    uint16_t* child;    // [nC]
    uint16_t* cnt16;    // [nI]  completion counts saturated at 65535
    uint16_t* soff;     // [nI]  word offset of the ideal's suffix rows in `suf`, 0xFFFF when it is not a block root
    uint8_t* depth;     // [nI]  |J|
    uint32_t* suf;      // suffix words
    int nI, nC, suf_words;
};

AMBI_HD bool bt_is_root(const BlockTables& B, int i) { return B.soff[i] != 0xFFFF; }
AMBI_HD int bt_child(const BlockTables& B, int i, uint64_t av, int v) {
    return B.child[B.cbase[i] + popc64(av & ((1ull << v) - 1))];
}

// Position-independent image: the arrays are laid out in `mem` as a pure function of (nI, nC); the image is built
// once per unit (ambi_blocks_build_kernel), parked in HBM and copied back into LDS by every emitting workgroup.
struct BlockImageHeader { int32_t fits, nI, nC, suf_words, image_bytes, pad[3]; };

AMBI_HD int64_t carve_block_tables(uint8_t* mem, int nI, int nC, BlockTables& B) {
    int64_t o = 0;
    B.avail = reinterpret_cast<uint64_t*>(mem + o); o += 8ll * nI;
    B.rec = reinterpret_cast<uint32_t*>(mem + o); o += 4ll * nI;
    B.cbase = reinterpret_cast<uint16_t*>(mem + o); o += 2ll * nI;
    B.child = reinterpret_cast<uint16_t*>(mem + o); o += 2ll * nC;
    B.cnt16 = reinterpret_cast<uint16_t*>(mem + o); o += 2ll * nI;
    B.soff = reinterpret_cast<uint16_t*>(mem + o); o += 2ll * nI;
    B.depth = mem + o; o += nI;
    o = (o + 15) & ~int64_t(15);
    B.suf = reinterpret_cast<uint32_t*>(mem + o);
    B.nI = nI; B.nC = nC;
    return o;
}

// Carve + fill the tables from the unit's automaton (T in HBM).  Returns false when they do not fit `mem_bytes`.
// SPMD over group g; ends with a sync.
template <class G>
AMBI_HD bool stage_block_tables(const G& g, const IdealTable& T, int K, int NW, uint8_t* mem, int64_t mem_bytes, BlockTables& B) {
    const int nI = T.counter[0], nC = T.counter[1];
    if (nC >= 65535 || nI >= 65535) return false;
    const int64_t o = carve_block_tables(mem, nI, nC, B);
    if (o > mem_bytes) return false;
    const int64_t suf_cap_words = (mem_bytes - o) / 4;
    for (int i = g.tid(); i < nI; i += g.size()) {
        const uint64_t av = T.a_avail[i];
        const uint64_t c = T.a_cnt[i];
        B.avail[i] = av;
        B.cbase[i] = (uint16_t)T.a_cbase[i];
        B.rec[i] = av ? make_rec(av, T.a_child[T.a_cbase[i]]) : 0u;
        B.cnt16[i] = (uint16_t)(c > 65535 ? 65535 : c);
        B.soff[i] = 0xFFFF;
    }
    for (int i = g.tid(); i < nC; i += g.size()) B.child[i] = T.a_child[i];
    for (int d = 0; d <= K; d++)
        for (int p = T.lvl_off[d] + g.tid(); p < T.lvl_off[d + 1] && p < nI; p += g.size()) B.depth[p] = (uint8_t)d;
    g.sync();
    // possible block roots: small ideals with a large parent (marked through the parents' child links), or the empty ideal
    for (int q = g.tid(); q < nI; q += g.size()) {
        if (B.cnt16[q] <= kBlockMax) continue;
        const int k0 = B.cbase[q], k1 = k0 + popc64(B.avail[q]);
        for (int k = k0; k < k1; k++) { const int c = B.child[k]; if (B.cnt16[c] <= kBlockMax) B.soff[c] = 0xFFFE; }
    }
    if (g.tid() == 0 && B.cnt16[0] <= kBlockMax) B.soff[0] = 0xFFFE;
    g.sync();
    // offsets (serial over the ideals: nI is small) -- every thread computes the same numbers
    int64_t words = 0;
    bool fits = true;
    for (int p = 0; p < nI; p++) {
        if (B.soff[p] != 0xFFFE) continue;
        const int stride = NW - (B.depth[p] >> 2);
        if (words + (int64_t)B.cnt16[p] * stride > suf_cap_words || words > 65000) { fits = false; break; }
        words += (int64_t)B.cnt16[p] * stride;
    }
    g.sync();
    if (!fits) return false;
    if (g.tid() == 0) {
        int64_t w = 0;
        for (int p = 0; p < nI; p++) {
            if (B.soff[p] != 0xFFFE) continue;
            B.soff[p] = (uint16_t)w;
            w += (int64_t)B.cnt16[p] * (NW - (B.depth[p] >> 2));
        }
    }
    B.suf_words = (int)words;
    g.sync();
    // suffix rows: row r of root p = r-th completion of p in lexicographic order, bytes in their final positions
    for (int p = 0; p < nI; p++) {
        if (B.soff[p] == 0xFFFF) continue;
        const int D = B.depth[p], wd = D >> 2, stride = NW - wd, c = B.cnt16[p];
        for (int r = g.tid(); r < c; r += g.size()) {
            uint8_t* db = reinterpret_cast<uint8_t*>(B.suf + B.soff[p] + r * stride);
            for (int d = 4 * wd; d < D; d++) db[d - 4 * wd] = 0;
            int j = p, rr = r;
            for (int d = D; d < K; d++) {
                uint64_t av = B.avail[j];
                int k = B.cbase[j];
                int chosen = 0, nxt = 0;
                while (av) {
                    int v = ctz64(av);
                    av &= av - 1;
                    nxt = B.child[k++];
                    int cc = B.cnt16[nxt];
                    if (rr < cc) { chosen = v; break; }
                    rr -= cc;
                }
                db[d - 4 * wd] = (uint8_t)chosen;
                j = nxt;
            }
            for (int d = K; d < NW * 4; d++) db[d - 4 * wd] = 0xFF;
        }
    }
    g.sync();
    return true;
}

// Wave-uniform walker over block roots in lexicographic order.
template <int NW>
struct BlockWalker {
    uint32_t pw[NW];     // prefix words (bytes >= D are zero)
    int J, D, top;       // current block root, its depth, deepest branch depth of the prefix (-1: none)
    AMBI_HD void set_byte(int d, uint32_t val) {
        const int wi = d >> 2, sh = (d & 3) * 8;
#pragma unroll
        for (int k = 0; k < NW; k++) pw[k] = (k == wi) ? ((pw[k] & ~(0xFFu << sh)) | (val << sh)) : pw[k];
    }
    AMBI_HD uint32_t get_byte(int d) const {
        uint32_t r = 0;
#pragma unroll
        for (int k = 0; k < NW; k++) r = (k == (d >> 2)) ? pw[k] : r;
        return (r >> ((d & 3) * 8)) & 0xFFu;
    }
    AMBI_HD void clear_from(int D_) {   // zero the bytes at positions >= D_
        const int wi = D_ >> 2;
        const uint32_t keep = (D_ & 3) ? ((1u << ((D_ & 3) * 8)) - 1) : 0u;
#pragma unroll
        for (int k = 0; k < NW; k++) pw[k] = (k > wi) ? 0u : ((k == wi) ? (pw[k] & keep) : pw[k]);
    }
};

// Position the walker on the block that contains rank `r` (exact 64-bit counts from the automaton in HBM);
// returns the row offset inside that block.  idx/prev: per-WAVE branch stacks (K entries) in group memory.
template <int NW>
AMBI_HD int bw_seek(BlockWalker<NW>& W, const BlockTables& B, const AutoView& V, uint64_t r, uint16_t* idx, uint8_t* prev, bool lane0) {
#pragma unroll
    for (int k = 0; k < NW; k++) W.pw[k] = 0;
    W.top = -1;
    int i = 0, d = 0;
    while (!bt_is_root(B, i)) {
        uint64_t av = V.avail[i];
        int k = V.cbase[i];
        int chosen = -1, j = 0;
        while (av) {
            int v = ctz64(av);
            av &= av - 1;
            j = V.child[k++];
            uint64_t c = V.cnt[j];
            if (r < c) { chosen = v; break; }
            r -= c;
        }
        if (chosen < 0) break;
        if (av) { if (lane0) { idx[d] = (uint16_t)i; prev[d] = (uint8_t)(W.top < 0 ? 0xFF : W.top); } W.top = d; }
        W.set_byte(d, (uint32_t)chosen);
        i = j; d++;
    }
    W.J = i; W.D = d;
    return (int)r;
}

// Advance to the next block root in lexicographic order; false after the last block.
template <int NW>
AMBI_HD bool bw_next(BlockWalker<NW>& W, const BlockTables& B, uint16_t* idx, uint8_t* prev, bool lane0) {
    const int d = W.top;
    if (d < 0) return false;
    const int i = uni(idx[d]);
    const uint64_t av = B.avail[i];
    const int v = (int)W.get_byte(d);
    const uint64_t cand = av & ~((2ull << v) - 1);
    const int w = ctz64(cand);
    int nt = d;
    if ((cand & (cand - 1)) == 0) { int p = uni(prev[d]); nt = (p == 0xFF) ? -1 : p; }
    int j = uni(bt_child(B, i, av, w));
    W.clear_from(d);
    W.set_byte(d, (uint32_t)w);
    int e = d + 1;
    while (uni(B.soff[j]) == 0xFFFF) {
        const uint32_t rc = (uint32_t)uni((int)B.rec[j]);
        W.set_byte(e, (rc >> 16) & 0xFFu);
        if (rc & (1u << 24)) {
            if (lane0) { idx[e] = (uint16_t)j; prev[e] = (uint8_t)(nt < 0 ? 0xFF : nt); }
            nt = e;
        }
        j = (int)(rc & 0xFFFFu);
        e++;
    }
    W.J = j; W.D = e; W.top = nt;
    return true;
}

// Rows [rlo, rhi) of one unit's table, written by ONE wave.  `table` = the unit's rows as dwords (16-byte aligned).
// lane_lo/lane_hi: the lanes this call stands for ([lane, lane+1) on the GPU, [0, 64) in the host simulation);
// wave_sync(): orders the wave's LDS traffic.  idx/prev/pw_lds: per-wave scratch in group memory (K, K, NW entries).
template <int NW, class SYNC>
AMBI_HD void emit_blocks_wave(const BlockTables& B, const AutoView& V, uint64_t rlo, uint64_t rhi, uint32_t* table,
                              uint16_t* idx, uint8_t* prev, uint32_t* pw_lds, int lane_lo, int lane_hi, const SYNC& wave_sync) {
    if (rlo >= rhi) return;
    const bool lane0 = (lane_lo == 0);
    BlockWalker<NW> W;
    int off = bw_seek<NW>(W, B, V, rlo, idx, prev, lane0);
    wave_sync();
    uint64_t cur = rlo;
    while (cur < rhi) {
        const int c = B.cnt16[W.J];
        uint64_t take = (uint64_t)(c - off);
        if (take > rhi - cur) take = rhi - cur;
        if (lane0) {
#pragma unroll
            for (int k = 0; k < NW; k++) pw_lds[k] = W.pw[k];
        }
        wave_sync();
        const int wd = W.D >> 2, stride = NW - wd;
        const uint32_t* suf = B.suf + B.soff[W.J];
        const int64_t row0 = (int64_t)cur - off;
        auto dword = [&](int64_t t) -> uint32_t {
            const int64_t row = t / NW;
            const int k = (int)(t - row * NW);
            uint32_t v = pw_lds[k];
            if (k >= wd) v |= suf[(int)(row - row0) * stride + (k - wd)];
            return v;
        };
        const int64_t g0 = (int64_t)cur * NW, g1 = (int64_t)(cur + take) * NW;
        int64_t a0 = (g0 + 3) & ~int64_t(3); if (a0 > g1) a0 = g1;
        int64_t a1 = g1 & ~int64_t(3); if (a1 < a0) a1 = a0;
        for (int lane = lane_lo; lane < lane_hi; lane++) {
            if (g0 + lane < a0) table[g0 + lane] = dword(g0 + lane);                 // head (< 4 dwords)
            if (a1 + lane < g1) table[a1 + lane] = dword(a1 + lane);                 // tail (< 4 dwords)
        }
        for (int64_t base = a0; base < a1; base += 256) {
            for (int lane = lane_lo; lane < lane_hi; lane++) {
                const int64_t t = base + 4 * lane;
                if (t < a1) store4(table + t, dword(t), dword(t + 1), dword(t + 2), dword(t + 3));
            }
        }
        cur += take;
        off = 0;
        if (cur < rhi) {
            wave_sync();
            if (!bw_next<NW>(W, B, idx, prev, lane0)) break;
            wave_sync();
        }
    }
}

}  // namespace ambi
