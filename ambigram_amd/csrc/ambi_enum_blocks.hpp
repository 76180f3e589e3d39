// ambi_enum_blocks.hpp -- order-table enumeration by SUFFIX BLOCKS + a BLOCK DIRECTORY (the fast path of the
// enumerate stage).
//
// Observation: all completions of an order ideal J form a block of cnt[J] consecutive rows of the table whose last
// K - |J| columns depend on J only, and whose first |J| columns are one common prefix.  Walking the order tree from
// the top and stopping at the first ideal with cnt <= block_max (a BLOCK ROOT) cuts the table into nB blocks.
//
// Built ONCE per unit (build_block_image, one workgroup, ambi_blocks_build_kernel) into a position-independent
// image that is parked in HBM and copied into LDS by every emitting workgroup:
//   * suffix rows: for every possible root J the cnt[J] completions of J as FULL-WIDTH rows (NW dwords, bytes < |J|
//     zero, bytes >= K 0xFF), in lexicographic order;
//   * directory: for every block b in table order  { first row, LDS offset of its root's suffix rows, prefix words }.
//     Block b is found by unranking b over "blocks below an ideal" counts, its first row by summing the 64-bit
//     completion counts of the skipped siblings -- every block independently, so the directory is filled in parallel.
// Emission (emit_blocks_wave) then has no tree walk at all: a wave binary-searches the directory for its first row
// and streams block after block; lane j owns the j-th 16-byte group of the block's contiguous byte range, ORs four
// prefix dwords with four suffix dwords (LDS reads at consecutive addresses) and issues one fully coalesced 16-byte
// store.  Every byte of the table is written exactly once, in ascending address order per wave.
//
// The per-lane DFS of ambi_orders.hpp (enumerate_rows) stays as the general path for units whose image does not fit
// the LDS budget.  Both produce the reference's `orders` (LGM.cpp:3380-3409) byte for byte.
#pragma once
#include "ambi_orders.hpp"

namespace ambi {

constexpr bool kEmitRows = true;     // narrow rows leave one row per lane (emit_piece_rows); false: the 16-byte-group form for every width
constexpr int kBlockMaxLimit = 1024;   // upper bound of BatchArgs::block_max (keeps in-block dword offsets < 2^15)

// wave-uniform value: broadcast lane 0's copy so that the compiler keeps it in scalar registers
AMBI_HD int uni(int x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_readfirstlane(x);
#else
    return x;
#endif
}
AMBI_HD uint32_t uniu(uint32_t x) { return (uint32_t)uni((int)x); }
AMBI_HD uint64_t uniu64(uint64_t x) { return ((uint64_t)uniu((uint32_t)(x >> 32)) << 32) | uniu((uint32_t)x); }

// x / N for the small x of block emission (in-block dword offsets, x <= kBlockMaxLimit * N) without the 32x32
// multiply-high of a generic constant division: halve even N, 16-bit multiply-shift by 2^16/N for odd N.  The
// static_asserts below prove exactness over the whole range for every row width the engine instantiates.
template <int N>
AMBI_HD constexpr uint32_t small_div(uint32_t x) {
    if constexpr (N == 1) return x;
    else if constexpr ((N & 1) == 0) return small_div<N / 2>(x >> 1);
    else return ((x & 0xFFFFu) * (uint32_t)(65536 / N + 1)) >> 16;
}
template <int N>
AMBI_HD constexpr uint32_t small_mod(uint32_t x) {
    if constexpr ((N & (N - 1)) == 0) return x & (uint32_t)(N - 1);
    else return x - small_div<N>(x) * (uint32_t)N;
}
template <int N>
constexpr bool small_div_exact() {
    for (uint32_t x = 0; x <= (uint32_t)(kBlockMaxLimit * N + 256); x++)
        if (small_div<N>(x) != x / N || small_mod<N>(x) != x % N) return false;
    return true;
}
static_assert(small_div_exact<3>() && small_div_exact<5>() && small_div_exact<6>(), "small_div must be exact over the in-block offset range");
static_assert(small_div_exact<7>() && small_div_exact<9>(), "small_div must be exact over the in-block offset range");
static_assert(small_div_exact<10>() && small_div_exact<11>(), "small_div must be exact over the in-block offset range");
static_assert(small_div_exact<12>(), "small_div must be exact over the in-block offset range");

// header of a unit's image, kept in HBM (BatchArgs::block_hdr, 8 ints per unit)
struct BlockImageHeader { int32_t fits, nB, suf_words, image_bytes, nI, nC, block_max, pad; };

// Image layout (dwords): directory entries of S dwords  { row0, soff, pw[..] }  for b = 0..nB-1, one
// sentinel dword (= R) in the row0 slot of entry nB, padding to 16 bytes, then the suffix rows.  pw[x] = prefix word
// x % NW, so that the four words a lane needs, pw[k .. k+3] with k < NW, are consecutive.
// S = NW + 5 with the three wrap copies of the prefix words the 16-byte-group emission reads; rows of up to five dwords leave one
// row per lane (emit_piece_rows), which reads pw[0 .. NW) only: S = NW + 2 (K = 19: 20 instead of 32 bytes per block, 2.9 KB of
// a 27.8 KB image -- room on the CU for a second finish workgroup beside five enumerate workgroups).
AMBI_HD constexpr int dir_stride(int NW) { return (kEmitRows && NW <= 5) ? NW + 2 : NW + 5; }
AMBI_HD int64_t dir_words(int nB, int NW) { return ((int64_t)nB * dir_stride(NW) + 1 + 3) & ~int64_t(3); }

// automaton copy used while building (group memory)
struct BuildTables {
    uint64_t* avail;    // [nI]
    uint64_t* cnt64;    // [nI]  exact completion counts (first rows of the directory entries)
    uint32_t* nblk;     // [nI]  blocks below the ideal (1 for ideals with cnt <= block_max), saturated
    uint32_t* link;     // [nC]  child link with the child's (saturated) count: child | cnt16 << 16 -- one read per sibling
    uint16_t* cbase;    // [nI]
    uint16_t* child;    // [nC]
    uint16_t* cnt16;    // [nI]  completion counts saturated at 65535
    uint16_t* soff;     // [nI]  dword offset of the ideal's suffix rows, 0xFFFF when it is not a possible root
    uint16_t* roots;    // [nI]  ideal indices of the possible roots
    uint32_t* root_row; // [nI+1] prefix sums of their row counts
    uint8_t* depth;     // [nI]
    int32_t* misc;      // [4]   nRoots, suffix words, fits flag
};
AMBI_HD int64_t carve_build_tables(uint8_t* mem, int nI, int nC, BuildTables& B) {
    int64_t o = 0;
    B.avail = reinterpret_cast<uint64_t*>(mem + o); o += 8ll * nI;
    B.cnt64 = reinterpret_cast<uint64_t*>(mem + o); o += 8ll * nI;
    B.nblk = reinterpret_cast<uint32_t*>(mem + o); o += 4ll * nI;
    B.link = reinterpret_cast<uint32_t*>(mem + o); o += 4ll * nC;
    B.root_row = reinterpret_cast<uint32_t*>(mem + o); o += 4ll * (nI + 1);
    B.misc = reinterpret_cast<int32_t*>(mem + o); o += 16;
    B.cbase = reinterpret_cast<uint16_t*>(mem + o); o += 2ll * nI;
    B.child = reinterpret_cast<uint16_t*>(mem + o); o += 2ll * nC;
    B.cnt16 = reinterpret_cast<uint16_t*>(mem + o); o += 2ll * nI;
    B.soff = reinterpret_cast<uint16_t*>(mem + o); o += 2ll * nI;
    B.roots = reinterpret_cast<uint16_t*>(mem + o); o += 2ll * nI;
    B.depth = mem + o; o += nI;
    return (o + 15) & ~int64_t(15);
}

// Builds the image of one unit into `image` (group memory, image_bytes available) using `scratch` for the automaton
// copy.  Returns false when the unit has to take the general path.  SPMD over group g.
template <class G>
AMBI_HD bool build_block_image(const G& g, const IdealTable& T, int K, int NW, int64_t R, int block_max,
                               uint8_t* scratch, int64_t scratch_bytes, uint8_t* image, int64_t image_bytes,
                               BlockImageHeader& H, int64_t* clk = nullptr, bool with_directory = true) {
    clk_mark(g, clk, 31);
    const int nI = T.counter[0], nC = T.counter[1];
    H.fits = 0; H.nB = 0; H.suf_words = 0; H.image_bytes = 0; H.nI = nI; H.nC = nC; H.block_max = block_max; H.pad = 0;
    if (nC >= 65535 || nI >= 65535 || R <= 0 || R > 0xFFFFFFF0ll) return false;
    if (block_max < 1) block_max = 1;
    if (block_max > kBlockMaxLimit) block_max = kBlockMaxLimit;
    BuildTables B;
    if (carve_build_tables(scratch, nI, nC, B) > scratch_bytes) return false;
    const int S = dir_stride(NW);
    const int fb = row_bits(K);     // bits per node of a row
    uint32_t* img = reinterpret_cast<uint32_t*>(image);
    // automaton, levels and blocks-below counts as the prepare stage left them (ideal_build_and_count)
    for (int i = g.tid(); i < nI; i += g.size()) {
        const uint64_t c = T.a_cnt[i];
        B.cnt64[i] = c;
        B.avail[i] = T.a_avail[i];
        B.cbase[i] = (uint16_t)T.a_cbase[i];
        B.cnt16[i] = (uint16_t)(c > 65535 ? 65535 : c);
        B.soff[i] = 0xFFFF;
        B.nblk[i] = T.a_nblk[i];
        B.depth[i] = T.a_depth[i];
    }
    for (int i = g.tid(); i < nC; i += g.size()) B.child[i] = T.a_child[i];
    g.sync();
    clk_mark(g, clk, 13);
    for (int i = g.tid(); i < nC; i += g.size()) { const uint32_t c = B.child[i]; B.link[i] = c | ((uint32_t)B.cnt16[c] << 16); }
    // possible block roots: small ideals with a large parent (marked through the parents' child links), or the empty ideal
    for (int q = g.tid(); q < nI; q += g.size()) {
        if (B.cnt16[q] <= block_max) continue;
        const int k0 = B.cbase[q], k1 = k0 + popc64(B.avail[q]);
        for (int k = k0; k < k1; k++) { const int c = B.child[k]; if (B.cnt16[c] <= block_max) B.soff[c] = 0xFFFE; }
    }
    if (g.tid() == 0 && B.cnt16[0] <= block_max) B.soff[0] = 0xFFFE;
    g.sync();
    clk_mark(g, clk, 14);
    // root list in ideal order with the first row and the word offset of every root's suffix rows: two prefix sums
    // over the marks (the same numbers a serial walk over the ideals would produce)
    {
        int nr = 0, rows = 0, ok = 1;
        int64_t words = 0;
        for (int base = 0; base < nI; base += g.size()) {
            const int p = base + g.tid();
            const int is_root = (p < nI && B.soff[p] == 0xFFFE) ? 1 : 0;
            const int c = is_root ? (int)B.cnt16[p] : 0;
            int tr, tc;
            const int er = g.exscan_i32(is_root, &tr);
            const int ec = g.exscan_i32(c, &tc);
            if (is_root) {
                const int64_t w0 = words + (int64_t)ec * NW;
                if (w0 > 65000) ok = 0;                   // offsets are 16 bits wide
                else { B.soff[p] = (uint16_t)w0; B.roots[nr + er] = (uint16_t)p; B.root_row[nr + er] = (uint32_t)(rows + ec); }
            }
            nr += tr; rows += tc; words += (int64_t)tc * NW;
        }
        ok = g.any(ok == 0) ? 0 : 1;
        if (g.tid() == 0) {
            B.root_row[nr] = (uint32_t)rows;
            B.misc[0] = nr; B.misc[1] = (int32_t)(words > 0x7fffffff ? 0x7fffffff : words); B.misc[2] = ok;
        }
    }
    g.sync();
    clk_mark(g, clk, 15);
    const int nRoots = B.misc[0];
    const int64_t suf_words = B.misc[1];
    // without the directory (tables + suffix rows only) the emission walks the blocks itself (emit_blocks_dfs_wave): the
    // form for units with so many rows that one directory entry per block does not fit
    const int64_t nB = with_directory ? (int64_t)B.nblk[0] : 0;
    const int64_t dw = with_directory ? dir_words((int)(nB > (1 << 24) ? (1 << 24) : nB), NW) : 0;
    if (!B.misc[2] || nB > (1 << 24) || 4 * (dw + suf_words) > image_bytes) return false;
    uint32_t* suf = img + dw;
    // ---- directory: block b = b-th stop of the walk, unranked over nblk; first row from the exact 64-bit counts ----
    // (the image may live in HBM: whole words only, nothing is read back)
    for (int64_t b = g.tid(); b < nB; b += g.size()) {
        uint32_t* e = img + b * S;
        int i = 0, d = 0;
        uint32_t rem = (uint32_t)b;
        uint64_t row = 0;
        uint32_t w0 = 0, w1 = 0, w2 = 0;   // the first three prefix words (wrap copies)
        RowBits rb;                        // the prefix fields, fb bits each (ambi_orders.hpp: 5 up to 32 nodes, else 8)
        auto flush = [&](int wi, uint32_t w) { e[2 + wi] = w; if (wi == 0) w0 = w; else if (wi == 1) w1 = w; else if (wi == 2) w2 = w; };
        while (B.cnt16[i] > block_max) {
            uint64_t av = B.avail[i];
            int k = B.cbase[i];
            int chosen = 0, nxt = 0;
            while (av) {
                const int v = ctz64(av);
                av &= av - 1;
                nxt = B.child[k++];
                const uint32_t nb = B.nblk[nxt];
                if (rem < nb) { chosen = v; break; }
                rem -= nb;
                row += B.cnt64[nxt];
            }
            rb.put((uint32_t)chosen, fb, flush);
            i = nxt; d++;
        }
        {   // the partial word and the all-zero words behind the prefix
            int wi = rb.wi;
            if (rb.n != 0) { flush(wi, (uint32_t)rb.acc); wi++; }
            for (; wi < NW; wi++) e[2 + wi] = 0;
        }
        e[0] = (uint32_t)row;
        e[1] = B.soff[i];
        if (S == NW + 5) for (int x = NW; x < NW + 3; x++) { const int y = x % NW; e[2 + x] = y == 0 ? w0 : (y == 1 ? w1 : w2); }   // wrap copies
    }
    if (with_directory && g.tid() == 0) img[nB * S] = (uint32_t)R;
    clk_mark(g, clk, 29);
    // Node records for the walks below (round 4): { available nodes, first link, second link, index of the first link } of every ideal in ONE
    // 16-byte word, so that a level of a walk is one round trip to group memory where it was two or three (mask and link base, then the
    // links one after the other; the wide tier's ideals have at most two children).  The records overlay the 64-bit masks and counts at the
    // front of the tables, which nothing reads any more once the directory is written -- hence only with a directory (the directory-free
    // emission walks the tables itself) and up to 32 nodes.
    // (they are put together in the suffix-row area of the image, which is written only afterwards, and copied over the tables behind a barrier)
    const bool fast_rows = with_directory && K <= 32 && 4ll * nI <= suf_words;
    if (fast_rows) {
        g.sync();                                               // (the directory loop above reads avail / cnt64)
        for (int i = g.tid(); i < nI; i += g.size()) {
            const uint32_t av = (uint32_t)B.avail[i];
            const int k0 = B.cbase[i], nch = __builtin_popcount(av);
            store4(suf + 4 * i, av, nch >= 1 ? B.link[k0] : 0u, nch >= 2 ? B.link[k0 + 1] : 0u, (uint32_t)k0);
        }
        g.sync();
        uint32_t* rec = reinterpret_cast<uint32_t*>(B.avail);   // 16 bytes per ideal over avail[] and cnt64[] (8 + 8 bytes per ideal, one behind the other)
        for (int i = g.tid(); i < nI; i += g.size()) {
            uint32_t a, b, c, d;
            load4(suf + 4 * i, a, b, c, d);
            store4(rec + 4 * i, a, b, c, d);
        }
        g.sync();
    }
    // ---- suffix rows: row r of root p = r-th completion of p in lexicographic order, bytes in their final positions ----
    const int total_rows = (int)B.root_row[nRoots];
    for (int f = g.tid(); f < total_rows; f += g.size()) {
        int q = 0;
        { int hi = nRoots; while (hi - q > 1) { const int mid = (q + hi) >> 1; if (B.root_row[mid] <= (uint32_t)f) q = mid; else hi = mid; } }   // last root whose first row is <= f
        const int p = B.roots[q];
        int rr = f - (int)B.root_row[q];
        uint32_t* dst = suf + B.soff[p] + rr * NW;
        const int D = B.depth[p];
        RowBits rb;                           // fields < D stay zero
        rb.start(D * fb);
        for (int x = 0; x < rb.wi; x++) dst[x] = 0;
        auto flush = [&](int wi, uint32_t w) { dst[wi] = w; };
        int j = p, d = D;
        if (fast_rows) {
            const uint32_t* rec = reinterpret_cast<const uint32_t*>(B.avail);
            for (; d < K; d++) {
                uint32_t av, l0, l1, k0;
                load4(rec + 4 * j, av, l0, l1, k0);
                int chosen = __builtin_ctz(av), nxt = (int)(l0 & 0xFFFFu), cc = (int)(l0 >> 16);
                if (rr >= cc) {
                    rr -= cc; av &= av - 1;
                    chosen = __builtin_ctz(av); nxt = (int)(l1 & 0xFFFFu); cc = (int)(l1 >> 16);
                    if (rr >= cc) {   // a third or later child: the links one by one, as below
                        rr -= cc; av &= av - 1;
                        int k = (int)k0 + 2;
                        chosen = 0;
                        while (av) {
                            const int v = __builtin_ctz(av);
                            av &= av - 1;
                            const uint32_t lk = B.link[k++];
                            const int c3 = (int)(lk >> 16);
                            nxt = (int)(lk & 0xFFFFu);
                            if (rr < c3) { chosen = v; break; }
                            rr -= c3;
                        }
                    }
                }
                rb.put((uint32_t)chosen, fb, flush);
                j = nxt;
            }
        } else
        if (K <= 32) {   // 32-bit masks (low halves of the 64-bit ones), one packed read per sibling
            const uint32_t* av32 = reinterpret_cast<const uint32_t*>(B.avail);
            for (; d < K; d++) {
                uint32_t av = av32[2 * j];
                int k = B.cbase[j];
                int chosen = 0, nxt = 0;
                while (av) {
                    const int v = __builtin_ctz(av);
                    av &= av - 1;
                    const uint32_t lk = B.link[k++];
                    const int cc = (int)(lk >> 16);
                    nxt = (int)(lk & 0xFFFFu);
                    if (rr < cc) { chosen = v; break; }
                    rr -= cc;
                }
                rb.put((uint32_t)chosen, fb, flush);
                j = nxt;
            }
        }
        for (; d < K; d++) {
            uint64_t av = B.avail[j];
            int k = B.cbase[j];
            int chosen = 0, nxt = 0;
            while (av) {
                const int v = ctz64(av);
                av &= av - 1;
                const uint32_t lk = B.link[k++];
                const int cc = (int)(lk >> 16);
                nxt = (int)(lk & 0xFFFFu);
                if (rr < cc) { chosen = v; break; }
                rr -= cc;
            }
            rb.put((uint32_t)chosen, fb, flush);
            j = nxt;
        }
        rb.finish(NW, flush);                 // ones behind the K nodes
    }
    g.sync();
    clk_mark(g, clk, 30);
    H.fits = with_directory ? 1 : 2; H.nB = (int32_t)nB; H.suf_words = (int32_t)suf_words; H.image_bytes = (int32_t)(4 * (dw + suf_words));
    return true;
}

// One piece of the table: rows [cur, end) of the block whose first row is r0, whose suffix rows start at dword `so` of
// `suf` and whose prefix words (with three wrap copies) are pp[0 .. NW+2].  lane_lo/lane_hi: the lanes this call
// stands for ([lane, lane+1) on the GPU, [0, 64) in the host simulation).
// Narrow rows (up to five dwords: the packed rows of up to 32 nodes): one ROW per lane and step instead of one 16-byte group --
// the prefix words are the same for every row of the piece and stay in registers, a row is NW suffix dwords ORed with them and
// stored as one NW-dword store; consecutive lanes write consecutive rows, so a wavefront's store covers 64 * 4 NW contiguous bytes.
template <int NW>
AMBI_HD void emit_piece_rows(const uint32_t* suf, uint32_t so, uint32_t r0, uint32_t cur, uint32_t end, const uint32_t* pp, uint32_t* table,
                             int lane_lo, int lane_hi) {
    uint32_t pw[NW];
#pragma unroll
    for (int w = 0; w < NW; w++) pw[w] = pp[w];
    const uint32_t* sp = suf + so + (cur - r0) * NW;
    uint32_t* out = table + (int64_t)cur * NW;
    const int n = (int)(end - cur);
    for (int base = 0; base < n; base += 64) {
        for (int lane = lane_lo; lane < lane_hi; lane++) {
            const int r = base + lane;
            if (r < n) {
                uint32_t v[NW];
#pragma unroll
                for (int w = 0; w < NW; w++) v[w] = sp[r * NW + w] | pw[w];
#pragma unroll
                for (int w = 0; w < NW; w++) out[r * NW + w] = v[w];
            }
        }
    }
}

template <int NW>
AMBI_HD void emit_piece(const uint32_t* suf, uint32_t so, uint32_t r0, uint32_t cur, uint32_t end, const uint32_t* pp, uint32_t* table,
                        int lane_lo, int lane_hi) {
    if constexpr (NW <= 5) { if (kEmitRows) { emit_piece_rows<NW>(suf, so, r0, cur, end, pp, table, lane_lo, lane_hi); return; } }
    const int n = (int)(end - cur) * NW;                 // dwords of this piece (starts at a row boundary)
    const int64_t g0 = (int64_t)cur * NW;
    int head = (int)((-g0) & 3); if (head > n) head = n;   // dwords before the first 16-byte boundary
    const int body_end = head + ((n - head) & ~3);
    const uint32_t* sp = suf + so + (cur - r0) * NW;     // suffix dword of in-piece offset 0
    uint32_t* out = table + g0;
    for (int lane = lane_lo; lane < lane_hi; lane++) {
        if (lane < head) out[lane] = sp[lane] | pp[lane];                        // lane < 4 <= NW + 3
        const int rel = body_end + lane;
        if (rel < n) out[rel] = sp[rel] | pp[small_mod<NW>((uint32_t)rel)];
    }
    for (int base = head; base < body_end; base += 256) {
        for (int lane = lane_lo; lane < lane_hi; lane++) {
            const int rel = base + 4 * lane;
            if (rel < body_end) {
                const uint32_t k = small_mod<NW>((uint32_t)rel);
                store4(out + rel, sp[rel] | pp[k], sp[rel + 1] | pp[k + 1], sp[rel + 2] | pp[k + 2], sp[rel + 3] | pp[k + 3]);
            }
        }
    }
}

// Rows [rlo, rhi) of one unit's table, written by ONE wave from the unit's image (`img`, nB blocks).  `table` = the
// unit's rows as dwords (16-byte aligned).
// `part` / `parts`: the blocks of the range are dealt round-robin to `parts` waves (block b of the range goes to wave
// b mod parts), so that the waves of a workgroup write ONE compact window of the table that moves forward, instead of
// `parts` separate streams a quarter of the range apart (measured on a pure store stream, profiles/tools/hbm_write_pat.hip:
// 5.70 -> 5.98 TB/s).  parts = 1: the whole range.
template <int NW>
AMBI_HD void emit_blocks_wave(const uint32_t* img, int nB, uint32_t rlo, uint32_t rhi, uint32_t* table, int lane_lo, int lane_hi,
                              int part = 0, int parts = 1) {
    if (rlo >= rhi || nB <= 0) return;
    constexpr int S = dir_stride(NW);
    const uint32_t* suf = img + dir_words(nB, NW);
    int b = 0;
    {   // last block whose first row is <= rlo
        int lo = 0, hi = nB;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (uniu(img[mid * S]) <= rlo) lo = mid; else hi = mid; }
        b = lo;
    }
    for (b += part; b < nB; b += parts) {
        const uint32_t* e = img + b * S;
        const uint32_t r0 = uniu(e[0]), r1 = uniu(e[S]), so = uniu(e[1]);
        if (r0 >= rhi) break;
        const uint32_t cur = r0 > rlo ? r0 : rlo;        // the first / last block of the range may be cut
        const uint32_t end = r1 < rhi ? r1 : rhi;
        emit_piece<NW>(suf, so, r0, cur, end, e + 2, table, lane_lo, lane_hi);
    }
}

// The same rows without a directory: the wave walks the blocks itself.  A block is a root of the cut (an ideal with at
// most block_max completions whose parent has more); the walk is the depth-first traversal of the order tree above the
// roots in child order -- unrank the first row once (exact 64-bit counts), then per block: emit, climb to the first
// ancestor with a further child, take it, descend along first children to the next root.  Everything about the walk is
// uniform over the wave (every lane reads the same table entries); per wave it keeps the ideals on the current path
// (`stack`, one uint16 per depth) and the prefix words with their three wrap copies (`pw`, NW + 3 dwords) in group
// memory.  One step of the walk costs a handful of dependent reads per ~block_max rows, which is what makes this form
// slower than the directory when the directory fits, and the only block form when it does not (R beyond ~10^5 rows).
template <int NW>
AMBI_HD void emit_blocks_dfs_wave(const BuildTables& B, const uint32_t* suf, int K, int block_max, uint32_t rlo, uint32_t rhi, uint32_t* table,
                                  uint16_t* stack, uint32_t* pw, int lane_lo, int lane_hi) {
    if (rlo >= rhi) return;
    // every lane of the wave runs this bookkeeping with identical values; the stores to the wave's own stack / pw slots
    // are the same from all of them
    const int fb = row_bits(K);     // bits per node of a row (a field may straddle two words)
    const uint32_t fmask = (1u << fb) - 1u;
    auto put_word = [&](int wi, uint32_t w) { pw[wi] = w; if (wi < 3) pw[NW + wi] = w; };
    auto set_byte = [&](int d, uint32_t v) {
        const int bit = d * fb, wi = bit >> 5, sh = bit & 31;
        put_word(wi, (uniu(pw[wi]) & ~(fmask << sh)) | (v << sh));
        if (sh > 32 - fb) put_word(wi + 1, (uniu(pw[wi + 1]) & ~(fmask >> (32 - sh))) | (v >> (32 - sh)));
    };
    auto get_byte = [&](int d) {
        const int bit = d * fb, wi = bit >> 5, sh = bit & 31;
        uint32_t v = uniu(pw[wi]) >> sh;
        if (sh > 32 - fb) v |= uniu(pw[wi + 1]) << (32 - sh);
        return (int)(v & fmask);
    };
    for (int x = 0; x < NW + 3; x++) pw[x] = 0;
    int i = 0, d = 0;
    uint64_t rem = rlo;
    while (uni(B.cnt16[i]) > block_max) {          // unrank rlo down to its root
        uint64_t av = uniu64(B.avail[i]);
        int k = B.cbase[i], chosen = 0, nxt = 0;
        while (av) {
            const int v = ctz64(av);
            av &= av - 1;
            nxt = uni(B.child[k++]);
            const uint64_t cc = uniu64(B.cnt64[nxt]);
            if (rem < cc) { chosen = v; break; }
            rem -= cc;
        }
        stack[d] = (uint16_t)i;
        set_byte(d, (uint32_t)chosen);
        i = nxt; d++;
    }
    uint32_t cur = rlo;
    uint32_t r0 = rlo - (uint32_t)rem;
    while (true) {
        const uint32_t r1 = r0 + uniu(B.cnt16[i]);
        const uint32_t end = r1 < rhi ? r1 : rhi;
        emit_piece<NW>(suf, uniu(B.soff[i]), r0, cur, end, pw, table, lane_lo, lane_hi);
        cur = end;
        if (cur >= rhi) break;
        // next block: first ancestor with a child behind the one taken, then first children down to a root
        int nxt = -1;
        while (d > 0) {
            d--;
            const int p = uni(stack[d]);
            const int v = get_byte(d);                                                  // the node taken at depth d
            const uint64_t av = uniu64(B.avail[p]);
            const uint64_t rest = v >= 63 ? 0ull : (av & ~((2ull << v) - 1ull));       // nodes behind it
            if (rest) {
                const int v2 = ctz64(rest);
                const int k = uni(B.cbase[p]) + popc64(av & ((1ull << v2) - 1ull));
                set_byte(d, (uint32_t)v2);
                nxt = uni(B.child[k]);
                d++;
                break;
            }
            set_byte(d, 0u);                                                            // bytes behind the prefix stay zero
        }
        if (nxt < 0) break;                     // no further block (cur < rhi <= R should not get here)
        i = nxt;
        while (uni(B.cnt16[i]) > block_max) {
            const uint64_t av = uniu64(B.avail[i]);
            stack[d] = (uint16_t)i;
            set_byte(d, (uint32_t)ctz64(av));
            i = uni(B.child[uni(B.cbase[i])]);
            d++;
        }
        r0 = cur;
    }
}

}  // namespace ambi
