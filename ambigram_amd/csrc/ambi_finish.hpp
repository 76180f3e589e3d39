// ambi_finish.hpp -- post-assembly edits of the per-segment path and output-junction synthesis.
//
// Restates LocalGenomicMap::indelBFB (LGM.cpp:3746-3837) and the output junction loop of main()
// (localhap.cpp:267-289) as SPMD code over a workgroup.  The path (int16 signed local ids) sits in LDS.
//
// indelBFB walks the structural variants (SVs) in junction order, chains them into groups with a deque, and for
// every group looks its end vertices up in the path with std::find.  MI355X form:
//   * first/last-occurrence tables of every vertex (LDS, atomic min/max in one sweep over the path) answer every
//     std::find of a two-vertex group in O(1); only an actual edit (rare) rebuilds them;
//   * all SVs that cannot chain with a later SV ("singletons", decided once by an all-pairs test) are evaluated in
//     parallel against the tables; the first one (in junction order) that edits the path -- or that needs the
//     general deque chaining -- is found with one min-reduction, handled, and the sweep resumes behind it.  The
//     skipped SVs are exactly those for which the reference executes `continue`, so the result is the reference's.
#pragma once
#include "ambi_common.hpp"
#include "ambi_eval.hpp"
#include "ambi_group.hpp"
#include "ambi_orders.hpp"

namespace ambi {

struct OutJunc { int32_t u, v, count; };   // consecutive vertices (u,v) of a path that are not a reference adjacency

AMBI_HD void atomic_min_i32(int* p, int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMin(p, v);
#else
    if (v < *p) *p = v;
#endif
}
AMBI_HD void atomic_max_i32(int* p, int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMax(p, v);
#else
    if (v > *p) *p = v;
#endif
}

// std::find(path+from, path+to, val): first index in [from,to) or `to`.  from > to behaves like an empty range.
template <class G>
AMBI_HD int find_first(const G& g, const cell_t* path, int from, int to, int val) {
    int best = 0x7fffffff;
    for (int i = from + g.tid(); i < to; i += g.size())
        if (path[i] == val) { best = i; break; }
    best = g.min_i32(best);
    return best == 0x7fffffff ? to : best;
}

// Scratch for indel_bfb (group memory).
struct IndelScratch {
    int32_t* sv;        // [m]      qualifying junction indices, in junction order
    uint8_t* taken;     // [m]
    uint8_t* has_ext;   // [m]      a later SV could chain onto this one
    int32_t* grp;       // [2m+4]   deque storage of the general path
    int32_t* first;     // [2n+1]   first occurrence of vertex v at [v+n]
    int32_t* last;      // [2n+1]   last occurrence
};

// Two views of the per-segment path.  LdsPath: the cells themselves (group memory).  RunPath: the path as the runs it
// was expanded from -- breakpoint pair j covers positions [offs[j], offs[j+1]) with the values bkp[2j] + k (both the
// '+' run a, a+1, .. and the '-' run -|a|, -(|a|-1), .. count up by one); a cell is found by bisection over the run
// offsets.  The lean finish stage works on runs only and never materialises the cells in group memory.
struct LdsPath {
    const cell_t* p;
    AMBI_HD int at(int q) const { return p[q]; }
    // first q in [from, hi] with at(q) == val, -1 if none
    AMBI_HD int scan_for(int from, int hi, int val) const {
        for (int q = from; q <= hi; q++) if (p[q] == val) return q;
        return -1;
    }
};
struct RunPath {
    const cell_t* bkp;      // [2*np]
    const int32_t* offs;    // [np+1], offs[np] = P
    int np;
    AMBI_HD int at(int q) const {   // 0 <= q < P
        int lo = 0, hi = np;        // last run whose offset is <= q (empty runs share their offset with the next one)
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (offs[mid] <= q) lo = mid; else hi = mid; }
        return (int)bkp[2 * lo] + (q - offs[lo]);
    }
    // first q in [from, hi] with at(q) == val, -1 if none: ONE bisection for the run of `from`, then run by run (inside a run
    // the values count up by one, so the run holds val at most once, at a position that follows from its first value)
    AMBI_HD int scan_for(int from, int hi, int val) const {
        if (from > hi) return -1;
        int lo = 0, up = np;
        while (up - lo > 1) { const int mid = (lo + up) >> 1; if (offs[mid] <= from) lo = mid; else up = mid; }
        for (; lo < np; lo++) {
            const int o0 = offs[lo], o1 = offs[lo + 1];
            if (o0 > hi) break;
            if (o1 <= o0) continue;                               // empty run
            const int q = o0 + (val - (int)bkp[2 * lo]);          // where this run would hold val
            if (q >= o0 && q < o1 && q >= from && q <= hi) return q;
        }
        return -1;
    }
};

struct SingleEdit { int kind, a, b, empty; };   // kind 0: none, 1: erase [a,b), 2: duplicate [a,b) after b-1... (see apply); empty: kind 0 because the range to erase is empty (both ends found)

// Decision of LGM.cpp:3779-3818 for a two-vertex group (g0,g1), answered from the occurrence tables.
template <class PATH>
AMBI_HD SingleEdit eval_single(int g0, int g1, int n, const PATH& path, int P, const int32_t* first, const int32_t* last) {
    SingleEdit E{0, 0, 0, 0};
    const bool same = (g0 > 0) == (g1 > 0);
    const bool deletion = same && ((g0 > 0 && iabs(g0) < iabs(g1)) || (g0 < 0 && iabs(g0) > iabs(g1)));
    if (!same || deletion) {
        const int limit = same ? 3 : 5;
        int pos1 = P, pos2 = P;     // pos2: P = absent, -1 = present but farther than `limit`
        for (int attempt = 0; attempt < 2; attempt++) {
            int f0 = first[g0 + n];
            pos1 = f0 == 0x7fffffff ? P : f0;
            pos2 = P;
            if (pos1 != P && last[g1 + n] > pos1) {
                pos2 = -1;
                int hi = pos1 + limit < P - 1 ? pos1 + limit : P - 1;
                const int q = path.scan_for(pos1 + 1, hi, g1);
                if (q >= 0) pos2 = q;
            }
            if (attempt == 0 && (pos1 == P || pos2 == P)) { int t = g0; g0 = -g1; g1 = -t; continue; }
            break;
        }
        if (pos1 == P || pos2 == P || pos2 < 0) return E;
        if (pos2 - (pos1 + 1) <= 0) { E.empty = 1; return E; }   // erase of an empty range: no state change
        E.kind = 1; E.a = pos1 + 1; E.b = pos2;
        return E;
    }
    // duplication
    int pos1 = P, pos2 = P;
    for (int attempt = 0; attempt < 2; attempt++) {
        int f0 = first[g0 + n], f1 = first[g1 + n];
        pos1 = f0 == 0x7fffffff ? P : f0;
        pos2 = (f1 != 0x7fffffff && f1 < pos1) ? f1 : pos1;
        if (attempt == 0 && (pos1 == P || pos2 == pos1)) { int t = g0; g0 = -g1; g1 = -t; continue; }
        break;
    }
    if (pos1 == P || pos2 == pos1) return E;
    E.kind = 2; E.a = pos2; E.b = pos1 + 1;   // insert a copy of [a,b) at b
    return E;
}

// Front part of indelBFB shared by the full and the lean finish stage: the qualifying SVs in junction order
// (S.sv, S.taken cleared) and, per SV, whether a later one could chain onto it (S.has_ext).  Uses S.first / S.last as
// scratch.  Returns the number of SVs.
template <class G>
AMBI_HD int indel_collect(const G& g, int n, const JuncEnds* ends, int m, const IndelScratch& S) {
    // -- qualifying SVs in junction order (LGM.cpp:3750-3759), compacted with a group scan
    int nsv = 0;
    for (int base = 0; base < m; base += g.size()) {
        int ji = base + g.tid();
        int q = 0;
        if (ji < m) {
            const JuncEnds E = ends[ji];
            const int s = iabs(E.s), t = iabs(E.t);
            const bool same = (E.s < 0) == (E.t < 0);
            bool in = !(s < 1 || s > n || t < 1 || t > n);
            bool fbi = !same && iabs(s - t) <= 2;
            bool normal = same && ((E.s > 0 && t - s == 1) || (E.s < 0 && s - t == 1));
            q = (in && !fbi && !normal) ? 1 : 0;
        }
        int tot;
        int ex = g.exscan_i32(q, &tot);
        if (q) { S.sv[nsv + ex] = ji; S.taken[nsv + ex] = 0; }
        nsv += tot;
    }
    g.sync();
    if (nsv == 0) return 0;

    // -- which SVs could be chained onto by a LATER one (superset of what the deque chaining can do): SV j chains onto
    // (front, back) when  a_tgt(j) == front, b_tgt(j) = -a_src(j) == front, a_src(j) == back or b_src(j) = -a_tgt(j)
    // == back.  The occurrence tables (not yet in use) hold, per vertex, the LAST SV that has it as its a_src / a_tgt.
    for (int i = g.tid(); i < 2 * n + 1; i += g.size()) { S.first[i] = -1; S.last[i] = -1; }
    g.sync();
    for (int i = g.tid(); i < nsv; i += g.size()) {
        const JuncEnds E = ends[S.sv[i]];
        atomic_max_i32(&S.first[E.s + n], i);   // by a_src
        atomic_max_i32(&S.last[E.t + n], i);    // by a_tgt
    }
    g.sync();
    for (int i = g.tid(); i < nsv; i += g.size()) {
        const JuncEnds E = ends[S.sv[i]];
        const int front = E.s, back = E.t;
        const bool ext = S.last[front + n] > i || S.first[-front + n] > i || S.first[back + n] > i || S.last[-back + n] > i;
        S.has_ext[i] = (uint8_t)(ext ? 1 : 0);
    }
    g.sync();

    return nsv;
}

// LGM.cpp:3746-3837.  path/P are updated in place.  Returns 1 when the reference prints the
// "BFB path with insertion, deletion, or duplication:" caption (any qualifying SV exists), 0 otherwise,
// negative Status on capacity errors.
template <class G>
AMBI_HD int indel_bfb(const G& g, int n, const JuncEnds* ends, int m, cell_t* path, int* P_io, int pcap,
                      const IndelScratch& S, bool* edited = nullptr) {
    int P = *P_io;
    if (edited) *edited = false;
    const int nsv = indel_collect(g, n, ends, m, S);
    if (nsv == 0) return 0;

    auto rebuild_tables = [&]() {
        for (int i = g.tid(); i < 2 * n + 1; i += g.size()) { S.first[i] = 0x7fffffff; S.last[i] = -1; }
        g.sync();
        // eight cells per thread and round: their loads are issued together (the path may live in device memory, the direct
        // full-finish launch: one round trip per CELL made a rebuild ~15 k cycles, and every edit is followed by one)
        for (int base = 0; base < P; base += 8 * g.size()) {
            int v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) { const int i = base + k * g.size() + g.tid(); v[k] = i < P ? (int)path[i] : 0; }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int i = base + k * g.size() + g.tid();
                if (i < P) { atomic_min_i32(&S.first[v[k] + n], i); atomic_max_i32(&S.last[v[k] + n], i); }
            }
        }
        g.sync();
    };
    auto apply_single = [&](const SingleEdit& E) -> int {
        if (E.kind == 1) {
            int cnt = E.b - E.a;
            shift_down(g, path, P, E.a, cnt);
            P -= cnt;
        } else if (E.kind == 2) {
            int cnt = E.b - E.a;
            if (P + cnt > pcap) return ST_ERR_PATH_CAPACITY;
            shift_up(g, path, P, E.b, cnt);
            for (int k = g.tid(); k < cnt; k += g.size()) path[E.b + k] = path[E.a + k];
            P += cnt;
            g.sync();
        }
        return 0;
    };

    bool tables_ok = false;
    int f = 0;
    while (f < nsv) {
        if (!tables_ok) { rebuild_tables(); tables_ok = true; }
        // parallel sweep: first SV at or after f that is not a no-op
        int stop = 0x7fffffff;
        for (int i = f + g.tid(); i < nsv; i += g.size()) {
            if (S.taken[i]) continue;
            if (S.has_ext[i]) { stop = i; break; }
            const JuncEnds J = ends[S.sv[i]];
            SingleEdit E = eval_single(J.s, J.t, n, LdsPath{path}, P, S.first, S.last);
            if (E.kind != 0) { stop = i; break; }
        }
        stop = g.min_i32(stop);
        if (stop == 0x7fffffff) break;
        if (!S.has_ext[stop]) {
            const JuncEnds J = ends[S.sv[stop]];
            SingleEdit E = eval_single(J.s, J.t, n, LdsPath{path}, P, S.first, S.last);   // uniform re-evaluation
            g.sync();
            int rc = apply_single(E);
            if (rc < 0) { *P_io = P; return rc; }
            if (edited) *edited = true;
            tables_ok = false;
            f = stop + 1;
            continue;
        }
        // ---- general path: deque chaining from SV `stop` (LGM.cpp:3762-3776) ----
        const int first = stop;
        int head = m + 2, tail = m + 2;
        {
            const JuncEnds J = ends[S.sv[first]];
            g.sync();
            if (g.tid() == 0) { S.grp[tail] = J.s; S.grp[tail + 1] = J.t; S.taken[first] = 1; }
            tail += 2;
            g.sync();
        }
        int cursor = first + 1;
        while (true) {
            int front = S.grp[head], back = S.grp[tail - 1];
            int cand = 0x7fffffff;
            for (int i = cursor + g.tid(); i < nsv; i += g.size()) {
                if (S.taken[i]) continue;
                const JuncEnds J = ends[S.sv[i]];
                if (J.t == front || -J.s == front || back == J.s || back == -J.t) { cand = i; break; }
            }
            cand = g.min_i32(cand);
            if (cand == 0x7fffffff) break;
            const JuncEnds J = ends[S.sv[cand]];
            int nf = head, nt = tail;
            int wpos = -1, wval = 0;
            if (J.t == front) { nf = head - 1; wpos = nf; wval = J.s; }
            else if (-J.s == front) { nf = head - 1; wpos = nf; wval = -J.t; }
            else if (back == J.s) { wpos = tail; wval = J.t; nt = tail + 1; }
            else { wpos = tail; wval = -J.s; nt = tail + 1; }
            g.sync();
            if (g.tid() == 0) { S.grp[wpos] = wval; S.taken[cand] = 1; }
            head = nf; tail = nt;
            g.sync();
            cursor = cand + 1;
        }
        f = stop + 1;
        // -- apply the group (LGM.cpp:3779-3832).  All std::finds of a group precede its edit, so the occurrence
        // tables (valid here) answer them; a real scan is needed only for "next occurrence after pos1" when the vertex
        // occurs both before and after pos1.
        auto t_find = [&](int val) -> int { int f0 = S.first[val + n]; return f0 == 0x7fffffff ? P : f0; };
        auto t_find_before = [&](int pos1, int val) -> int { int f0 = S.first[val + n]; return (f0 != 0x7fffffff && f0 < pos1) ? f0 : pos1; };
        auto t_find_after = [&](int pos1, int val) -> int {
            if (pos1 >= P) return P;
            int f0 = S.first[val + n];
            if (f0 != 0x7fffffff && f0 > pos1) return f0;
            if (S.last[val + n] <= pos1) return P;
            return find_first(g, path, pos1 + 1, P, val);
        };
        tables_ok = false;   // cleared up front; restored below when the group turns out to be a no-op
        int gs = tail - head;
        auto complement_all = [&]() {
            g.sync();
            if (g.tid() == 0) {
                for (int a = head, b = tail - 1; a < b; a++, b--) { int t = S.grp[a]; S.grp[a] = S.grp[b]; S.grp[b] = t; }
                for (int a = head; a < tail; a++) S.grp[a] = -S.grp[a];
            }
            g.sync();
        };
        if (gs == 2) {
            int g0 = S.grp[head], g1 = S.grp[head + 1];
            if ((g0 > 0) == (g1 > 0)) {
                bool deletion = (g0 > 0 && iabs(g0) < iabs(g1)) || (g0 < 0 && iabs(g0) > iabs(g1));
                if (deletion) {
                    int pos1 = t_find(g0);
                    int pos2 = t_find_after(pos1, g1);
                    if (pos1 == P || pos2 == P) {
                        complement_all();
                        g0 = S.grp[head]; g1 = S.grp[head + 1];
                        pos1 = t_find(g0);
                        pos2 = t_find_after(pos1, g1);
                    }
                    if (pos1 == P || pos2 == P || pos2 - pos1 > 3) { tables_ok = true; continue; }
                    int cnt = pos2 - (pos1 + 1);
                    if (cnt > 0) { shift_down(g, path, P, pos1 + 1, cnt); P -= cnt; }
                } else {   // duplication
                    int pos1 = t_find(g0);
                    int pos2 = t_find_before(pos1, g1);
                    if (pos1 == P || pos2 == pos1) {
                        complement_all();
                        g0 = S.grp[head]; g1 = S.grp[head + 1];
                        pos1 = t_find(g0);
                        pos2 = t_find_before(pos1, g1);
                    }
                    if (pos1 == P || pos2 == pos1) { tables_ok = true; continue; }
                    int cnt = pos1 + 1 - pos2;
                    if (P + cnt > pcap) { *P_io = P; return ST_ERR_PATH_CAPACITY; }
                    shift_up(g, path, P, pos1 + 1, cnt);
                    for (int k = g.tid(); k < cnt; k += g.size()) path[pos1 + 1 + k] = path[pos2 + k];
                    P += cnt;
                    g.sync();
                }
            } else {   // inversion
                int pos1 = t_find(g0);
                int pos2 = t_find_after(pos1, g1);
                if (pos1 == P || pos2 == P) {
                    complement_all();
                    g0 = S.grp[head]; g1 = S.grp[head + 1];
                    pos1 = t_find(g0);
                    pos2 = t_find_after(pos1, g1);
                }
                if (pos1 == P || pos2 == P || pos2 - pos1 > 5) { tables_ok = true; continue; }
                int cnt = pos2 - (pos1 + 1);
                if (cnt > 0) { shift_down(g, path, P, pos1 + 1, cnt); P -= cnt; }
            }
        } else {   // insertion
            int gf = S.grp[head], gb = S.grp[tail - 1];
            int pos1 = t_find(gf);
            int pos2 = t_find_after(pos1, gb);
            if (pos1 == P || pos2 == P) {
                complement_all();
                gf = S.grp[head]; gb = S.grp[tail - 1];
                pos1 = t_find(gf);
                pos2 = t_find_after(pos1, gb);
            }
            if (pos1 == P || pos2 == P) { tables_ok = true; continue; }
            int cnt = pos2 - (pos1 + 1);
            if (cnt > 0) { shift_down(g, path, P, pos1 + 1, cnt); P -= cnt; }
            int ins = gs - 2;
            if (P + ins > pcap) { *P_io = P; return ST_ERR_PATH_CAPACITY; }
            shift_up(g, path, P, pos1 + 1, ins);
            for (int k = g.tid(); k < ins; k += g.size()) path[pos1 + 1 + k] = (cell_t)S.grp[head + 1 + k];
            P += ins;
            g.sync();
        }
        if (edited) *edited = true;   // reached only when the group was applied (the no-op exits `continue` above)
    }
    *P_io = P;
    return 1;
}

// A junction step (u,v) of the path, both vertices in one word.
AMBI_HD int32_t pack_step(int u, int v) { return (int32_t)(((uint32_t)(u + 32768) << 16) | (uint32_t)(v + 32768)); }
AMBI_HD int step_u(int32_t w) { return (int)((uint32_t)w >> 16) - 32768; }
AMBI_HD int step_v(int32_t w) { return (int)((uint32_t)w & 0xFFFFu) - 32768; }

// Tail of the output-junction synthesis (localhap.cpp:275-289) on the nc junction steps in path order, given as packed
// (u,v) words in cand[0,nc); cand must hold 3*nc ints.  A step joins the FIRST earlier step that is the same edge or
// its complement edge (the reference bumps that entry's count); a record can never coexist with its complement, so
// classes are disjoint.  Returns the number of output junctions or a negative Status.
// htab / hsize: optional hash table (2 * hsize ints of group memory, hsize a power of two >= 2 * nc): the first step of every edge is
// then found through it -- one insertion (compare-and-swap on the key, minimum on the index) and one look-up per step -- instead of a
// scan of all earlier steps per step (nc^2 / 2 key reads: 15 k of the lean stage's 36 k cycles for the bench unit's ~100 steps).
AMBI_HD int synth_hash_size(int nc, int ints_available) {   // 0: no room, the scan
    int h = 64;
    while (h < 2 * nc) h <<= 1;
    return 2 * h <= ints_available ? h : 0;
}
template <class G>
AMBI_HD int synth_classes(const G& g, int32_t* cand, int nc, OutJunc* out, int cap, int seg_base, int32_t* htab = nullptr, int hsize = 0) {
    int32_t* rep = cand + nc;      // [nc] representative (first occurrence) of every step
    int32_t* key = cand + 2 * nc;  // [nc] canonical edge key: the smaller of (u,v) and its complement (-v,-u), packed
    if (htab != nullptr && hsize >= 2 * nc && (hsize & (hsize - 1)) == 0) {
        int32_t* hk = htab;            // [hsize] key of the slot, 0 = empty (a packed step is never 0: ids are non-zero)
        int32_t* hi = htab + hsize;    // [hsize] first step with that key
        for (int i = g.tid(); i < hsize; i += g.size()) { hk[i] = 0; hi[i] = 0x7fffffff; }
        g.sync();
        for (int c = g.tid(); c < nc; c += g.size()) {
            const int u = step_u(cand[c]), v = step_v(cand[c]);
            const int32_t k1 = pack_step(u, v), k2 = pack_step(-v, -u);
            const int32_t k = k1 < k2 ? k1 : k2;
            uint32_t h = ((uint32_t)k * 0x9E3779B1u) >> 7 & (uint32_t)(hsize - 1);
            while (true) {   // (the table is at most half full: an empty slot ends every probe sequence)
                const int old = atomic_cas_i32(&hk[h], 0, k);
                if (old == 0 || old == k) break;
                h = (h + 1) & (uint32_t)(hsize - 1);
            }
            atomic_min_i32(&hi[h], c);
            rep[c] = (int32_t)h;
        }
        g.sync();
        for (int c = g.tid(); c < nc; c += g.size()) rep[c] = hi[rep[c]];
        g.sync();
    } else {
    for (int c = g.tid(); c < nc; c += g.size()) {
        const int u = step_u(cand[c]), v = step_v(cand[c]);
        const int32_t k1 = pack_step(u, v), k2 = pack_step(-v, -u);
        key[c] = k1 < k2 ? k1 : k2;
    }
    g.sync();
    for (int c = g.tid(); c < nc; c += g.size()) {
        const int32_t mine_key = key[c];
        int r = c;
        for (int k0 = 0; k0 < c; k0 += 8) {   // eight independent loads per round trip instead of one
            int hit = c;
#pragma unroll
            for (int q = 7; q >= 0; q--) { const int k = k0 + q; if (k < c && key[k] == mine_key) hit = k; }
            if (hit < c) { r = hit; break; }
        }
        rep[c] = r;
    }
    g.sync();
    }
    // class sizes in group memory (the key array is free now): every step bumps its representative's counter
    for (int c = g.tid(); c < nc; c += g.size()) key[c] = 0;
    g.sync();
    for (int c = g.tid(); c < nc; c += g.size()) atomic_add_i32(&key[rep[c]], 1);
    g.sync();
    // leaders in first-appearance order -> finished records, one store each
    int nout = 0;
    for (int base = 0; base < nc; base += g.size()) {
        const int c = base + g.tid();
        const int lead = (c < nc && rep[c] == c) ? 1 : 0;
        int tot;
        const int ex = g.exscan_i32(lead, &tot);
        if (lead && nout + ex < cap) {
            OutJunc& o = out[nout + ex];
            const int pu = step_u(cand[c]), pv = step_v(cand[c]);   // local signed ids -> absolute
            o.u = pu > 0 ? pu + seg_base : pu - seg_base; o.v = pv > 0 ? pv + seg_base : pv - seg_base; o.count = key[c];
        }
        nout += tot;
    }
    if (nout > cap) return ST_ERR_OUTJUNC_CAPACITY;
    return nout;
}

// localhap.cpp:267-289: consecutive path vertices that are not |id difference| == 1 on one strand become
// output junctions; a repeat (same edge or its complement edge) bumps the count.  First-appearance order.
// cand = scratch of cand_cap ints.  Returns the number of output junctions or a negative Status.
template <class G>
AMBI_HD int synth_out_juncs(const G& g, const cell_t* path, int P, OutJunc* out, int cap, int32_t* cand, int cand_cap, int seg_base) {
    if (P <= 0) return 0;
    // every sub-group (wavefront) owns a contiguous slice of the path and walks it 64 cells at a time (conflict-free
    // LDS reads, flags ranked by ballot without a barrier); ONE group scan over the sub-group totals orders all steps
    // Vertices are non-zero, so "same strand and ids one apart" is simply |v - u| == 1.  ONE pass over the path, eight
    // cells per thread and step (one 16-byte read of group memory): the rare junction steps are appended to a list
    // through a counter, then put in path order by rank (the list is short: a few per breakpoint pair).
    const int steps = P - 1;
    int32_t* count = cand + cand_cap - 1;                 // the list counter lives in the last scratch word
    int32_t* unsorted = cand + 2 * ((cand_cap - 1) / 3);  // [<= (cand_cap-1)/3] appended positions
    const int list_cap = (cand_cap - 1) / 3;
    if (g.tid() == 0) *count = 0;
    g.sync();
    for (int i0 = 8 * g.tid(); i0 < steps; i0 += 8 * g.size()) {
        cell_t c[9];
        // the path is 16-byte aligned and padded by 16 bytes, so the four words and the ninth cell are always readable
        const uint32_t* w4 = reinterpret_cast<const uint32_t*>(__builtin_assume_aligned(path, 16)) + (i0 >> 1);
        const uint32_t qx = w4[0], qy = w4[1], qz = w4[2], qw = w4[3];
        c[0] = (cell_t)(qx & 0xFFFF); c[1] = (cell_t)(qx >> 16); c[2] = (cell_t)(qy & 0xFFFF); c[3] = (cell_t)(qy >> 16);
        c[4] = (cell_t)(qz & 0xFFFF); c[5] = (cell_t)(qz >> 16); c[6] = (cell_t)(qw & 0xFFFF); c[7] = (cell_t)(qw >> 16);
        c[8] = path[i0 + 8];
        uint32_t flags = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int d = (int)c[k + 1] - (int)c[k];
            flags |= (uint32_t)((d != 1 && d != -1) && (i0 + k < steps)) << k;
        }
        while (flags) {
            const int k = __builtin_ctz(flags);
            flags &= flags - 1;
            const int at = atomic_add_i32(count, 1);
            if (at < list_cap) unsorted[at] = i0 + k;
        }
    }
    g.sync();
    const int nc = *count;
    if (nc > list_cap || 3 * nc > cand_cap - 1) return ST_ERR_OUTJUNC_CAPACITY;
    for (int c = g.tid(); c < nc; c += g.size()) {
        const int pos = unsorted[c];
        int r = 0;
        for (int k = 0; k < nc; k++) r += unsorted[k] < pos ? 1 : 0;
        cand[r] = pos;   // cand [0, nc) never overlaps the unsorted list: nc <= list_cap and the list starts at 2*list_cap
    }
    g.sync();
    // positions -> the step's two vertices, packed (the tail below no longer needs the path)
    for (int c = g.tid(); c < nc; c += g.size()) {
        const int pos = cand[c];
        cand[c] = pack_step(path[pos], path[pos + 1]);
    }
    g.sync();
#if defined(__HIP_DEVICE_COMPILE__)
    if (G::kIsBlock && nc <= 256) {   // few steps: one wavefront goes through the phases of the tail without workgroup barriers (as the lean stage does)
        int v = 0;
        if (g.tid() < 64) { WaveGroup w; v = synth_classes(w, cand, nc, out, cap, seg_base); }
        return g.bcast_i32(v, 0);
    }
#endif
    return synth_classes(g, cand, nc, out, cap, seg_base);
}

// ---------------------------------------------------------------------------------------------------------------
// Lean finish (runs only, no path cells in group memory).  The common case -- indelBFB looks its SVs up and edits
// nothing -- needs the path only (a) as output, (b) for the occurrence tables and a handful of cell reads, (c) for the
// junction steps, and all three follow from the runs: (a) expand_runs streams the cells to HBM, (b) the tables are
// filled in the same sweep and single cells come from RunPath::at, (c) inside a run consecutive cells differ by one,
// so junction steps exist only between the last cell of a run and the first cell of the next non-empty one.
// A unit whose SVs chain or edit the path is handed to the full stage instead (rare).
// ---------------------------------------------------------------------------------------------------------------

// run offsets: offs[j] = first position of pair j, offs[np] = P.  Returns P.
template <class G>
AMBI_HD int run_offsets(const G& g, const cell_t* bkp, int L, int32_t* offs) {
    const int np = L / 2;
    int carry = 0;
    for (int base = 0; base < np; base += g.size()) {
        const int j = base + g.tid();
        int len = 0;
        if (j < np) {
            const int a = bkp[2 * j], b = bkp[2 * j + 1];
            len = a > 0 ? iabs(b) - a + 1 : (-a) - iabs(b) + 1;   // LGM.cpp:3661-3670
            if (len < 0) len = 0;
        }
        int tot;
        const int ex = g.exscan_i32(len, &tot);
        if (j < np) offs[j] = carry + ex;
        carry += tot;
    }
    if (g.tid() == 0) offs[np] = carry;
    g.sync();
    return carry;
}

// cells of every run to the result blob (absolute signed ids); with `first` != nullptr also the occurrence tables
// (initialised by the caller) -- run-major, one sub-group (wavefront) per run as in expand_bkp
template <class G>
AMBI_HD void expand_runs(const G& g, const cell_t* bkp, int np, const int32_t* offs, int16_t* gpath, int seg_base, int n,
                         int32_t* first, int32_t* last, int16_t* mirror = nullptr) {
    const int lanes = g.size() < 64 ? g.size() : 64;
    const int sub = g.tid() / lanes, nsub = g.size() / lanes, lane = g.tid() - sub * lanes;
    for (int j = sub; j < np; j += nsub) {
        const int a = bkp[2 * j], o0 = offs[j], len = offs[j + 1] - o0;
        for (int k = lane; k < len; k += lanes) {
            const int v = a + k;
#if !defined(AMBI_LEAN_SKIP) || !(AMBI_LEAN_SKIP & 2)
            gpath[o0 + k] = (int16_t)v;   // local id; readers add the base
#endif
            if (mirror) mirror[o0 + k] = (int16_t)v;   // the caller's copy in pinned host memory (express path), in the same pass
#if !defined(AMBI_LEAN_SKIP) || !(AMBI_LEAN_SKIP & 1)
            if (first) { atomic_min_i32(&first[v + n], o0 + k); atomic_max_i32(&last[v + n], o0 + k); }
#endif
        }
    }
    g.sync();
}

// indelBFB when nothing chains and nothing edits: 1 = the SVs are all look-ups without effect (the reference prints its
// caption and the unchanged path), 0 = the full stage has to take this unit.  nsv > 0, tables filled.
// (the loop of one thread: SVs tid, tid + size, ..; 1 = one of them chains or edits)
AMBI_HD int indel_lookups_thread(int tid, int size, int n, const JuncEnds* ends, int nsv, const RunPath& path, int P, const IndelScratch& S) {
    int stop = 0;
    for (int i = tid; i < nsv && !stop; i += size) {
        if (S.has_ext[i]) { stop = 1; break; }
        const JuncEnds J = ends[S.sv[i]];
        if (eval_single(J.s, J.t, n, path, P, S.first, S.last).kind != 0) stop = 1;
    }
    return stop;
}
template <class G>
AMBI_HD int indel_lookups_only(const G& g, int n, const JuncEnds* ends, int nsv, const RunPath& path, int P, const IndelScratch& S) {
    const int stop = indel_lookups_thread(g.tid(), g.size(), n, ends, nsv, path, P, S);
    return g.any(stop != 0) ? 0 : 1;
}

// output junctions from the runs; cand: 3 * (np + 1) ints
template <class G>
AMBI_HD int synth_out_juncs_runs(const G& g, const cell_t* bkp, int np, const int32_t* offs, OutJunc* out, int cap, int32_t* cand,
                                 int seg_base, int32_t* htab = nullptr, int htab_ints = 0) {
    int nc = 0;
    for (int base = 0; base < np; base += g.size()) {
        const int j = base + g.tid();
        int flag = 0, u = 0, v = 0;
        if (j < np && offs[j + 1] > offs[j]) {
            int nx = j + 1;
            while (nx < np && offs[nx + 1] == offs[nx]) nx++;      // next non-empty run
            if (nx < np) {
                u = bkp[2 * j] + (offs[j + 1] - offs[j] - 1);
                v = bkp[2 * nx];
                const int d = v - u;
                flag = (d != 1 && d != -1) ? 1 : 0;
            }
        }
        int tot;
        const int ex = g.exscan_i32(flag, &tot);
        if (flag) cand[nc + ex] = pack_step(u, v);
        nc += tot;
    }
    g.sync();
    return synth_classes(g, cand, nc, out, cap, seg_base, htab, htab ? synth_hash_size(nc, htab_ints) : 0);
}

// ---------------------------------------------------------------------------------------------------------------
// The final path in run-length form (BatchArgs::run_*): a run starts where a cell is not its predecessor + 1.  rs / rl: the
// unit's slots (cap of them).  Returns the number of runs, or -(number of runs) when they do not fit.  While the starts are
// being collected rl[r] holds the run's first POSITION; runs_fix_lengths turns positions into lengths.
// ---------------------------------------------------------------------------------------------------------------
AMBI_HD int32_t run_abs_cell(int v, int seg_base) { return v > 0 ? v + seg_base : v - seg_base; }   // (= abs_cell of ambi_batch.hpp, which includes this header)
template <class G>
AMBI_HD void runs_fix_lengths(const G& g, int32_t* rl, int nr, int P) {
    for (int base = 0; base < nr; base += g.size()) {
        const int r = base + g.tid();
        int len = 0;
        if (r < nr) len = (r + 1 < nr ? rl[r + 1] : P) - rl[r];   // (the next run's position is read before anyone overwrites it)
        g.sync();
        if (r < nr) rl[r] = len;
        g.sync();
    }
}
// from the breakpoint pairs (lean stage): pair j covers positions [offs[j], offs[j+1]) with the values bkp[2j] + k.  A pair starts a run
// when it is not empty and does not continue the previous non-empty pair; the run's length follows from the next such pair, which the
// starting thread looks for itself (a run rarely spans more than two or three pairs) -- nothing is read back from the slots, which lie
// in device memory, and no pass over them follows.
template <class G>
AMBI_HD int emit_runs_pairs(const G& g, const cell_t* bkp, int np, const int32_t* offs, int P, int seg_base, int32_t* rs, int32_t* rl, int cap) {
    auto starts = [&](int j) -> bool {   // pair j (not empty) opens a run
        int pj = j - 1;
        while (pj >= 0 && offs[pj + 1] == offs[pj]) pj--;      // previous non-empty pair
        return pj < 0 || (int)bkp[2 * pj] + (offs[pj + 1] - offs[pj]) != (int)bkp[2 * j];
    };
    int nr = 0;
    for (int base = 0; base < np; base += g.size()) {
        const int j = base + g.tid();
        int start = 0, a = 0, len = 0;
        if (j < np && offs[j + 1] > offs[j]) {
            a = bkp[2 * j];
            start = starts(j) ? 1 : 0;
            if (start) {
                int k = j + 1;
                while (k < np && (offs[k + 1] == offs[k] || !starts(k))) k++;
                len = (k < np ? offs[k] : P) - offs[j];
            }
        }
        int tot;
        const int ex = g.exscan_i32(start, &tot);
        if (start && nr + ex < cap) { rs[nr + ex] = run_abs_cell(a, seg_base); rl[nr + ex] = len; }
        nr += tot;
    }
    g.sync();
    return nr > cap ? -nr : nr;
}
// from the cells (full stage; the cells in group or device memory): every thread takes a contiguous stretch
template <class G>
AMBI_HD int emit_runs_cells(const G& g, const cell_t* path, int P, int seg_base, int32_t* rs, int32_t* rl, int cap) {
    const int per = (P + g.size() - 1) / g.size();
    int lo = g.tid() * per, hi = lo + per;
    if (lo > P) lo = P;
    if (hi > P) hi = P;
    int mine = 0;
    for (int i = lo; i < hi; i++) mine += (i == 0 || (int)path[i] != (int)path[i - 1] + 1) ? 1 : 0;
    int nr;
    int at = g.exscan_i32(mine, &nr);
    if (nr <= cap)
        for (int i = lo; i < hi; i++)
            if (i == 0 || (int)path[i] != (int)path[i - 1] + 1) { rs[at] = run_abs_cell(path[i], seg_base); rl[at] = i; at++; }
    g.sync();
    if (nr > cap) return -nr;
    runs_fix_lengths(g, rl, nr, P);
    return nr;
}

// ---------------------------------------------------------------------------------------------------------------
// indelBFB on the RUNS of the path (stage_finish_edit).  indel_bfb above moves the cells of the path for every edit and fills the
// occurrence tables again behind it -- two sweeps over ~14 000 cells per applied group for the bench unit, with the cells in device
// memory when the workgroup is to fit beside the order-table kernel.  Here the path stays what the lean stage works on, a list of
// runs (run r = the values val[2r] + k at the positions [off[r], off[r+1])), and nothing is ever moved:
//   * erase [a, b)                  = the runs of [0, a) followed by the runs of [b, P)
//   * duplicate [a, b) at b         = the runs of [0, b), of [a, b) again and of [b, P)
//   * insert cells c.. at q         = the runs of [0, q), one run per cell, the runs of [q, P)
//     -- a slice of the list is a contiguous range of runs with the first and the last one cut, so a new list is written by one
//     thread per run into the other of two buffers;
//   * the occurrence tables come from the runs, one thread per vertex: a run holds a value at most once, at a position that follows
//     from its first value (2n+1 vertices x ~100-250 runs of broadcast reads instead of two atomics per cell);
//   * "next occurrence behind pos1" is a minimum over the runs.
// Decisions, their order and the deque of chaining SVs are indel_bfb's, statement by statement.  The cells of the edited path are
// written once, at the end, from the final list (stage_finish_edit).
// ---------------------------------------------------------------------------------------------------------------
struct RunList {
    cell_t* val;     // [2 * cap]  first value of run r at val[2r] (the layout of the breakpoint pairs; odd entries unused)
    int32_t* off;    // [cap + 1]  first position of every run, off[n] = P
    int n, P;
};
constexpr int kRunsNoRoom = -1000;   // indel_bfb_runs: the run list outgrew its room (the caller hands the unit to the full stage)

// D = up to three slices [x[k], y[k]) of S in this order, with `nlit` one-cell runs (values lit[0 .. nlit)) behind the first
// `lit_at` slices.  Returns the number of runs of D, -1 when they do not fit `cap`.  loc: 8 ints of group memory.  D.n / D.P are
// set; ends with a barrier.  Empty slices are allowed.
template <class G>
AMBI_HD int runs_build(const G& g, const RunList& S, RunList& D, int cap, int32_t* loc, int ns, const int* x, const int* y,
                       int lit_at = 0, const int32_t* lit = nullptr, int nlit = 0) {
    // the (non-empty) runs that hold the first and the last position of every slice
    for (int r = g.tid(); r < S.n; r += g.size()) {
        const int o0 = S.off[r], o1 = S.off[r + 1];
        for (int k = 0; k < ns; k++) {
            if (y[k] <= x[k]) continue;
            if (o0 <= x[k] && x[k] < o1) loc[2 * k] = r;
            if (o0 <= y[k] - 1 && y[k] - 1 < o1) loc[2 * k + 1] = r;
        }
    }
    g.sync();
    int total = 0, newP = 0, r0[3] = {0, 0, 0}, cnt[3] = {0, 0, 0}, dn[3] = {0, 0, 0}, db[3] = {0, 0, 0}, ln = 0, lb = 0;
    for (int k = 0; k < ns; k++) {
        if (k == lit_at) { ln = total; lb = newP; total += nlit; newP += nlit; }
        if (y[k] > x[k]) { r0[k] = loc[2 * k]; cnt[k] = loc[2 * k + 1] - r0[k] + 1; }
        dn[k] = total; db[k] = newP;
        total += cnt[k]; newP += y[k] > x[k] ? y[k] - x[k] : 0;
    }
    if (lit_at >= ns) { ln = total; lb = newP; total += nlit; newP += nlit; }
    D.n = total; D.P = newP;
    if (total > cap) { g.sync(); return -1; }
    for (int k = 0; k < ns; k++) {
        for (int i = g.tid(); i < cnt[k]; i += g.size()) {
            const int r = r0[k] + i;
            const int o0 = S.off[r];
            const int lo = o0 > x[k] ? o0 : x[k];           // (a run in the middle of the slice may be empty: it stays an empty run)
            D.val[2 * (dn[k] + i)] = (cell_t)((int)S.val[2 * r] + (lo - o0));
            D.off[dn[k] + i] = db[k] + (lo - x[k]);
        }
    }
    for (int i = g.tid(); i < nlit; i += g.size()) { D.val[2 * (ln + i)] = (cell_t)lit[i]; D.off[ln + i] = lb + i; }
    if (g.tid() == 0) D.off[total] = newP;
    g.sync();
    return total;
}

// occurrence tables of the path from its runs: one thread per vertex (up to kTabV vertices per thread and pass over the runs), the
// runs in position order, eight runs per round -- their words are read together (one latency per round, not per run), and a vertex
// keeps the first and the last RUN that holds it (branch-free), its positions follow at the end.  Ends with a barrier.
constexpr int kTabV = 4;
template <int NK, class G>
AMBI_HD void tables_from_runs_pass(const G& g, int n, const RunList& R, int32_t* first, int32_t* last, int vb) {
    // vertex number j in [0, 2n) stands for the vertex j - n (j < n) or j - n + 1: there is no vertex 0
    int v[NK], rf[NK], rl[NK];
#pragma unroll
    for (int k = 0; k < NK; k++) { const int j = vb + k * g.size() + g.tid(); v[k] = j < n ? j - n : j - n + 1; rf[k] = 0x7fffffff; rl[k] = -1; }
    for (int r0 = 0; r0 < R.n; r0 += 8) {
        int o[9], a[8];
#pragma unroll
        for (int j = 0; j < 9; j++) { const int r = r0 + j < R.n ? r0 + j : R.n; o[j] = R.off[r]; }       // (runs behind the list: length 0)
#pragma unroll
        for (int j = 0; j < 8; j++) { const int r = r0 + j < R.n ? r0 + j : R.n - 1; a[j] = R.val[2 * r]; }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const unsigned len = (unsigned)(o[j + 1] - o[j]);
#pragma unroll
            for (int k = 0; k < NK; k++) {
                const bool hit = (unsigned)(v[k] - a[j]) < len;
                const int rr = r0 + j;
                rl[k] = hit ? rr : rl[k];
                rf[k] = hit && rr < rf[k] ? rr : rf[k];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NK; k++) {
        if (vb + k * g.size() + g.tid() >= 2 * n) continue;
        const int i = v[k] + n;
        first[i] = rl[k] < 0 ? 0x7fffffff : R.off[rf[k]] + (v[k] - (int)R.val[2 * rf[k]]);
        last[i] = rl[k] < 0 ? -1 : R.off[rl[k]] + (v[k] - (int)R.val[2 * rl[k]]);
    }
}
template <class G>
AMBI_HD void tables_from_runs(const G& g, int n, const RunList& R, int32_t* first, int32_t* last) {
    const int nv = 2 * n;
    if (g.tid() == 0) { first[n] = 0x7fffffff; last[n] = -1; }
    for (int vb = 0; vb < nv; vb += kTabV * g.size()) {
        const int nk = (nv - vb + g.size() - 1) / g.size();
        if (nk >= 4) tables_from_runs_pass<4>(g, n, R, first, last, vb);
        else if (nk == 3) tables_from_runs_pass<3>(g, n, R, first, last, vb);
        else if (nk == 2) tables_from_runs_pass<2>(g, n, R, first, last, vb);
        else tables_from_runs_pass<1>(g, n, R, first, last, vb);
    }
    g.sync();
}

// first position in [from, P) that holds val, P if none (find_first on the runs)
template <class G>
AMBI_HD int runs_find_first(const G& g, const RunList& R, int from, int val) {
    int best = 0x7fffffff;
    for (int r = g.tid(); r < R.n; r += g.size()) {
        const int o0 = R.off[r], d = val - (int)R.val[2 * r];
        if (d >= 0 && d < R.off[r + 1] - o0 && o0 + d >= from && o0 + d < best) best = o0 + d;
    }
    best = g.min_i32(best);
    return best == 0x7fffffff ? R.P : best;
}

// indel_bfb on the run list `cur` (buf: the two other lists an edit alternates between, `cap` runs each; loc: 8 ints of group
// memory).  Same return values as indel_bfb, or kRunsNoRoom.  *edited as there.  The occurrence tables need not be filled.
#if defined(AMBI_EDIT_TRACE_ON) && !defined(__HIP_DEVICE_COMPILE__)
static long g_edit_count[4];
#define EDIT_COUNT(x) (g_edit_count[x]++)
#else
#define EDIT_COUNT(x) ((void)0)
#endif
#if defined(AMBI_EDIT_MARKS)   // diagnostic build: shader-clock marks of the LAST pass of the loop in the slots of the prepare stage's marks
#define EDIT_MARK(x) clk_mark(g, clk, x)
#else
#define EDIT_MARK(x) ((void)0)
#endif
template <class G>
AMBI_HD int indel_bfb_runs(const G& g, int n, const JuncEnds* ends, int m, RunList& cur, RunList* buf, int cap, int pcap,
                           const IndelScratch& S, int32_t* loc, bool* edited, int64_t* clk = nullptr) {
    *edited = false;
    const int nsv = indel_collect(g, n, ends, m, S);
    if (nsv == 0) return 0;
    int which = 0;
    // the edit: slices of cur (and literal cells) -> the other buffer, which becomes cur
    auto rebuild = [&](int ns, const int* x, const int* y, int lit_at, const int32_t* lit, int nlit) -> bool {
        RunList nxt{buf[which].val, buf[which].off, 0, 0};
        if (runs_build(g, cur, nxt, cap, loc, ns, x, y, lit_at, lit, nlit) < 0) return false;
        cur = nxt;
        which ^= 1;
        return true;
    };
    auto erase = [&](int a, int b) -> bool { const int x[2] = {0, b}, y[2] = {a, cur.P}; return rebuild(2, x, y, 0, nullptr, 0); };
    auto duplicate = [&](int a, int b) -> bool { const int x[3] = {0, a, b}, y[3] = {b, b, cur.P}; return rebuild(3, x, y, 0, nullptr, 0); };

    bool tables_ok = false;
    int f = 0;
    while (f < nsv) {
        EDIT_MARK(1);
        if (!tables_ok) { tables_from_runs(g, n, cur, S.first, S.last); tables_ok = true; EDIT_COUNT(0); }
        EDIT_COUNT(1);
        EDIT_MARK(2);
        const RunPath RP{cur.val, cur.off, cur.n};
        const int P = cur.P;
        // parallel sweep: first SV at or after f that is not a no-op
        int stop = 0x7fffffff;
        for (int i = f + g.tid(); i < nsv; i += g.size()) {
            if (S.taken[i]) continue;
            if (S.has_ext[i]) { stop = i; break; }
            const JuncEnds J = ends[S.sv[i]];
            if (eval_single(J.s, J.t, n, RP, P, S.first, S.last).kind != 0) { stop = i; break; }
        }
        stop = g.min_i32(stop);
        EDIT_MARK(3);
        if (stop == 0x7fffffff) break;
        if (!S.has_ext[stop]) {
            const JuncEnds J = ends[S.sv[stop]];
            const SingleEdit E = eval_single(J.s, J.t, n, RP, P, S.first, S.last);   // uniform re-evaluation
            g.sync();
            if (E.kind == 2 && P + (E.b - E.a) > pcap) return ST_ERR_PATH_CAPACITY;
            if (!(E.kind == 1 ? erase(E.a, E.b) : duplicate(E.a, E.b))) return kRunsNoRoom;
            *edited = true;
            tables_ok = false;
            f = stop + 1;
            continue;
        }
        // ---- general path: deque chaining from SV `stop` (LGM.cpp:3762-3776) ----
        const int first = stop;
        int head = m + 2, tail = m + 2;
        {
            const JuncEnds J = ends[S.sv[first]];
            g.sync();
            if (g.tid() == 0) { S.grp[tail] = J.s; S.grp[tail + 1] = J.t; S.taken[first] = 1; }
            tail += 2;
            g.sync();
        }
        int cursor = first + 1;
        while (true) {
            int front = S.grp[head], back = S.grp[tail - 1];
            int cand = 0x7fffffff;
            for (int i = cursor + g.tid(); i < nsv; i += g.size()) {
                if (S.taken[i]) continue;
                const JuncEnds J = ends[S.sv[i]];
                if (J.t == front || -J.s == front || back == J.s || back == -J.t) { cand = i; break; }
            }
            EDIT_COUNT(2);
            cand = g.min_i32(cand);
            if (cand == 0x7fffffff) break;
            const JuncEnds J = ends[S.sv[cand]];
            int nf = head, nt = tail;
            int wpos = -1, wval = 0;
            if (J.t == front) { nf = head - 1; wpos = nf; wval = J.s; }
            else if (-J.s == front) { nf = head - 1; wpos = nf; wval = -J.t; }
            else if (back == J.s) { wpos = tail; wval = J.t; nt = tail + 1; }
            else { wpos = tail; wval = -J.s; nt = tail + 1; }
            g.sync();
            if (g.tid() == 0) { S.grp[wpos] = wval; S.taken[cand] = 1; }
            head = nf; tail = nt;
            g.sync();
            cursor = cand + 1;
        }
        f = stop + 1;
        EDIT_MARK(4);
        // -- apply the group (LGM.cpp:3779-3832), the std::finds answered from the occurrence tables as in indel_bfb
        auto t_find = [&](int val) -> int { int f0 = S.first[val + n]; return f0 == 0x7fffffff ? P : f0; };
        auto t_find_before = [&](int pos1, int val) -> int { int f0 = S.first[val + n]; return (f0 != 0x7fffffff && f0 < pos1) ? f0 : pos1; };
        auto t_find_after = [&](int pos1, int val) -> int {
            if (pos1 >= P) return P;
            int f0 = S.first[val + n];
            if (f0 != 0x7fffffff && f0 > pos1) return f0;
            if (S.last[val + n] <= pos1) return P;
            return runs_find_first(g, cur, pos1 + 1, val);
        };
        tables_ok = false;   // cleared up front; restored below when the group turns out to be a no-op
        int gs = tail - head;
        auto complement_all = [&]() {
            g.sync();
            if (g.tid() == 0) {
                for (int a = head, b = tail - 1; a < b; a++, b--) { int t = S.grp[a]; S.grp[a] = S.grp[b]; S.grp[b] = t; }
                for (int a = head; a < tail; a++) S.grp[a] = -S.grp[a];
            }
            g.sync();
        };
        if (gs == 2) {
            int g0 = S.grp[head], g1 = S.grp[head + 1];
            if ((g0 > 0) == (g1 > 0)) {
                bool deletion = (g0 > 0 && iabs(g0) < iabs(g1)) || (g0 < 0 && iabs(g0) > iabs(g1));
                if (deletion) {
                    int pos1 = t_find(g0);
                    int pos2 = t_find_after(pos1, g1);
                    if (pos1 == P || pos2 == P) {
                        complement_all();
                        g0 = S.grp[head]; g1 = S.grp[head + 1];
                        pos1 = t_find(g0);
                        pos2 = t_find_after(pos1, g1);
                    }
                    if (pos1 == P || pos2 == P || pos2 - pos1 > 3) { tables_ok = true; continue; }
                    if (pos2 - (pos1 + 1) > 0) { if (!erase(pos1 + 1, pos2)) return kRunsNoRoom; } else tables_ok = true;
                } else {   // duplication
                    int pos1 = t_find(g0);
                    int pos2 = t_find_before(pos1, g1);
                    if (pos1 == P || pos2 == pos1) {
                        complement_all();
                        g0 = S.grp[head]; g1 = S.grp[head + 1];
                        pos1 = t_find(g0);
                        pos2 = t_find_before(pos1, g1);
                    }
                    if (pos1 == P || pos2 == pos1) { tables_ok = true; continue; }
                    if (P + (pos1 + 1 - pos2) > pcap) return ST_ERR_PATH_CAPACITY;
                    if (!duplicate(pos2, pos1 + 1)) return kRunsNoRoom;
                }
            } else {   // inversion
                int pos1 = t_find(g0);
                int pos2 = t_find_after(pos1, g1);
                if (pos1 == P || pos2 == P) {
                    complement_all();
                    g0 = S.grp[head]; g1 = S.grp[head + 1];
                    pos1 = t_find(g0);
                    pos2 = t_find_after(pos1, g1);
                }
                if (pos1 == P || pos2 == P || pos2 - pos1 > 5) { tables_ok = true; continue; }
                if (pos2 - (pos1 + 1) > 0) { if (!erase(pos1 + 1, pos2)) return kRunsNoRoom; } else tables_ok = true;
            }
        } else {   // insertion
            int gf = S.grp[head], gb = S.grp[tail - 1];
            int pos1 = t_find(gf);
            int pos2 = t_find_after(pos1, gb);
            if (pos1 == P || pos2 == P) {
                complement_all();
                gf = S.grp[head]; gb = S.grp[tail - 1];
                pos1 = t_find(gf);
                pos2 = t_find_after(pos1, gb);
            }
            if (pos1 == P || pos2 == P) { tables_ok = true; continue; }
            const int cnt = pos2 - (pos1 + 1) > 0 ? pos2 - (pos1 + 1) : 0;
            const int ins = gs - 2;
            if (P - cnt + ins > pcap) return ST_ERR_PATH_CAPACITY;
            EDIT_MARK(5);
            const int x[2] = {0, pos2}, y[2] = {pos1 + 1, P};
            if (!rebuild(2, x, y, 1, S.grp + head + 1, ins)) return kRunsNoRoom;
        }
        EDIT_MARK(6);
        *edited = true;   // reached only when the group was applied (the no-op exits `continue` above)
    }
    return 1;
}

}  // namespace ambi
