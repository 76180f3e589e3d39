// ambi_finish.hpp -- post-assembly edits of the per-segment path and output-junction synthesis.
//
// Restates LocalGenomicMap::indelBFB (LGM.cpp:3746-3837) and the output junction loop of main()
// (localhap.cpp:267-289) as SPMD code over a workgroup: the path (int16 signed local ids) sits in LDS, every
// std::find is a strided scan + min-reduction, erase/insert are chunked LDS shifts, and the O(m^2) deque chaining
// of structural variants is replaced by "next matching junction" min-reductions that visit the junctions in
// exactly the reference's order.
#pragma once
#include "ambi_common.hpp"
#include "ambi_eval.hpp"
#include "ambi_group.hpp"

namespace ambi {

struct OutJunc { int32_t u, v, count; };   // consecutive vertices (u,v) of a path that are not a reference adjacency

// std::find(path+from, path+to, val): first index in [from,to) or `to`.  from > to behaves like an empty range.
template <class G>
AMBI_HD int find_first(const G& g, const cell_t* path, int from, int to, int val) {
    int best = 0x7fffffff;
    for (int i = from + g.tid(); i < to; i += g.size())
        if (path[i] == val) { best = i; break; }
    best = g.min_i32(best);
    return best == 0x7fffffff ? to : best;
}

// Scratch for indel_bfb: sv[m] junction indices, taken[m] flags, grp[2*m+4] deque storage.
struct IndelScratch {
    int32_t* sv;
    uint8_t* taken;
    int32_t* grp;
};

// LGM.cpp:3746-3837.  path/P are updated in place.  Returns 1 when the reference prints the
// "BFB path with insertion, deletion, or duplication:" caption (any qualifying SV exists), 0 otherwise,
// negative Status on capacity errors.
template <class G>
AMBI_HD int indel_bfb(const G& g, int n, const Junction* juncs, int m, cell_t* path, int* P_io, int pcap,
                      const IndelScratch& S) {
    int P = *P_io;
    // -- qualifying SVs in junction order (LGM.cpp:3750-3759), compacted with a group scan
    int nsv = 0;
    for (int base = 0; base < m; base += g.size()) {
        int ji = base + g.tid();
        int q = 0;
        if (ji < m) {
            const Junction& J = juncs[ji];
            int s = J.src, t = J.tgt;
            bool in = !(s < 1 || s > n || t < 1 || t > n);
            bool fbi = (J.sdir != J.tdir) && iabs(s - t) <= 2;
            bool normal = (J.sdir == J.tdir) && ((J.sdir > 0 && t - s == 1) || (J.sdir < 0 && s - t == 1));
            q = (in && !fbi && !normal) ? 1 : 0;
        }
        int tot;
        int ex = g.exscan_i32(q, &tot);
        if (q) { S.sv[nsv + ex] = ji; S.taken[nsv + ex] = 0; }
        nsv += tot;
    }
    g.sync();
    if (nsv == 0) return 0;

    int start_scan = 0;
    while (true) {
        // first junction not yet consumed starts a new group
        int first = 0x7fffffff;
        for (int i = start_scan + g.tid(); i < nsv; i += g.size())
            if (!S.taken[i]) { first = i; break; }
        first = g.min_i32(first);
        if (first == 0x7fffffff) break;
        start_scan = first + 1;
        int head = m + 2, tail = m + 2;
        {
            const Junction& J = juncs[S.sv[first]];
            g.sync();
            if (g.tid() == 0) { S.grp[tail] = a_src(J); S.grp[tail + 1] = a_tgt(J); S.taken[first] = 1; }
            tail += 2;
            g.sync();
        }
        int cursor = first + 1;
        while (true) {
            int front = S.grp[head], back = S.grp[tail - 1];
            int cand = 0x7fffffff;
            for (int i = cursor + g.tid(); i < nsv; i += g.size()) {
                if (S.taken[i]) continue;
                const Junction& J = juncs[S.sv[i]];
                if (a_tgt(J) == front || b_tgt(J) == front || back == a_src(J) || back == b_src(J)) { cand = i; break; }
            }
            cand = g.min_i32(cand);
            if (cand == 0x7fffffff) break;
            const Junction& J = juncs[S.sv[cand]];
            int nf = head, nt = tail;
            int wpos = -1, wval = 0;
            if (a_tgt(J) == front) { nf = head - 1; wpos = nf; wval = a_src(J); }
            else if (b_tgt(J) == front) { nf = head - 1; wpos = nf; wval = b_src(J); }
            else if (back == a_src(J)) { wpos = tail; wval = a_tgt(J); nt = tail + 1; }
            else { wpos = tail; wval = b_tgt(J); nt = tail + 1; }
            g.sync();
            if (g.tid() == 0) { S.grp[wpos] = wval; S.taken[cand] = 1; }
            head = nf; tail = nt;
            g.sync();
            cursor = cand + 1;
        }
        // -- apply the group (LGM.cpp:3779-3832)
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
                    int pos1 = find_first(g, path, 0, P, g0);
                    int pos2 = find_first(g, path, pos1 + 1, P, g1);
                    if (pos1 == P || pos2 == P) {
                        complement_all();
                        g0 = S.grp[head]; g1 = S.grp[head + 1];
                        pos1 = find_first(g, path, 0, P, g0);
                        pos2 = find_first(g, path, pos1 + 1, P, g1);
                    }
                    if (pos1 == P || pos2 == P || pos2 - pos1 > 3) continue;
                    int cnt = pos2 - (pos1 + 1);
                    if (cnt > 0) { shift_down(g, path, P, pos1 + 1, cnt); P -= cnt; }
                } else {   // duplication
                    int pos1 = find_first(g, path, 0, P, g0);
                    int pos2 = find_first(g, path, 0, pos1, g1);
                    if (pos1 == P || pos2 == pos1) {
                        complement_all();
                        g0 = S.grp[head]; g1 = S.grp[head + 1];
                        pos1 = find_first(g, path, 0, P, g0);
                        pos2 = find_first(g, path, 0, pos1, g1);
                    }
                    if (pos1 == P || pos2 == pos1) continue;
                    int cnt = pos1 + 1 - pos2;
                    if (P + cnt > pcap) { *P_io = P; return ST_ERR_PATH_CAPACITY; }
                    shift_up(g, path, P, pos1 + 1, cnt);
                    for (int k = g.tid(); k < cnt; k += g.size()) path[pos1 + 1 + k] = path[pos2 + k];
                    P += cnt;
                    g.sync();
                }
            } else {   // inversion
                int pos1 = find_first(g, path, 0, P, g0);
                int pos2 = find_first(g, path, pos1 + 1, P, g1);
                if (pos1 == P || pos2 == P) {
                    complement_all();
                    g0 = S.grp[head]; g1 = S.grp[head + 1];
                    pos1 = find_first(g, path, 0, P, g0);
                    pos2 = find_first(g, path, pos1 + 1, P, g1);
                }
                if (pos1 == P || pos2 == P || pos2 - pos1 > 5) continue;
                int cnt = pos2 - (pos1 + 1);
                if (cnt > 0) { shift_down(g, path, P, pos1 + 1, cnt); P -= cnt; }
            }
        } else {   // insertion
            int gf = S.grp[head], gb = S.grp[tail - 1];
            int pos1 = find_first(g, path, 0, P, gf);
            int pos2 = find_first(g, path, pos1 + 1, P, gb);
            if (pos1 == P || pos2 == P) {
                complement_all();
                gf = S.grp[head]; gb = S.grp[tail - 1];
                pos1 = find_first(g, path, 0, P, gf);
                pos2 = find_first(g, path, pos1 + 1, P, gb);
            }
            if (pos1 == P || pos2 == P) continue;
            int cnt = pos2 - (pos1 + 1);
            if (cnt > 0) { shift_down(g, path, P, pos1 + 1, cnt); P -= cnt; }
            int ins = gs - 2;
            if (P + ins > pcap) { *P_io = P; return ST_ERR_PATH_CAPACITY; }
            shift_up(g, path, P, pos1 + 1, ins);
            for (int k = g.tid(); k < ins; k += g.size()) path[pos1 + 1 + k] = (cell_t)S.grp[head + 1 + k];
            P += ins;
            g.sync();
        }
    }
    *P_io = P;
    return 1;
}

// localhap.cpp:267-289: consecutive path vertices that are not |id difference| == 1 on one strand become
// output junctions; a repeat (same edge or its complement edge) bumps the count.  First-appearance order.
// cand = scratch of P ints.  Returns the number of output junctions or a negative Status.
template <class G>
AMBI_HD int synth_out_juncs(const G& g, const cell_t* path, int P, OutJunc* out, int cap, int32_t* cand, int cand_cap) {
    if (P <= 0) return 0;
    int nc = 0;
    for (int base = 0; base + 1 < P; base += g.size()) {
        int i = base + g.tid();
        int q = 0;
        if (i + 1 < P) {
            int u = path[i], v = path[i + 1];
            bool adj = (iabs(iabs(u) - iabs(v)) == 1) && ((u > 0) == (v > 0));
            q = adj ? 0 : 1;
        }
        int tot;
        int ex = g.exscan_i32(q, &tot);
        if (q && nc + ex < cand_cap) cand[nc + ex] = i;
        nc += tot;
    }
    g.sync();
    if (nc > cand_cap) return ST_ERR_OUTJUNC_CAPACITY;
    int nout = 0, err = 0;
    if (g.tid() == 0) {
        for (int c = 0; c < nc; c++) {
            int u = path[cand[c]], v = path[cand[c] + 1];
            bool has = false;
            for (int k = 0; k < nout; k++)
                if ((out[k].u == u && out[k].v == v) || (-out[k].v == u && -out[k].u == v)) { has = true; out[k].count += 1; }
            if (!has) {
                if (nout >= cap) { err = 1; break; }
                out[nout].u = u; out[nout].v = v; out[nout].count = 1; nout++;
            }
        }
    }
    nout = g.bcast_i32(nout, 0);
    err = g.bcast_i32(err, 0);
    g.sync();
    return err ? (int)ST_ERR_OUTJUNC_CAPACITY : nout;
}

}  // namespace ambi
