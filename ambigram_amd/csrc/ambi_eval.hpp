// ambi_eval.hpp -- assembly of one candidate order into a fold-back palindrome breakpoint path.
//
// Restates the per-order body of LocalGenomicMap::getBFB (LGM.cpp:3519-3658) and imperfectFBI (:3431-3512) as
// SPMD code: the breakpoint path `bkp` (int16 signed local vertex ids) lives in the group's LDS; the reverse
// std::find with its parity / nesting tests becomes a strided scan + group max-reduction, the vector::insert
// of a loop's 4*cn breakpoints becomes a chunked LDS shift.  One wavefront evaluates one order.
#pragma once
#include "ambi_common.hpp"
#include "ambi_group.hpp"

namespace ambi {

typedef int16_t cell_t;

// fold-back map of the unit (getJuncCN's `inversions`): for local segment id i, the junction's two segment ids,
// or src[i]==0 when the segment has no entry.
struct InvMap {
    const int16_t* src;
    const int16_t* tgt;
};

template <class G>
AMBI_HD int find_last_slot(const G& g, const cell_t* bkp, int L, int target, bool less) {
    // LGM.cpp:3591-3602: last odd position f holding `target` that does not fail the nesting test
    int best = -1;
    for (int f = g.tid(); f < L; f += g.size()) {
        if ((f & 1) && bkp[f] == target) {
            bool skip = false;
            if (f < L - 2) {
                int x = iabs(bkp[f - 1]), y = iabs(bkp[f + 2]);
                skip = less ? (x < y) : (x > y);
            }
            if (!skip && f > best) best = f;
        }
    }
    return g.max_i32(best);
}

// open a gap of `count` cells at index `at` (cells [at,L) move up); group-cooperative, in place.  A round moves
// kShiftVec * size() cells: every thread reads its cells of the round (top-down, so nothing is overwritten before it was
// read by a LATER round), all wait, every thread writes -- two barriers per round instead of per size() cells.
constexpr int kShiftVec = 8;
template <class G, class T>
AMBI_HD void shift_up(const G& g, T* a, int L, int at, int count) {
    const int n = L - at, step = kShiftVec * g.size();
    for (int top = n; top > 0; top -= step) {
        T v[kShiftVec];
#pragma unroll
        for (int k = 0; k < kShiftVec; k++) {
            const int i = top - 1 - g.tid() - k * g.size();
            v[k] = i >= 0 ? a[at + i] : (T)0;
        }
        g.sync();
#pragma unroll
        for (int k = 0; k < kShiftVec; k++) {
            const int i = top - 1 - g.tid() - k * g.size();
            if (i >= 0) a[at + i + count] = v[k];
        }
        g.sync();
    }
}

// close the gap [at, at+count): cells [at+count, L) move down
template <class G, class T>
AMBI_HD void shift_down(const G& g, T* a, int L, int at, int count) {
    const int n = L - (at + count), step = kShiftVec * g.size();
    for (int base = 0; base < n; base += step) {
        T v[kShiftVec];
#pragma unroll
        for (int k = 0; k < kShiftVec; k++) {
            const int i = base + g.tid() + k * g.size();
            v[k] = i < n ? a[at + count + i] : (T)0;
        }
        g.sync();
#pragma unroll
        for (int k = 0; k < kShiftVec; k++) {
            const int i = base + g.tid() + k * g.size();
            if (i < n) a[at + i] = v[k];
        }
        g.sync();
    }
}

// LGM.cpp:3431-3512.  Returns false when the reference would have read out of bounds.
// The walk over pos is the reference's; inside a step nothing is left to one thread: the cells a step looks at are read
// by every thread (same addresses), the first later occurrence of -bkp[pos] is one flag rank per chunk of cells, the
// plain step's rewrite is computed by all and stored by one, and the rewrite of a palindrome -- in the reference a loop
// from the middle outwards -- runs one iteration per thread: iteration t touches the four cells p1, p1+1, p2-1, p2 with
// p1 = mid - 2t, p2 = mid + 1 + 2t, reads only bkp[p1], and no two iterations share a cell (parities), so they commute.
template <class G>
AMBI_HD bool imperfect_fbi(const G& g, cell_t* bkp, int L, const InvMap& inv) {
    int pos = 0;
    while (pos < L) {
        if (pos + 1 >= L) return false;
        const int c0 = bkp[pos], c1 = bkp[pos + 1];
        int r = L;
        if (pos + 3 < L) {
            const int want = -c0;
            for (int base = pos + 3; base < L; base += g.size()) {
                const int q = base + g.tid();
                const int hit = g.first_flag(q < L && bkp[q] == want);
                if (hit >= 0) { r = base + hit; break; }
            }
        }
        const int l = r - 1;
        const bool plain = (r == L) || (bkp[l] != -c1);
        g.sync();
        if (plain) {
            int n0 = c0, n1 = c1;
            {
                const int id = iabs(c1);
                const int s = inv.src[id];
                if (s != 0) {
                    const int t = inv.tgt[id];
                    n1 = c1 > 0 ? ((s < t) ? s : t) : ((s < t) ? -t : -s);
                }
            }
            if (pos > 0) {
                const int id = iabs(c0);
                const int s = inv.src[id];
                if (s != 0 && iabs(bkp[pos - 1]) == id) {
                    const int other = (s == id) ? inv.tgt[id] : s;
                    n0 = c0 > 0 ? other : -other;
                }
            }
            if (n0 > 0 && iabs(n0) > iabs(n1)) n1 = n0;
            if (n0 < 0 && iabs(n0) < iabs(n1)) n1 = n0;
            if (g.tid() == 0) { bkp[pos] = (cell_t)n0; bkp[pos + 1] = (cell_t)n1; }
            g.sync();
            pos += 2;
        } else {
            const int mid = pos + ((l - pos) / 2);
            int bad = 0;
            // iterations t = 0, 1, ..: p1 = mid - 2t while p1 >= pos - 1 and p1 > 0
            for (int t = g.tid();; t += g.size()) {
                const int p1 = mid - 2 * t, p2 = mid + 1 + 2 * t;
                if (!(p1 >= pos - 1 && p1 > 0)) break;
                const int v = bkp[p1];
                const int id = iabs(v);
                const int s = inv.src[id];
                if (s != 0) {
                    const int tt = inv.tgt[id];
                    if (p1 + 1 >= L) { bad = 1; break; }
                    int w0, w1;
                    if (v > 0) { if (s < tt) { w0 = s; w1 = -tt; } else { w0 = tt; w1 = -s; } }
                    else { if (s < tt) { w0 = -tt; w1 = s; } else { w0 = -s; w1 = tt; } }
                    bkp[p1] = (cell_t)w0; bkp[p1 + 1] = (cell_t)w1;
                    if (p2 != p1 + 1) {
                        if (p1 > pos - 1) { if (p2 >= L) { bad = 1; break; } bkp[p2] = (cell_t)-w0; }
                        bkp[p2 - 1] = (cell_t)-w1;
                    }
                }
            }
            g.sync();
            if (g.any(bad != 0)) return false;
            pos = r + 1;
        }
    }
    return true;
}

#if defined(__HIP_DEVICE_COMPILE__)
__device__ inline int imperfect_fbi_regs(cell_t* bkp, int L, const InvMap& inv);   // (below, with the register form of the placement)
#endif
// Second part of the evaluation: imperfectFBI (LGM.cpp:3656, always before the validity test) and the verdict.
// `placed` = what eval_place returned (>= 0).
template <class G>
AMBI_HD int eval_finish(const G& g, int placed, int K, cell_t* bkp, int L, const InvMap& inv, bool try_regs = false) {
    bool ok;
    int fast = -1;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (G::kLaneArrays) { if (try_regs) fast = imperfect_fbi_regs(bkp, L, inv); }   // cells in registers (up to 256 of them; stage_express)
#endif
    ok = fast >= 0 ? fast != 0 : imperfect_fbi(g, bkp, L, inv);
    // imperfectFBI reading past the end of the breakpoint vector (LGM.cpp:3436-3442 with pos+1 == end) is undefined in the
    // reference.  On an order that did not place all its elements the outcome is "invalid" whatever that read returns
    // (validity is i == K, fixed before imperfectFBI runs, and the breakpoints of an invalid order are thrown away,
    // LGM.cpp:3522), so the order is simply invalid here as well -- what the reference reports whenever it survives the
    // read.  On a VALID order the printed path would depend on the stray value: that is refused.
    if (!ok && placed == K) return ST_ERR_REF_UB;
    return (placed == K) ? 1 : 0;
}

// Placement part of the evaluation of one order (LGM.cpp:3519-3646): seeds and places the elements; needs the DAG only.
// Returns the number of elements placed (== K: all) or a negative Status; *L_out = bkp length.
// DAG: anything with K, pat[][3], loop[][3] -- Dag (up to 63 nodes) or WideDag (ambi_wide.hpp, up to 127).
template <class G, class DAG>
AMBI_HD int eval_place(const G& g, const DAG& D, const uint8_t* ord, bool forward, cell_t* bkp, int cap, int* L_out) {
    const int K = D.K;
    int L = 0;
    int x = ord[0];
    bool isPat = D.pat[x][0] != 0, isLoop = D.loop[x][0] != 0;
    if (!isPat && !isLoop) { *L_out = 0; return ST_ERR_REF_UB; }   // reference indexes an empty vector
    {
        int s = isPat ? D.pat[x][0] : D.loop[x][0], e = isPat ? D.pat[x][1] : D.loop[x][1];
        int q0, q1, q2, q3, len;
        if (forward) { q0 = s; q1 = e; q2 = -e; q3 = -s; } else { q0 = -e; q1 = -s; q2 = s; q3 = e; }
        len = isPat ? 2 : 4 * D.loop[x][2];
        if (len > cap) { *L_out = 0; return ST_ERR_BKP_CAPACITY; }
        for (int i = g.tid(); i < len; i += g.size()) {
            int w = i & 3;
            bkp[i] = (cell_t)(w == 0 ? q0 : w == 1 ? q1 : w == 2 ? q2 : q3);
        }
        L = len;
        g.sync();
    }
    int i;
    for (i = 1; i < K; i++) {
        x = ord[i];
        if (D.pat[x][0] != 0) {   // LGM.cpp:3572-3585
            int s = D.pat[x][0], e = D.pat[x][1];
            if (L == 0) { *L_out = 0; return ST_ERR_REF_UB; }
            int back = bkp[L - 1];
            if (L + 2 > cap) { *L_out = L; return ST_ERR_BKP_CAPACITY; }
            g.sync();
            if (back == -s) { if (g.tid() == 0) { bkp[L] = (cell_t)s; bkp[L + 1] = (cell_t)e; } L += 2; }
            else if (back == e) { if (g.tid() == 0) { bkp[L] = (cell_t)-e; bkp[L + 1] = (cell_t)-s; } L += 2; }
            else break;
            g.sync();
        } else if (D.loop[x][0] != 0) {   // LGM.cpp:3586-3644
            int s = D.loop[x][0], e = D.loop[x][1], cn = D.loop[x][2];
            // LGM.cpp:3591-3602: the last slot holding -s (nesting test "<"), else the last slot holding e (">"): both
            // searches in one sweep and one reduction (a -s hit outranks every e hit)
            int best = -1;
            for (int q = g.tid(); q < L; q += g.size()) {
                if (!(q & 1)) continue;
                const int c = bkp[q];
                if (c != -s && c != e) continue;
                bool skip = false;
                if (q < L - 2) {
                    const int x = iabs(bkp[q - 1]), y = iabs(bkp[q + 2]);
                    skip = (c == -s) ? (x < y) : (x > y);
                }
                if (!skip) { const int enc = (c == -s) ? 0x10000 + q : q; if (enc > best) best = enc; }
            }
            best = g.max_i32(best);
            if (best < 0) break;
            const bool viaV1 = best >= 0x10000;
            const int f = viaV1 ? best - 0x10000 : best;
            int cnt = 4 * cn;
            if (L + cnt > cap) { *L_out = L; return ST_ERR_BKP_CAPACITY; }
            bool hasNext = (f + 1 != L);
            shift_up(g, bkp, L, f + 1, cnt);
            int q0, q1, q2, q3, fix0, fix1;
            if (viaV1) { q0 = s; q1 = e; q2 = -e; q3 = -s; fix0 = -s; fix1 = s; }
            else { q0 = -e; q1 = -s; q2 = s; q3 = e; fix0 = e; fix1 = -e; }
            for (int k = g.tid(); k < cnt; k += g.size()) {
                int w = k & 3;
                bkp[f + 1 + k] = (cell_t)(w == 0 ? q0 : w == 1 ? q1 : w == 2 ? q2 : q3);
            }
            if (g.tid() == 0) {
                bkp[f] = (cell_t)fix0;                       // *temp (LGM.cpp:3627 / :3640)
                if (hasNext) bkp[f + 1 + cnt] = (cell_t)fix1;   // *(temp+1), written before the insert shifts it
            }
            L += cnt;
            g.sync();
        }
        // both slots empty (possible after the library sort for K > 16): nothing is placed, the loop goes on
    }
    *L_out = L;
    return i;
}

// The placement part for ONE wavefront with the breakpoint cells in REGISTERS (device code): cell i lives in lane i & 63,
// register i >> 6 -- up to kRegCells cells.  A placement of the group-memory form above costs ~2 700 cycles on a lone wavefront
// (a strided scan, a max-reduction, a chunked shift through group memory with two barriers per round, the writes: ~25 dependent
// round trips); here the candidate test is a compare per lane with its neighbours fetched by lane shuffles, the last candidate a
// wave-wide maximum, and the insert's shift one lane rotation per register.  Same cells, same return values; kRegsGiveUp when the
// path outgrows the registers (the caller then runs eval_place).  Writes the cells to `bkp` (group memory) at the end.
constexpr int kRegCells = 256;
constexpr int kRegsGiveUp = -1000;
#if defined(__HIP_DEVICE_COMPILE__)
// NR registers per lane hold 64 * NR cells; everything that loops over the registers is unrolled at compile time
template <int NR>
struct RegCells {
    int c[NR];
    // cell i, i uniform over the wavefront
    __device__ inline int get_u(int i) const {
        int v = c[0];
#pragma unroll
        for (int r = 1; r < NR; r++) v = (i >> 6) == r ? c[r] : v;
        return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(i & 63));
    }
    __device__ inline void set_u(int i, int v, int lane) {
        const bool me = lane == (i & 63);
#pragma unroll
        for (int r = 0; r < NR; r++) c[r] = (me && (i >> 6) == r) ? v : c[r];
    }
};
template <int NR, class DAG>
__device__ inline int eval_place_regs_n(const DAG& D, const uint8_t* ord, bool forward, cell_t* bkp, int cap, int* L_out) {
    constexpr int kCells = 64 * NR;
    const int lane = (int)(threadIdx.x & 63u);
    WaveGroup g;
    const int K = D.K;
    if (K > 64) return kRegsGiveUp;
    int L = 0;
    // the elements in placement order, one per lane: the loop below fetches element i with v_readlane instead of a chain of
    // dependent reads (order -> node -> record)
    int e_p0 = 0, e_p1 = 0, e_l0 = 0, e_l1 = 0, e_l2 = 0;
    if (lane < K) { const int xj = ord[lane]; e_p0 = D.pat[xj][0]; e_p1 = D.pat[xj][1]; e_l0 = D.loop[xj][0]; e_l1 = D.loop[xj][1]; e_l2 = D.loop[xj][2]; }
    auto elem = [&](int v, int i) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(i)); };
    const bool isPat = elem(e_p0, 0) != 0, isLoop = elem(e_l0, 0) != 0;
    if (!isPat && !isLoop) { *L_out = 0; return ST_ERR_REF_UB; }
    RegCells<NR> R;
    {
        const int s = isPat ? elem(e_p0, 0) : elem(e_l0, 0), e = isPat ? elem(e_p1, 0) : elem(e_l1, 0);
        int q0, q1, q2, q3;
        if (forward) { q0 = s; q1 = e; q2 = -e; q3 = -s; } else { q0 = -e; q1 = -s; q2 = s; q3 = e; }
        const int len = isPat ? 2 : 4 * elem(e_l2, 0);
        if (len > cap) { *L_out = 0; return ST_ERR_BKP_CAPACITY; }
        if (len > kCells) return kRegsGiveUp;
        const int w = lane & 3, v = w == 0 ? q0 : w == 1 ? q1 : w == 2 ? q2 : q3;      // (64 is a multiple of 4: the pattern is the same in every register)
#pragma unroll
        for (int r = 0; r < NR; r++) R.c[r] = 64 * r + lane < len ? v : 0;
        L = len;
    }
    int i;
    for (i = 1; i < K; i++) {
        const int p0 = elem(e_p0, i), l0 = elem(e_l0, i);
        if (p0 != 0) {   // LGM.cpp:3572-3585
            const int s = p0, e = elem(e_p1, i);
            if (L == 0) { *L_out = 0; return ST_ERR_REF_UB; }
            const int back = R.get_u(L - 1);
            if (L + 2 > cap) { *L_out = L; return ST_ERR_BKP_CAPACITY; }
            if (L + 2 > kCells) return kRegsGiveUp;
            if (back == -s) { R.set_u(L, s, lane); R.set_u(L + 1, e, lane); L += 2; }
            else if (back == e) { R.set_u(L, -e, lane); R.set_u(L + 1, -s, lane); L += 2; }
            else break;
        } else if (l0 != 0) {   // LGM.cpp:3586-3644
            const int s = l0, e = elem(e_l1, i), cn = elem(e_l2, i);
            // candidates: odd slots holding -s or e that pass the nesting test |cell[q-1]| vs |cell[q+2]| (only below L-2); the
            // neighbours come from the lane below (same register: q is odd) and from two lanes above (the next register for lanes 62, 63)
            const int up = (lane + 2) & 63;
            const bool wrap = lane >= 62;
            int above[NR + 1];
#pragma unroll
            for (int r = 0; r < NR; r++) above[r] = __shfl(R.c[r], up, 64);
            above[NR] = 0;
            int best = -1;
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const int q = 64 * r + lane, c = R.c[r];
                const bool cand = q < L && (q & 1) && (c == -s || c == e);
                const int below = __builtin_amdgcn_update_dpp(0, c, 0x111 /* row_shr:1 */, 0xf, 0xf, false);   // lane - 1 (candidates sit in odd lanes: never the first lane of a row)
                bool skip = false;
                if (q < L - 2) { const int xx = iabs(below), yy = iabs(wrap ? above[r + 1] : above[r]); skip = (c == -s) ? (xx < yy) : (xx > yy); }
                const int enc = (c == -s) ? 0x10000 + q : q;
                if (cand && !skip && enc > best) best = enc;
            }
            best = g.max_i32(best);
            if (best < 0) break;
            const bool viaV1 = best >= 0x10000;
            const int f = viaV1 ? best - 0x10000 : best;
            const int cnt = 4 * cn;
            if (L + cnt > cap) { *L_out = L; return ST_ERR_BKP_CAPACITY; }
            if (L + cnt > kCells) return kRegsGiveUp;
            const bool hasNext = (f + 1 != L);
            // cells [f+1, L) move up by cnt: cell i takes cell i - cnt -- a lane rotation by cnt & 63, the source register one lower for
            // the lanes the rotation wraps (and cnt >> 6 lower for long loops)
            {
                const int rot = cnt & 63, sl = (lane - rot) & 63, down = (cnt >> 6) + (lane < rot ? 1 : 0);
                int sh[NR];
#pragma unroll
                for (int r = 0; r < NR; r++) sh[r] = __shfl(R.c[r], sl, 64);
                int qq0, qq1, qq2, qq3;
                if (viaV1) { qq0 = s; qq1 = e; qq2 = -e; qq3 = -s; } else { qq0 = -e; qq1 = -s; qq2 = s; qq3 = e; }
                const int w = (lane - (f + 1)) & 3, ins = w == 0 ? qq0 : w == 1 ? qq1 : w == 2 ? qq2 : qq3;   // (the register base is a multiple of 4)
#pragma unroll
                for (int r = NR - 1; r >= 0; r--) {
                    const int ci = 64 * r + lane;
                    int moved = 0;
#pragma unroll
                    for (int t = 0; t < NR; t++) moved = (r - down == t) ? sh[t] : moved;
                    R.c[r] = ci >= f + 1 + cnt ? moved : (ci >= f + 1 ? ins : R.c[r]);
                }
            }
            R.set_u(f, viaV1 ? -s : e, lane);                          // *temp (LGM.cpp:3627 / :3640)
            if (hasNext) R.set_u(f + 1 + cnt, viaV1 ? s : -e, lane);   // *(temp+1): the cell that stood behind f, now behind the insert
            L += cnt;
        }
        // both slots empty (possible after the library sort for K > 16): nothing is placed, the loop goes on
    }
#pragma unroll
    for (int r = 0; r < NR; r++) if (64 * r + lane < L) bkp[64 * r + lane] = (cell_t)R.c[r];
    g.sync();
    *L_out = L;
    return i;
}
// imperfectFBI (LGM.cpp:3431-3512) with the breakpoint cells in registers (cell i in lane i mod 64 of register i / 64).
// The reference walks pos over the path; a step either rewrites the two cells at pos ("plain") or a whole palindrome
// [pos, r] from its middle outwards.  What a step SEES ahead of pos is untouched by the steps before it (a step writes
// cells pos-1 .. r only), so the sequence of steps follows from the ORIGINAL cells alone, and the iterations p1 = mid - 2t
// of a palindrome read nothing but their own original cell bkp[p1] -- except the last one, p1 = pos - 1, whose cell the
// previous step may have rewritten (plain steps read bkp[pos-1] in the same way).  A step of a serial walk costs a lone
// wavefront 600-1000 cycles (a chain of vector -> scalar hand-overs), so everything that can be is done for all cells at once:
//   pass 0  every cell j as if the walk stood there: r(j) = first q >= j + 3 with cell[q] == -cell[j] (one v_readlane per q,
//           compared by all lanes), plain(j), next(j), mid(j);
//   pass 1  the walk itself is then one v_readlane per step (next(pos)); it gives every cell of a palindrome its (pos, mid);
//   pass 2  ALL palindromes at once, every lane for the cell it holds: cell j takes w0(j) if j is a p1 of its palindrome,
//           w1(j-1) if j-1 is, -w0(2 mid + 1 - j) as the mirrored p2, -w1(2 mid - j) as p2 - 1 (disjoint by parity, as the
//           reference's iterations never share a cell): one shift by a lane and two lane reversals per register;
//   pass 3  the walk again, in order, for what depends on the step before: the plain rewrites and the iteration
//           p1 = pos - 1 of every palindrome (cells pos-1, pos and p2-1; no other iteration of that palindrome writes them).
// Cells are read from / written back to `bkp` (group memory) once.  Returns false where imperfect_fbi does.
template <int NR>
__device__ inline bool imperfect_fbi_regs_n(cell_t* bkp, int L, const InvMap& inv) {
    const int lane = (int)(threadIdx.x & 63u);
    RegCells<NR> R, O;          // current / original cells
    int MID[NR], POS[NR];       // the palindrome a cell lies in (-1: none)
#pragma unroll
    for (int r = 0; r < NR; r++) { R.c[r] = 64 * r + lane < L ? (int)bkp[64 * r + lane] : 0; O.c[r] = R.c[r]; MID[r] = -1; POS[r] = -1; }
    // pass 0.  Cells behind L hold 0 and no cell of the path is 0: they match nothing.
    RegCells<NR> NXP;           // next(j) * 2 + plain(j)
    {
        int RH[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) RH[r] = L;
#pragma unroll
        for (int t = NR - 1; t >= 0; t--) {
#pragma unroll
            for (int ql = 63; ql >= 0; ql--) {      // descending: the smallest q is assigned last.  Fully unrolled: the lane of every
                const int q = 64 * t + ql;          // v_readlane is a constant and the iterations do not wait for one another
                if (q < 3) continue;
                const int want_of = -__builtin_amdgcn_readlane(O.c[t], ql);     // cell j matches if cell[j] == -cell[q]  (0 behind L: no match)
#pragma unroll
                for (int r = 0; r < NR; r++) if (64 * r + 3 <= q) RH[r] = (64 * r + lane + 3 <= q && O.c[r] == want_of) ? q : RH[r];
            }
        }
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int j = 64 * r + lane, idx = RH[r] - 1;      // cell l = r - 1 and cell j + 1
            int ol = 0, o1 = 0;
#pragma unroll
            for (int t = 0; t < NR; t++) {
                const int a = __shfl(O.c[t], idx & 63, 64), b = __shfl(O.c[t], (lane + 1) & 63, 64);
                ol = (idx >> 6) == t ? a : ol;
                o1 = (lane == 63 ? r + 1 : r) == t ? b : o1;
            }
            const bool plain = RH[r] == L || ol != -o1;
            NXP.c[r] = plain ? 2 * (j + 2) + 1 : 2 * (RH[r] + 1);
        }
    }
    // pass 1
    bool any_plain = false;
    for (int pos = 0; pos < L;) {
        if (pos + 1 >= L) return false;
        const int np = NXP.get_u(pos), nxt = np >> 1;
        if (!(np & 1)) {
            const int rh = nxt - 1, mid = pos + ((rh - 1 - pos) / 2);
#pragma unroll
            for (int r = 0; r < NR; r++) { const int j = 64 * r + lane; const bool in = j >= pos && j <= rh; MID[r] = in ? mid : MID[r]; POS[r] = in ? pos : POS[r]; }
        } else any_plain = true;
        pos = nxt;
    }
    // pass 2
    {
        int W0[NR], W1[NR];     // what iteration "p1 = my cell" writes (0: nothing)
        bool bad = false;
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int j = 64 * r + lane, mid = MID[r], pos = POS[r];
            W0[r] = 0; W1[r] = 0;
            if (mid >= 0 && j >= (pos > 1 ? pos : 1) && j <= mid && ((mid - j) & 1) == 0) {
                const int v = O.c[r], id = iabs(v);
                const int s = inv.src[id];
                if (s != 0) {
                    const int tt = inv.tgt[id];
                    if (j + 1 >= L) bad = true;
                    if (v > 0) { if (s < tt) { W0[r] = s; W1[r] = -tt; } else { W0[r] = tt; W1[r] = -s; } }
                    else { if (s < tt) { W0[r] = -tt; W1[r] = s; } else { W0[r] = -s; W1[r] = tt; } }
                    const int p2 = 2 * mid + 1 - j;
                    if (p2 != j + 1 && p2 >= L) bad = true;
                }
            }
        }
        if (__ballot(bad)) return false;
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int j = 64 * r + lane, mid = MID[r], pos = POS[r];
            const int lo = pos > 1 ? pos : 1;
            const int i1 = 2 * mid + 1 - j, i0 = 2 * mid - j;     // the p1 whose p2 / p2 - 1 this cell is
            int below = 0, rev0 = 0, rev1 = 0;
#pragma unroll
            for (int t = 0; t < NR; t++) {
                const int b = __shfl(W1[t], (lane + 63) & 63, 64);
                below = (lane == 0 ? r - 1 : r) == t ? b : below;
                const int x0 = __shfl(W0[t], i1 & 63, 64), x1 = __shfl(W1[t], i0 & 63, 64);
                rev0 = (i1 >> 6) == t ? x0 : rev0;
                rev1 = (i0 >> 6) == t ? x1 : rev1;
            }
            int c = R.c[r];
            if (W0[r] != 0) c = W0[r];                                                                  // a p1
            if (mid >= 0 && j >= 1 && j - 1 >= lo && j - 1 <= mid && below != 0) c = below;             // p1 + 1
            if (mid >= 0 && i1 >= lo && i1 < mid && rev0 != 0) c = -rev0;                               // p2 (not of t = 0)
            if (mid >= 0 && i0 >= lo && i0 < mid && rev1 != 0) c = -rev1;                               // p2 - 1 (not of t = 0)
            R.c[r] = c;
        }
    }
    // pass 3, all at once when no step feeds on what the step before it writes here: no plain step, and wherever the last
    // iteration of a palindrome writes the cell in front of the NEXT palindrome (its p2 - 1 is that palindrome's r) and the next
    // one reads it, the value written is the value the cell holds already (always so with perfect fold-backs, whose rewrites
    // change nothing).  The cell in front of palindrome (pos, mid) is pos - 1: its lane computes the iteration (w0 into its
    // own cell), cell pos takes w1 from the lane below, cell 2 mid - pos + 1 takes -w1; a later step's write wins.
    bool done3 = false;
    if (!any_plain) {
        int W0[NR], W1[NR];
        bool reads[NR];          // my cell is the p1 of the palindrome that starts behind it
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int j = 64 * r + lane;
            int mid_n = -1, pos_n = -1;          // the palindrome of cell j + 1
#pragma unroll
            for (int t = 0; t < NR; t++) {
                const int a = __shfl(MID[t], (lane + 1) & 63, 64), b = __shfl(POS[t], (lane + 1) & 63, 64);
                const bool me = (lane == 63 ? r + 1 : r) == t;
                mid_n = me ? a : mid_n; pos_n = me ? b : pos_n;
            }
            W0[r] = 0; W1[r] = 0;
            reads[r] = j >= 1 && j + 1 < L && pos_n == j + 1 && ((mid_n - j) & 1) == 0;
            if (reads[r]) {
                const int v = R.c[r], id = iabs(v);
                const int s = inv.src[id];
                if (s != 0) {
                    const int tt = inv.tgt[id];
                    if (v > 0) { if (s < tt) { W0[r] = s; W1[r] = -tt; } else { W0[r] = tt; W1[r] = -s; } }
                    else { if (s < tt) { W0[r] = -tt; W1[r] = s; } else { W0[r] = -s; W1[r] = tt; } }
                }
            }
        }
        bool conflict = false;
        int below[NR], far[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int pos = POS[r];
            below[r] = 0; far[r] = 0;
#pragma unroll
            for (int t = 0; t < NR; t++) {       // (every lane takes part in the exchanges)
                const int b = __shfl(W1[t], (lane + 63) & 63, 64), a = __shfl(W1[t], (pos - 1) & 63, 64);
                below[r] = (lane == 0 ? r - 1 : r) == t ? b : below[r];
                far[r] = ((pos - 1) >> 6) == t ? a : far[r];
            }
            const int x = 64 * r + lane, mid = MID[r];
            const bool third = mid >= 0 && pos >= 2 && x == 2 * mid - pos + 1 && far[r] != 0;   // my cell is the p2 - 1 of my palindrome's last iteration
            if (!third) far[r] = 0;
            if (third && reads[r] && -far[r] != R.c[r]) conflict = true;     // ... and the next palindrome reads it: it must find what it assumed
        }
        if (!__ballot(conflict)) {
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const int x = 64 * r + lane;
                int c = R.c[r];
                if (far[r] != 0) c = -far[r];                                                   // p2 - 1 of the palindrome I lie in
                if (MID[r] >= 0 && POS[r] == x && x >= 2 && below[r] != 0) c = below[r];        // p1 + 1 = pos
                if (W0[r] != 0) c = W0[r];                                                      // p1 = pos - 1 of the palindrome behind me (a later step)
                R.c[r] = c;
            }
            done3 = true;
        }
    }
    // pass 3, step by step
    for (int pos = 0; pos < L && !done3;) {
        const int np = NXP.get_u(pos), nxt = np >> 1;
        if (np & 1) {
            const int c0 = O.get_u(pos), c1 = O.get_u(pos + 1);
            int n0 = c0, n1 = c1;
            {
                const int id = iabs(c1);
                const int s = inv.src[id];
                if (s != 0) {
                    const int t = inv.tgt[id];
                    n1 = c1 > 0 ? ((s < t) ? s : t) : ((s < t) ? -t : -s);
                }
            }
            if (pos > 0) {
                const int id = iabs(c0);
                const int s = inv.src[id];
                if (s != 0 && iabs(R.get_u(pos - 1)) == id) {
                    const int other = (s == id) ? inv.tgt[id] : s;
                    n0 = c0 > 0 ? other : -other;
                }
            }
            if (n0 > 0 && iabs(n0) > iabs(n1)) n1 = n0;
            if (n0 < 0 && iabs(n0) < iabs(n1)) n1 = n0;
            R.set_u(pos, n0, lane); R.set_u(pos + 1, n1, lane);
        } else {
            const int rh = nxt - 1, mid = pos + ((rh - 1 - pos) / 2);
            const int p1 = pos - 1;
            if (p1 >= 1 && ((mid - p1) & 1) == 0) {     // the last iteration of this palindrome: the cell in front of it
                const int v = R.get_u(p1), id = iabs(v);
                const int s = inv.src[id];
                if (s != 0) {
                    const int tt = inv.tgt[id];
                    int w0, w1;
                    if (v > 0) { if (s < tt) { w0 = s; w1 = -tt; } else { w0 = tt; w1 = -s; } }
                    else { if (s < tt) { w0 = -tt; w1 = s; } else { w0 = -s; w1 = tt; } }
                    R.set_u(p1, w0, lane); R.set_u(p1 + 1, w1, lane);
                    const int p2 = 2 * mid + 1 - p1;
                    if (p2 != p1 + 1) R.set_u(p2 - 1, -w1, lane);
                }
            }
        }
        pos = nxt;
    }
#pragma unroll
    for (int r = 0; r < NR; r++) if (64 * r + lane < L) bkp[64 * r + lane] = (cell_t)R.c[r];
    WaveGroup g;
    g.sync();
    return true;
}
// -1: path too long for the register form
__device__ inline int imperfect_fbi_regs(cell_t* bkp, int L, const InvMap& inv) {
    if (L <= 64) return imperfect_fbi_regs_n<1>(bkp, L, inv) ? 1 : 0;
    if (L <= 128) return imperfect_fbi_regs_n<2>(bkp, L, inv) ? 1 : 0;
    return -1;   // (pass 0 costs L x (2 + 4 NR) instructions: beyond two registers the serial walk in group memory is no slower)
}
template <class DAG>
__device__ inline int eval_place_regs(const DAG& D, const uint8_t* ord, bool forward, cell_t* bkp, int cap, int* L_out) {
    if (cap <= 64) return eval_place_regs_n<1>(D, ord, forward, bkp, cap, L_out);
    if (cap <= 128) return eval_place_regs_n<2>(D, ord, forward, bkp, cap, L_out);
    if (cap <= 256) return eval_place_regs_n<4>(D, ord, forward, bkp, cap, L_out);
    return kRegsGiveUp;
}
#endif

// Evaluate one order.  Returns 1 valid / 0 invalid / negative Status on error.  *L_out = bkp length.
template <class G, class DAG>
AMBI_HD int eval_order(const G& g, const DAG& D, const uint8_t* ord, bool forward, const InvMap& inv,
                       cell_t* bkp, int cap, int* L_out, int64_t* clk = nullptr) {
    const int K = D.K;
    int L = 0;
    const int i = eval_place(g, D, ord, forward, bkp, cap, &L);
    *L_out = L;
    if (i < 0) return i;
    clk_mark(g, clk, 25);
    return eval_finish(g, i, K, bkp, L, inv);
}


// LGM.cpp:3661-3670: breakpoint pairs -> per-segment path (int16 local signed ids).  `offs` = scratch of
// L/2+1 ints.  Returns P or a negative Status.
template <class G, class GP = int16_t>
AMBI_HD int expand_bkp(const G& g, const cell_t* bkp, int L, cell_t* path, int pcap, int32_t* offs, GP* gpath = nullptr,
                       int seg_base = 0) {
    int np = L / 2;
    int carry = 0;
    for (int base = 0; base < np; base += g.size()) {
        int j = base + g.tid();
        int len = 0;
        if (j < np) {
            int a = bkp[2 * j], b = bkp[2 * j + 1];
            if (a > 0) { len = iabs(b) - a + 1; } else { len = (-a) - iabs(b) + 1; }
            if (len < 0) len = 0;
        }
        int tot;
        int ex = g.exscan_i32(len, &tot);
        if (j < np) offs[j] = carry + ex;
        carry += tot;
    }
    const int P = carry;
    if (g.tid() == 0) offs[np] = P;
    g.sync();
    if (P > pcap) return ST_ERR_PATH_CAPACITY;
    // run-major: a sub-group of up to 64 threads (one wavefront on the GPU) writes the run of one pair; both the
    // '+' run a, a+1, .. and the '-' run -|a|, -(|a|-1), .. are  a + k.  The absolute-id copy for the result blob
    // (gpath, optional) leaves in the same pass.
    const int lanes = g.size() < 64 ? g.size() : 64;
    const int sub = g.tid() / lanes, nsub = g.size() / lanes, lane = g.tid() - sub * lanes;
    for (int j = sub; j < np; j += nsub) {
        const int a = bkp[2 * j], o0 = offs[j], len = offs[j + 1] - o0;
        for (int k = lane; k < len; k += lanes) {
            const int v = a + k;
            if (path) path[o0 + k] = (cell_t)v;
            // the result blob holds local ids in 2 bytes (readers add the base); 4-byte callers (--all paths) get absolute ids
            if (gpath) gpath[o0 + k] = sizeof(GP) == 2 ? (GP)v : (GP)(v > 0 ? v + seg_base : v - seg_base);
        }
    }
    g.sync();
    return P;
}

}  // namespace ambi
